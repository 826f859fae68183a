#!/usr/bin/env python3
"""Sustained matrix-core rates of the chip (nsgp_mfma_rate_probe: register-only MFMA loops, no memory traffic), per
instruction and grid size, with the shader clock the chip held meanwhile -- the context for bench.py's roofline fractions.

    python tools/probes/mfma_rate.py
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from nsgp import _lib, ops  # noqa: E402

PEAK = {'f32 32x32x2': 157.3, 'f64 16x16x4': 78.6, 'i8 32x32x32': 5000.0}


def main():
    dev = torch.device('cuda', 0)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    for name, kind, iters, ops_per in (('f32 32x32x2', 0, 400, 4096), ('f64 16x16x4', 1, 400, 2048), ('i8 32x32x32', 2, 800, 65536)):
        for wgs in (256, 512, 1024, 2048, 4096):
            buf = torch.zeros(8 * wgs, dtype=torch.int64, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e30
            for _ in range(3):
                e0.record()
                _lib.call('nsgp_mfma_rate_probe', kind, wgs, iters, ops._p(buf), ops._p(sink), ops._stream())
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1))
            a = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
            clk = np.median(a[:, 0] / a[:, 1]) * 0.1
            cyc = np.median(a[:, 0]) / (iters * 8)
            rate = wgs * 4 * iters * 8 * ops_per / (best * 1e-3) / 1e12
            print(f'{name:12s} {wgs:5d} workgroups x 4 waves: {best * 1e3:8.1f} us  {rate:8.1f} T(FL)OP/s = {rate / PEAK[name]:5.3f} of the '
                  f'data-sheet peak; shader clock {clk:5.3f} GHz; {cyc:6.1f} cycles per MFMA per wave')


if __name__ == '__main__':
    main()
