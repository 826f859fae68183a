#!/usr/bin/env python3
"""Gibbs K build + potrf(K + noise I) probe (SURVEY 8d, cfg2 synthetic): the workload behind the
`gibbs_build_*` / `potrf_*` fields of bench.py, sized by N, for rocprofv3 kernel-trace / PMC passes.

    python tools/build_chol_probe.py [N ...]          default: 4096 16384
"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
from nsgp import ops  # noqa: E402


def run(N, dtype, reps=3):
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(173)
    side = int(round(N ** 0.5))
    gx, gy = torch.meshgrid(torch.arange(side, dtype=torch.float64), torch.arange(N // side, dtype=torch.float64),
                            indexing='ij')
    x = torch.stack([gx.reshape(-1), gy.reshape(-1)], -1)
    x = ((x - x.mean(0)) / x.std(0)).to(dtype).to(dev)
    n = x.shape[0]
    ell = torch.exp(0.3 * torch.randn(2, n, generator=g, dtype=torch.float64) + torch.log(torch.tensor(0.3)))
    ell = ell.to(dtype).to(dev).contiguous()
    os_ = torch.tensor(0.644, dtype=dtype, device=dev)
    noise = torch.tensor(0.011, dtype=dtype, device=dev)
    K = torch.empty(n, n, dtype=dtype, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tb, tc = [], []
    for _ in range(reps + 1):
        ev[0].record()
        ops.gibbs_build(x, x, ell, ell, outputscale=os_, diag_add=noise, out=K)
        ev[1].record()
        L, info = ops.potrf(K, overwrite=True)
        ev[2].record()
        torch.cuda.synchronize()
        tb.append(ev[0].elapsed_time(ev[1]))
        tc.append(ev[1].elapsed_time(ev[2]))
    assert int(info.max()) == 0
    s = K.element_size()
    tb, tc = min(tb[1:]), min(tc[1:])
    gb = s * (n * n + 2 * 2 * (n + n)) / 1e9
    print(f'N={n} {str(dtype)[6:]}: build {tb * 1e3:9.1f} us  {gb / (tb * 1e-3):8.1f} GB/s (algorithmic {gb * 1e3:.1f} MB)'
          f' | potrf {tc:8.3f} ms  {n ** 3 / 3 / (tc * 1e-3) / 1e12:6.2f} TFLOP/s', flush=True)


if __name__ == '__main__':
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 16384]
    for N in sizes:
        for dt in (torch.float32, torch.float64):
            run(N, dt)
