#!/usr/bin/env python3
"""Gibbs K build at the launch-ramp sizes (N = 1024 .. 4096) next to a plain fill of the same buffer: min over 30 single
launches (HIP events) and the average of 30 back-to-back launches.  python tools/probes/build_small.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
from nsgp import ops  # noqa: E402

dev = torch.device('cuda', 0)


def single(fn, reps=30):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3


def chain(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N in (1024, 2048, 4096, 8192):
    for dt in (torch.float32, torch.float64):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, 2, generator=g, dtype=torch.float64).to(dt).to(dev)
        ell = torch.exp(0.3 * torch.randn(2, N, generator=g, dtype=torch.float64) - 1.2).to(dt).to(dev).contiguous()
        os_ = torch.tensor(0.644, dtype=dt, device=dev)
        nz = torch.tensor(0.011, dtype=dt, device=dev)
        K = torch.empty(N, N, dtype=dt, device=dev)
        build = lambda: ops.gibbs_build(x, x, ell, ell, outputscale=os_, diag_add=nz, out=K)
        fill = lambda: K.fill_(1.5)
        nbytes = K.numel() * K.element_size()
        tb, tbc, tf, tfc = single(build), chain(build), single(fill), chain(fill)
        print(f'N={N:5d} {str(dt)[6:]:8s} build {tb:6.1f} us ({nbytes / tb / 1e3:5.0f} GB/s) back-to-back {tbc:6.1f} us '
              f'({nbytes / tbc / 1e3:5.0f} GB/s) | fill_ {tf:6.1f} us ({nbytes / tf / 1e3:5.0f} GB/s) back-to-back {tfc:6.1f} us '
              f'({nbytes / tfc / 1e3:5.0f} GB/s)', flush=True)
