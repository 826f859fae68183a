#!/usr/bin/env python3
"""Is potrf's time data-dependent?  N = 4096 float64: Gibbs K on random points, on the B2 lattice, and a random SPD matrix."""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd')); sys.path.insert(0, ROOT)
import torch
from nsgp import ops
import bench
dev = 'cuda'
n = 4096
for dt in (torch.float64, torch.float32):
    g = torch.Generator().manual_seed(173)
    x = torch.randn(n, 2, generator=g).to(dev, dt)
    e = torch.exp(0.3 * torch.randn(2, n, generator=g) + math.log(0.3)).to(dev, dt)
    os_ = torch.tensor([0.644], dtype=dt, device=dev); nz = torch.tensor([0.011], dtype=dt, device=dev)
    Kr = ops.gibbs_build(x, x, e, e, os_, nz)
    xl, el = bench._b2_inputs(n, dev, dt)
    Kl = ops.gibbs_build(xl, xl, el, el, os_, nz)
    Q = torch.randn(n, n, generator=g).to(dev, dt)
    Ks = Q @ Q.T / n + torch.eye(n, device=dev, dtype=dt)
    for name, K in (('random x', Kr), ('lattice', Kl), ('random SPD', Ks)):
        t = bench._timeit(lambda: ops.potrf(K), reps=5)
        L, info = ops.potrf(K)
        print(dt, name, 'potrf %.3f ms' % t, 'info', int(info.max()), 'min|L|>0', float(L[L != 0].abs().min()), 'subnormal entries', int(((L != 0) & (L.abs() < (2.3e-308 if dt == torch.float64 else 1.2e-38))).sum()))
