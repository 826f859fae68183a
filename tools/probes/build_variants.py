#!/usr/bin/env python3
"""Times the Gibbs build (min of 20) for every experimental library tools/probes/_bin/lib_*.so, each in its own process:
    python tools/probes/build_variants.py"""
import glob
import os
import shutil
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, 'nonstationary-precip_amd', 'nsgp', 'libnsgp_hip.so')

CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from nsgp import ops
dev = torch.device('cuda', 0)
for N in (4096, 16384):
    for dt in (torch.float32, torch.float64):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, 2, generator=g, dtype=torch.float64).to(dt).to(dev)
        ell = torch.exp(0.3 * torch.randn(2, N, generator=g, dtype=torch.float64) - 1.2).to(dt).to(dev).contiguous()
        os_ = torch.tensor(0.644, dtype=dt, device=dev); nz = torch.tensor(0.011, dtype=dt, device=dev)
        K = torch.empty(N, N, dtype=dt, device=dev)
        best = 1e9
        for _ in range(20):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gibbs_build(x, x, ell, ell, outputscale=os_, diag_add=nz, out=K); e1.record()
            torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
        print('   N=%%5d %%-8s %%7.1f us  %%6.0f GB/s' %% (N, str(dt)[6:], best * 1e3, K.numel() * K.element_size() / best / 1e6), flush=True)
''' % os.path.join(ROOT, 'nonstationary-precip_amd')

keep = LIB + '.keep'
shutil.copy(LIB, keep)
try:
    for lib in sorted(glob.glob(os.path.join(ROOT, 'tools', 'probes', '_bin', 'lib_*.so'))):
        shutil.copy(lib, LIB)
        print(os.path.basename(lib), flush=True)
        subprocess.run([sys.executable, '-c', CHILD], check=True)
finally:
    shutil.move(keep, LIB)
