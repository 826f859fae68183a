#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM on the six (M x M x n) products of one SVGP layer step
(BASELINE B4 last layer: M=1024, n=S*B=40960).  Prints TFLOP/s (algorithmic: triangular halves)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
from nsgp import ops  # noqa: E402

M, n = 1024, int(os.environ.get('NCOLS', 40960))
reps = int(os.environ.get('REPS', 10))
dt = torch.float32
g = torch.Generator().manual_seed(0)
W = torch.tril(torch.randn(M, M, generator=g)).cuda()
Lq = torch.tril(torch.randn(M, M, generator=g)).cuda()
K = torch.randn(M, n, generator=g).cuda()
A = torch.randn(M, n, generator=g).cuda()
out = torch.empty(M, n, device='cuda')

cases = [
    ('A=W K      NN  A_LOWER', lambda: ops.gemm(W, K, flags=ops.GEMM_A_LOWER), M * M * n),
    ('C=Lq^T A   TN  A_UPPER', lambda: ops.gemm(Lq, A, ta=True, flags=ops.GEMM_A_UPPER), M * M * n),
    ('Ab+=Lq C2  NN  beta=1 ', lambda: ops.gemm(Lq, K, flags=ops.GEMM_A_LOWER, beta=1.0, out=out), M * M * n),
    ('Lqb=A C2^T NT  C_LOWER', lambda: ops.gemm(A, K, tb=True, flags=ops.GEMM_C_LOWER), M * M * n),
    ('Kb=W^T Ab  TN  A_UPPER', lambda: ops.gemm(W, A, ta=True, flags=ops.GEMM_A_UPPER), M * M * n),
    ('plain NN   full       ', lambda: ops.gemm(W, K), 2 * M * M * n),
]
only = os.environ.get('ONLY')
for name, fn, flops in cases:
    if only is not None and only not in name:
        continue
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f'{name}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s', flush=True)
