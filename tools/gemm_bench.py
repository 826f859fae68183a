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
PAD = int(os.environ.get('PAD', 0))


def padded(t):
    if not PAD:
        return t.cuda()
    buf = torch.empty(t.shape[0], t.shape[1] + PAD, device='cuda')
    v = buf[:, :t.shape[1]]
    v.copy_(t)
    return v


K = padded(torch.randn(M, n, generator=g))
A = padded(torch.randn(M, n, generator=g))
W = padded(W.cpu())
Lq = padded(Lq.cpu())
out = torch.empty(M, n, device='cuda')
Bt = torch.randn(n, M, generator=g).cuda()
At = torch.randn(n, M, generator=g).cuda()
print('PAD', PAD, 'strides', K.stride(), W.stride())

cases = [
    ('A=W K      NN  A_LOWER', lambda: ops.gemm(W, K, flags=ops.GEMM_A_LOWER), M * M * n),
    ('C=Lq^T A   TN  A_UPPER', lambda: ops.gemm(Lq, A, ta=True, flags=ops.GEMM_A_UPPER), M * M * n),
    ('Ab+=Lq C2  NN  beta=1 ', lambda: ops.gemm(Lq, K, flags=ops.GEMM_A_LOWER, beta=1.0, out=out), M * M * n),
    ('Lqb=A C2^T NT  C_LOWER', lambda: ops.gemm(A, K, tb=True, flags=ops.GEMM_C_LOWER), M * M * n),
    ('Kb=W^T Ab  TN  A_UPPER', lambda: ops.gemm(W, A, ta=True, flags=ops.GEMM_A_UPPER), M * M * n),
    ('plain NN   full       ', lambda: ops.gemm(W, K), 2 * M * M * n),
    ('NT wide  W * Bt^T full', lambda: ops.gemm(W, Bt, tb=True), 2 * M * M * n),
    ('TT? no: TN wide full  ', lambda: ops.gemm(W, K, ta=True), 2 * M * M * n),
    ('NT long-K full nosplit', lambda: ops.gemm(A, K, tb=True, flags=ops.GEMM_NO_SPLITK), 2 * M * M * n),
    ('NT long-K full split  ', lambda: ops.gemm(A, K, tb=True), 2 * M * M * n),
    ('TN long-K full (k-maj)', lambda: ops.gemm(At, Bt, ta=True), 2 * M * M * n),
    ('TN long-K C_LOWER kmaj', lambda: ops.gemm(At, Bt, ta=True, flags=ops.GEMM_C_LOWER), M * M * n),
    ('transpose M x n -> n x M', lambda: A.t().contiguous(), 0),
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
