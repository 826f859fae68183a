#!/usr/bin/env python3
"""Micro-benchmark of the MFMA GEMM on the (M x M x n) products of one SVGP layer step (BASELINE B4: M=1024;
last layer n = S*B = 40960, one GP; hidden layer n = 4096, two GPs).  Interleaved rounds in ONE process
(cdna_hip_programming.md rule 24): every case runs once per round, ROUNDS rounds; prints the median and the minimum
time and the TFLOP/s of the median (algorithmic flops: triangular halves counted once).

    python tools/gemm_bench.py                 # last-layer shapes
    NCOLS=4096 BATCH=2 python tools/gemm_bench.py
    ENVAB=NSGP_GEMM_XCD_GROUP python tools/gemm_bench.py     # A/B an environment switch of the library (0 / 1)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
from nsgp import ops  # noqa: E402

M, n = int(os.environ.get('MROWS', 1024)), int(os.environ.get('NCOLS', 40960))
b = int(os.environ.get('BATCH', 1))
rounds = int(os.environ.get('ROUNDS', 15))
g = torch.Generator().manual_seed(0)
W = torch.tril(torch.randn(b, M, M, generator=g)).cuda()
Lq = torch.tril(torch.randn(b, M, M, generator=g)).cuda()
K = torch.randn(b, M, n, generator=g).cuda()
A = torch.randn(b, M, n, generator=g).cuda()
m = torch.randn(b, M, generator=g).cuda()
gm, gv = torch.randn(b, n, generator=g).cuda(), torch.randn(b, n, generator=g).cuda()
os_ = torch.ones(b).cuda()
Cc = torch.randn(b, M, n, generator=g).cuda()
tri = M * M * n * b

W64 = W.double()
cases = [
    ('fwd  A=W K, C=Lq^T A (+colstats)', lambda: ops.svgp_project(W, K, Lq, m, os_), 2 * tri),
    ('fwd  same, A accumulated in f64 ', lambda: ops.svgp_project(W, K, Lq, m, os_, W64f=W64), 2 * tri),
    ('bwd  Abar (epi 2) + Lqbar (ksc)  ', lambda: ops.svgp_project_bwd(Lq, m, A, Cc, gm, gv), 2 * tri),
    ('bwd  Kzxbar = W^T Abar  A_UPPER  ', lambda: ops.gemm(W, A, ta=True, flags=ops.GEMM_A_UPPER), tri),
    ('bwd  Wbar = tril(Abar Kzx^T)     ', lambda: ops.gemm(A, K, tb=True, flags=ops.GEMM_C_LOWER), tri),
    ('     A=W K   A_LOWER plain       ', lambda: ops.gemm(W, K, flags=ops.GEMM_A_LOWER), tri),
    ('     dense NN (full W)           ', lambda: ops.gemm(W, K), 2 * tri),
    ('     dense NT long-K, split      ', lambda: ops.gemm(A, K, tb=True), 2 * tri),
]
only = os.environ.get('ONLY')
if only:
    cases = [c for c in cases if only in c[0]]
envab = os.environ.get('ENVAB')
vals = os.environ.get('ENVVALS', '0,1').split(',')
variants = [('', None)] if not envab else [(f' [{envab}={v}]', v) for v in vals]
times = {(c[0], v[0]): [] for c in cases for v in variants}
for r in range(rounds + 2):
    for name, fn, flops in cases:
        for tag, val in variants:
            if val is not None:
                os.environ[envab] = val
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            if r >= 2:
                times[(name, tag)].append(e0.elapsed_time(e1))
print(f'M={M} n={n} batch={b} rounds={rounds}')
for name, fn, flops in cases:
    for tag, _ in variants:
        t = sorted(times[(name, tag)])
        med, mn = t[len(t) // 2], t[0]
        print(f'{name}{tag}: median {med * 1e3:8.1f} us  min {mn * 1e3:8.1f} us  {flops / med / 1e9:7.1f} TFLOP/s', flush=True)
