#!/usr/bin/env python3
"""Micro-benchmark of the batched Kzz -> potrf -> trtri chain (3 x 1024^2 float64, as in one DSVI step)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
import torch  # noqa: E402
from nsgp import ops  # noqa: E402

M, b = int(os.environ.get('M', 1024)), int(os.environ.get('BATCH', 3))
g = torch.Generator().manual_seed(0)
Z = torch.randn(b, M, 3, generator=g, dtype=torch.float64).cuda()
ls = torch.full((b, 3), 0.7, dtype=torch.float64).cuda()
os_ = torch.full((b,), 0.7, dtype=torch.float64).cuda()
K = ops.rbf_build(Z, Z, ls, os_, diag_add=1e-4)


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


L, info = ops.potrf(K)
print('info', info.tolist())
print(f'potrf  {timeit(lambda: ops.potrf(K)):8.1f} us')
print(f'trtri  {timeit(lambda: ops.trtri(L)):8.1f} us')
# Cholesky adjoint of the whitening chain (three float64 M x M x M triangular products, WhitenFn.backward)
W = ops.trtri(L)
Wbar = torch.tril(torch.randn(b, M, M, generator=g, dtype=torch.float64)).cuda()


def chol_bwd_dense():
    """round-1 form: Kbar = -1/2 W^T (Phi + Phi^T) W with the symmetrised S materialised (4/3 of a dense product)."""
    Bm = ops.gemm(Wbar, W, tb=True, flags=ops.GEMM_A_LOWER | ops.GEMM_B_UPPER)
    S = ops.chol_bwd_phi_sym(Bm)
    T = ops.gemm(S, W, flags=ops.GEMM_B_LOWER)
    return ops.gemm(W, T, ta=True, alpha=-0.5, flags=ops.GEMM_A_UPPER)


def chol_bwd():
    """WhitenFn.backward: G = -W^T Phi W, every product triangular (2/3 of a dense product)."""
    fl = ops.GEMM_C_LOWER | ops.GEMM_C_NOFILL
    Phi = ops.gemm(Wbar, W, tb=True, flags=ops.GEMM_A_LOWER | ops.GEMM_B_UPPER | fl)
    ops.scale_diag_(Phi, 0.5)
    T = ops.gemm(Phi, W, flags=ops.GEMM_A_LOWER | ops.GEMM_B_LOWER | fl)
    return ops.gemm(W, T, ta=True, alpha=-1.0, flags=ops.GEMM_A_UPPER | ops.GEMM_B_LOWER)


G, Kb = chol_bwd(), chol_bwd_dense()
print('G + G^T == 2 Kbar:', float(((G + G.transpose(-1, -2)) - 2 * Kb).abs().max() / Kb.abs().max()))
print(f'chol adjoint, symmetrised S (round 1)  {timeit(chol_bwd_dense):8.1f} us')
print(f'chol adjoint, triangular products      {timeit(chol_bwd):8.1f} us')
