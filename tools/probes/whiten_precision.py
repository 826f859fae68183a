#!/usr/bin/env python3
"""Where does the float32 whitened projection A = L^-1 Kzx lose accuracy at the headline shape (M = 1024, last layer,
2-D inputs, kappa(Kzz + 1e-4 I) ~ 1e6)?  The reference solves L A = Kzx in float64 and rounds A to float32 once
(gpytorch VariationalStrategy); this path multiplies by W = L^-1.  Compares, against the float64 solve:
  f32     : A = f32(W) Kzx                 (one exact-f32 MFMA GEMM, round 1)
  split   : A = W_hi Kzx + W_lo Kzx        (W64 = W_hi + W_lo in two float32 terms; float32 accumulation)
  f64     : A = W64 Kzx64                  (float64 MFMA GEMM), rounded once
and the posterior mean A^T m / variance that follow.  Run on the GPU box: python tools/probes/whiten_precision.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-precip_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from nsgp import ops  # noqa: E402

torch.manual_seed(0)
M, n, D = 1024, int(os.environ.get('NCOLS', 8192)), 2
g = torch.Generator().manual_seed(3)
Z = torch.randn(M, D, generator=g)
x = 1.2 * torch.randn(n, D, generator=g)
ls = torch.tensor([[0.75, 0.9]])
os_ = torch.tensor([0.8])
m = 0.5 * torch.randn(M, generator=g)
Lq = torch.tril(0.05 * torch.randn(M, M, generator=g)) + 0.6 * torch.eye(M)
dev = 'cuda'
Zd, xd = Z.double(), x.double()
Kzz = ops.rbf_build(Zd.to(dev), Zd.to(dev), ls.double().to(dev), os_.double().to(dev), diag_add=1e-4)
ev = torch.linalg.eigvalsh(Kzz[0].cpu())
print('kappa(Kzz + 1e-4 I) = %.3g' % float(ev[-1] / ev[0]))
L, info = ops.potrf(Kzz)
W64 = ops.trtri(L)[0]
Kzx32 = ops.rbf_build(Z.to(dev), x.to(dev), ls.to(dev), os_.to(dev))[0]          # float32, as the reference builds it
# reference: float64 triangular solve of the float32 Kzx, rounded to float32
A_ref = torch.linalg.solve_triangular(L[0].cpu(), Kzx32.double().cpu(), upper=False)
mean_ref = A_ref.T @ m.double()
C_ref = torch.tril(Lq.double()).T @ A_ref
var_ref = float(os_) + 1e-4 + (C_ref ** 2).sum(0) - (A_ref ** 2).sum(0)


def report(tag, A):
    A = A.double().cpu()
    mean = A.T @ m.double()
    C = torch.tril(Lq.double()).T @ A
    var = float(os_) + 1e-4 + (C ** 2).sum(0) - (A ** 2).sum(0)
    print('%-6s  A max-norm rel %.3g   mean max-norm rel %.3g  (2-norm rel %.3g)   var max-norm rel %.3g' % (
        tag, float((A - A_ref).abs().max() / A_ref.abs().max()), float((mean - mean_ref).abs().max() / mean_ref.abs().max()),
        float((mean - mean_ref).norm() / mean_ref.norm()), float((var - var_ref).abs().max() / var_ref.abs().max())))


W_hi = W64.float()
W_lo = (W64 - W_hi.double()).float()
report('round', A_ref.float())                                                  # the reference's own rounding of A
report('f32', ops.gemm(W_hi, Kzx32, flags=ops.GEMM_A_LOWER))
A_split = ops.gemm(W_hi, Kzx32, flags=ops.GEMM_A_LOWER)
ops.gemm(W_lo, Kzx32, flags=ops.GEMM_A_LOWER, beta=1.0, out=A_split)
report('split', A_split)
report('f64', ops.gemm(W64, Kzx32.double(), flags=ops.GEMM_A_LOWER).float())
print('max |W| = %.3g, max |W_lo| = %.3g, max |A| = %.3g' % (float(W64.abs().max()), float(W_lo.abs().max()), float(A_ref.abs().max())))
