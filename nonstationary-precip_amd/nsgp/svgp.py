"""Whitened SVGP layer marginals as ONE autograd node with a hand-derived backward.

What gpytorch's VariationalStrategy.forward does per layer (driven by
models/dgps.py:48-51 through DeepGPLayer.__call__):

    Kzz = K(Z,Z) + jitter I ; L = chol(Kzz.double()) ; A = L^-1 Kzx (double, cast back)
    mean = A^T m (+ mu(x), added by the caller) ; var = kxx + 1e-4 + colsum(A o ((Lq Lq^T - I) A))

MI355X formulation (same math, no per-sample redundancy, everything on the matrix cores):
    W   = chol(Kzz)^-1              float64 potrf + trtri, once per layer per step (M^3 work)
    A   = W Kzx                     f32 MFMA GEMM, lower-triangular W skips half the K-tiles
    C   = Lq^T A                    f32 MFMA GEMM, upper-triangular operand
    var = base + colsum(C o C) - colsum(A o A)
The backward is 4 more (M x M x n) GEMMs + the M^3 Cholesky adjoint; every identity is checked
against torch autograd of the oracle in tests/test_gpu_svgp.py.
"""
import torch

from . import ops
from .ops import GEMM_A_LOWER, GEMM_A_UPPER, GEMM_B_LOWER, GEMM_B_UPPER, GEMM_C_LOWER

VAR_JITTER = 1e-4          # data_data_covar.add_jitter(1e-4) in VariationalStrategy.forward


class SVGPLayerFn(torch.autograd.Function):
    """(x, Z, ls, os, m, Lq) -> (mean_without_prior_mean:(b,n), var:(b,n), info:(b,))

    x:(n,D) shared by the b output GPs, or (b,n,D);  Z:(b,M,D)  ls:(b,D)  os:(b,)  m:(b,M)  Lq:(b,M,M)
    (only the lower triangle of Lq is used, like CholeskyVariationalDistribution.forward).
    """

    @staticmethod
    def forward(ctx, x, Z, ls, os_, m, Lq, jitter, chol_bwd_f64):
        work = x.dtype
        Z64, ls64, os64 = Z.double(), ls.double(), os_.double()
        Kzz = ops.rbf_build(Z64, Z64, ls64, os64, diag_add=jitter)              # (b,M,M) f64
        L, info = ops.potrf(Kzz, overwrite=True)
        W64 = ops.trtri(L)
        W = ops.cast(W64, work)
        Kzx = ops.rbf_build(Z, x, ls, os_)                                       # (b,M,n)
        A = ops.gemm(W, Kzx, flags=GEMM_A_LOWER)
        C = ops.gemm(Lq, A, ta=True, flags=GEMM_A_UPPER)
        mean, var = ops.svgp_colstats(A, C, m, os_ + VAR_JITTER)
        ctx.save_for_backward(x, Z, ls, os_, m, Lq, W64, W, Kzx, A, C)
        ctx.chol_bwd_f64 = chol_bwd_f64
        ctx.mark_non_differentiable(info)
        return mean, var, info

    @staticmethod
    def backward(ctx, gmean, gvar, _ginfo):
        x, Z, ls, os_, m, Lq, W64, W, Kzx, A, C = ctx.saved_tensors
        gmean, gvar = gmean.contiguous(), gvar.contiguous()
        Abar, C2, mbar = ops.svgp_colstats_bwd(A, C, m, gmean, gvar)
        ops.gemm(Lq, C2, flags=GEMM_A_LOWER, beta=1.0, out=Abar)                # Abar += Lq C2
        Lqbar = ops.gemm(A, C2, tb=True, flags=GEMM_C_LOWER)                     # tril(A C2^T)
        Kzxbar = ops.gemm(W, Abar, ta=True, flags=GEMM_A_UPPER)                  # W^T Abar
        Wbar = ops.gemm(Abar, Kzx, tb=True, flags=GEMM_C_LOWER)                  # tril(Abar Kzx^T)
        # Cholesky-inverse adjoint:  Kzz_bar = -1/2 W^T (Phi(B) + Phi(B)^T) W,  B = tril(Wbar) W^T
        if ctx.chol_bwd_f64 or W.dtype == torch.float64:
            Wb, Wc = ops.cast(Wbar, torch.float64), W64
        else:
            Wb, Wc = Wbar, W
        Bm = ops.gemm(Wb, Wc, tb=True, flags=GEMM_A_LOWER | GEMM_B_UPPER)
        S = ops.chol_bwd_phi_sym(Bm)
        T = ops.gemm(S, Wc, flags=GEMM_B_LOWER)
        Kzzbar = ops.gemm(Wc, T, ta=True, alpha=-0.5, flags=GEMM_A_UPPER)
        need_x = ctx.needs_input_grad[0]
        gZ1, gx, gls1, gos1 = ops.rbf_build_bwd(Z, x, ls, os_, Kzxbar, need_x1=True, need_x2=need_x)
        if Kzzbar.dtype == torch.float64:
            Zk, lsk, osk = Z.double(), ls.double(), os_.double()
        else:
            Zk, lsk, osk = Z, ls, os_
        gZa, gZb, gls2, gos2 = ops.rbf_build_bwd(Zk, Zk, lsk, osk, Kzzbar)
        work = x.dtype
        Zbar = gZ1 + (gZa + gZb).to(work)
        lsbar = gls1 + gls2.to(work)
        osbar = gos1 + gos2.to(work) + gvar.sum(-1)
        if need_x and x.dim() == 2:
            gx = gx.sum(0)
        return (gx if need_x else None, Zbar, lsbar.reshape(ls.shape), osbar.reshape(os_.shape), mbar, Lqbar,
                None, None)


def svgp_marginal(x, Z, ls, os_, m, Lq, jitter=1e-4, chol_bwd_f64=True):
    """mean (without the prior mean function) and variance of q(f) at x for b whitened SVGPs."""
    return SVGPLayerFn.apply(x, Z, ls, os_, m, Lq, float(jitter), bool(chol_bwd_f64))
