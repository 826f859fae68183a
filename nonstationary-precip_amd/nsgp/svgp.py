"""Whitened SVGP layer marginals as fused autograd nodes with hand-derived backward passes.

What gpytorch's VariationalStrategy.forward does per layer (driven by
models/dgps.py:48-51 through DeepGPLayer.__call__):

    Kzz = K(Z,Z) + jitter I ; L = chol(Kzz.double()) ; A = L^-1 Kzx (double, cast back)
    mean = A^T m (+ mu(x), added by the caller) ; var = kxx + 1e-4 + colsum(A o ((Lq Lq^T - I) A))

MI355X formulation (same math, no per-sample redundancy, everything on the matrix cores):
    W   = chol(Kzz)^-1              float64 potrf + trtri  -- parameter-only work, done ONCE per step
                                    for every layer of the model in one batched chain (WhitenFn)
    A   = W Kzx                     f32 MFMA GEMM, lower-triangular W skips half the K-tiles
    C   = Lq^T A                    f32 MFMA GEMM, upper-triangular operand
    var = base + colsum(C o C) - colsum(A o A)      reduced in the two GEMMs' epilogues (A, C not re-read)
The backward is 4 more (M x M x n) GEMMs per layer (SVGPLayerFn; the elementwise parts of Abar / Lqbar are
folded into a GEMM epilogue and an operand loader) + one batched M^3 Cholesky adjoint
(WhitenFn); every identity is checked against torch autograd of the oracle in tests/test_gpu_svgp.py.
"""
import torch

from . import ops
from .ops import GEMM_A_LOWER, GEMM_A_UPPER, GEMM_B_LOWER, GEMM_B_UPPER, GEMM_C_LOWER, GEMM_C_NOFILL, GEMM_C_HALFDIAG

VAR_JITTER = 1e-4          # data_data_covar.add_jitter(1e-4) in VariationalStrategy.forward


class WhitenFn(torch.autograd.Function):
    """W[g] = chol(os_g RBF(Z_g, Z_g; ls_g) + jitter I)^-1 in float64 for a list of GP groups.

    inputs: jitter, chol_bwd_f64, then (Z_i:(b_i,M,D_i), ls_i:(b_i,D_i), os_i:(b_i,)) per group, all with
    the same M.  Outputs: W64_i:(b_i, M, M) float64 per group, info:(sum b_i,), then the parameters again
    (Z_i, ls_i, os_i pass-throughs).  A layer that takes its kernel parameters from the pass-throughs sends its Kzx-path
    gradients back through this node, where they join the Kzz-path gradients in one multi-tensor add instead of six
    accumulation launches of the autograd engine.  The Gram matrices are built
    per group (the input dimension differs between layers), then ONE batched potrf + trtri chain
    factors them all; the adjoint  Kbar = -1/2 W^T (Phi(B) + Phi(B)^T) W,  B = tril(Wbar) W^T  is three
    batched MFMA GEMMs.
    """

    @staticmethod
    def forward(ctx, jitter, chol_bwd_f64, out_dtype, *params):
        groups = [params[i:i + 3] for i in range(0, len(params), 3)]
        z64, off = [], 0
        M = groups[0][0].shape[-2]
        K = torch.empty((sum(g[0].shape[0] for g in groups), M, M), dtype=torch.float64, device=groups[0][0].device)
        # float64 copies of all (small) parameters with ONE multi-tensor copy instead of a cast launch per tensor
        flat = [t for g in groups for t in g]
        if all(t.dtype == torch.float64 for t in flat):
            f64 = list(flat)
        else:
            f64 = [torch.empty(t.shape, dtype=torch.float64, device=t.device) for t in flat]
            torch._foreach_copy_(f64, [t.detach() for t in flat])
        for gi, (Z, ls, os_) in enumerate(groups):          # Gram matrices straight into the batched buffer
            Zd, lsd, osd = f64[3 * gi:3 * gi + 3]
            z64.append((Zd, lsd, osd))
            ops.rbf_build(Zd, Zd, lsd, osd, diag_add=jitter, out=K[off:off + Z.shape[0]])
            off += Z.shape[0]
        # K is consumed; the factor itself is never needed.  A float32 model also gets its float32 W from the same launches
        if out_dtype == torch.float32:
            W64, info, W32 = ops.potrf_trtri_(K, want_f32=True)
        else:
            (W64, info), W32 = ops.potrf_trtri_(K), None
        ctx.save_for_backward(W64, *[t for g in z64 for t in g])
        ctx.sizes = [g[0].shape[0] for g in groups]
        ctx.dtypes = [g[0].dtype for g in groups]
        ctx.chol_bwd_f64 = chol_bwd_f64
        ctx.mark_non_differentiable(info)
        ctx.set_materialize_grads(False)         # no zero-fill launches for outputs nobody differentiated
        # the layers consume W in their own dtype: ONE cast of the batched result here instead of one per layer
        Wout = W64 if out_dtype in (None, torch.float64) else (W32 if W32 is not None else ops.cast(W64, out_dtype))
        outs, outs64, off = [], [], 0
        for b in ctx.sizes:                      # one output per group (views of the batched result)
            outs.append(Wout[off:off + b])
            outs64.append(W64[off:off + b].detach() if Wout is not W64 else Wout[off:off + b].detach())
            off += b
        # the float64 W rides along (non-differentiable): a float32 layer accumulates A = W Kzx in float64 with it
        ctx.mark_non_differentiable(*outs64)
        return (*outs, info, *[t.view_as(t) for t in flat], *outs64)

    @staticmethod
    def backward(ctx, *gouts):
        W64, *flat = ctx.saved_tensors
        ng = len(ctx.sizes)
        gpass = gouts[ng + 1:ng + 1 + 3 * ng]                # gradients that came in through the pass-throughs
        if ng == 1 and gouts[0] is not None and gouts[0].dtype == W64.dtype:
            Wbar = gouts[0]
        else:                                    # float64 batched Wbar: cast + placement in one multi-tensor copy
            Wbar = torch.empty_like(W64)
            dst, src, off = [], [], 0
            for gi, b in enumerate(ctx.sizes):
                if gouts[gi] is None:
                    Wbar[off:off + b].zero_()
                else:
                    dst.append(Wbar[off:off + b])
                    src.append(gouts[gi])
                off += b
            if dst and all(g.dtype == torch.float32 and g.is_cuda for g in src) and Wbar.dtype == torch.float64:
                for d_, s_ in zip(dst, src):             # one cast launch per layer: the multi-tensor copy of three 1024^2
                    ops.cast(s_, torch.float64, out=d_)  # matrices took 27 us (few, large tensors), these take 5 + 8
            elif dst:
                torch._foreach_copy_(dst, src)
        if ctx.chol_bwd_f64:
            Wb, Wc = Wbar.contiguous(), W64
        else:
            Wb, Wc = ops.cast(Wbar.contiguous(), torch.float32), ops.cast(W64, torch.float32)
        # Kbar = -1/2 W^T (Phi + Phi^T) W with Phi = tril(Wbar W^T), diagonal halved.  The kernel backward below sums the
        # row and the column side of its input (sym=True), i.e. it sees Kbar + Kbar^T only, so G = -W^T Phi W stands in for
        # Kbar (G + G^T = 2 Kbar): every product keeps its triangular structure -- lower triangle of a lower x upper
        # product, lower x lower (lower result), upper x lower -- 2/3 of a dense M^3 product in total instead of 4/3.
        # The strict upper triangles of Phi and Phi W are never written and never read (LOWER operand flags mask them).
        Phi = ops.gemm(Wb, Wc, tb=True, flags=GEMM_A_LOWER | GEMM_B_UPPER | GEMM_C_LOWER | GEMM_C_NOFILL | GEMM_C_HALFDIAG)
        T = ops.gemm(Phi, Wc, flags=GEMM_A_LOWER | GEMM_B_LOWER | GEMM_C_LOWER | GEMM_C_NOFILL)
        Kbar = ops.gemm(Wc, T, ta=True, alpha=-1.0, flags=GEMM_A_UPPER | GEMM_B_LOWER)
        grads, off = [None, None], 0
        for gi, b in enumerate(ctx.sizes):
            Zd, lsd, osd = flat[3 * gi:3 * gi + 3]
            Kb = Kbar[off:off + b]
            off += b
            if Kb.dtype != torch.float64:
                Zk, lsk, osk = Zd.float(), lsd.float(), osd.float()
            else:
                Zk, lsk, osk = Zd, lsd, osd
            gZ, _, gls, gos = ops.rbf_build_bwd(Zk, Zk, lsk, osk, Kb.contiguous(), sym=True)   # both sides summed
            grads += [gZ, gls, gos]
        # back to the parameters' dtypes with ONE multi-tensor copy
        outs = [g if g.dtype == ctx.dtypes[i // 3] else torch.empty(g.shape, dtype=ctx.dtypes[i // 3], device=g.device)
                for i, g in enumerate(grads[2:])]
        src = [g for g, o in zip(grads[2:], outs) if o is not g]
        dst = [o for g, o in zip(grads[2:], outs) if o is not g]
        if dst:
            torch._foreach_copy_(dst, src)
        extra = [(o, g) for o, g in zip(outs, gpass) if g is not None]
        if extra:
            torch._foreach_add_([o for o, _ in extra], [g.reshape(o.shape) for o, g in extra])
        return (None, None, None, *outs)


def whiten(groups, jitter=1e-4, chol_bwd_f64=True, passthrough=False, out_dtype=None, with_f64=False):
    """groups: list of (Z:(b,M,D), ls:(b,D), os:(b,)).  Returns (list of W:(b,M,M) per group -- float64 unless
    `out_dtype` asks for the layers' dtype --, info); with
    passthrough=True also the list of (Z, ls, os) pass-through triples a layer should build its Kzx from (WhitenFn);
    with_f64=True appends the list of float64 W per group (non-differentiable companions of the float32 W: the forward
    projection of a float32 layer accumulates in float64 with them, settings.whiten_matmul_f64)."""
    flat = [t for g in groups for t in g]
    res = WhitenFn.apply(float(jitter), bool(chol_bwd_f64), out_dtype, *flat)
    ng = len(groups)
    Ws, info, rest, W64s = res[:ng], res[ng], res[ng + 1:ng + 1 + 3 * ng], res[ng + 1 + 3 * ng:]
    from .gp import settings
    if settings.check_variational_cholesky.on():
        bad = info.nonzero()
        if bad.numel():                                   # host sync: debugging aid only
            from .gp.utils.cholesky import NotPSDError
            b = int(bad[0, 0])
            raise NotPSDError(f'Kzz + {jitter:g} I of GP {b} (of {info.numel()} in the whitening chain) is not positive '
                              f'definite: leading minor {int(info[b])} failed')
    out = [list(Ws), info]
    if passthrough:
        out.append([tuple(rest[3 * i:3 * i + 3]) for i in range(ng)])
    if with_f64:
        out.append(list(W64s))
    return tuple(out)


class SVGPLayerFn(torch.autograd.Function):
    """(x, Z, ls, os, m, Lq, W64, mean_w, mean_c) -> (mean:(b,n), var:(b,n))

    x:(n,D) shared by the b output GPs, or (b,n,D);  Z:(b,M,D)  ls:(b,D)  os:(b,)  m:(b,M)  Lq:(b,M,M)
    (only the lower triangle of Lq is used, like CholeskyVariationalDistribution.forward);
    W64:(b,M,M) = chol(Kzz)^-1 from WhitenFn (float64, or already in x's dtype).  The dependence of Kzz on (Z, ls, os) flows
    through W64's gradient; this node differentiates the Kzx path.
    mean_w:(D,) or (b,D) and mean_c:(1,) or (b,) are the weights / constant of an affine prior mean function
    (gpytorch LinearMean / ConstantMean, models/dgps.py:40-43); either may be None.  The prior mean is added in the
    kernel that assembles the column statistics and its gradients come out of the `rowdot` launch of the backward,
    so the mean module costs no launches of its own.
    """

    @staticmethod
    def forward(ctx, x, Z, ls, os_, m, Lq, W64, mean_w, mean_c, W64f=None, kzx_f64=False):
        W = W64 if W64.dtype == x.dtype else ops.cast(W64, x.dtype)
        if W64f is None and W64.dtype == torch.float64 and x.dtype == torch.float32:
            W64f = W64
        from .gp import settings
        if x.dtype != torch.float32 or not (settings.whiten_matmul_f64.on() or settings.forward_precision.value() == 'bf16_all'):
            W64f = None
        affine = None if (mean_w is None and mean_c is None) else (x, mean_w, mean_c)
        fp = settings.forward_precision.value()
        # Kzx never materialised in the forward pass: its tiles are generated inside the loader of A = W Kzx (float32 layers
        # with the float64-accumulating product, whole tiles); the backward builds it once for Wbar = tril(Abar Kzx^T)
        fuse = fp == 'f32' and settings.fuse_kzx.on() and W64f is not None and \
            ops.svgp_kzx_fusable(W64f, Z, x, x.shape[-2])
        # Which arithmetic for A = W Kzx (float32 layers with the float64 W of the whitening chain):
        #  * the exact int8 digit-plane product (settings.whiten_matmul_i8; M <= 4096, D <= 4): four Kzx planes / 14 plane
        #    products (1e-6 of max|A|), five / 19 for a layer that feeds the next one (kzx_f64: A to float32 rounding);
        #  * int8 off: the float64-accumulating product on the float32 Kzx (round 2) -- on a float64 Kzx for a layer that
        #    feeds the next one (settings.hidden_kzx_f64).
        # And for C = Lq^T A: float32, except for a layer that feeds the next one and sees at most 8192 points -- the first
        # hidden layer of a deep GP (settings.hidden_var_f64) -- where it accumulates in float64 with float64 column-statistic
        # partials: the variance os + colsum(C^2 - A^2) cancels to << os once q(u) has trained and reaches the next layer's
        # inputs through sqrt(var) eps.  Measured after 1000 Adam steps at the headline shape (max-norm, vs the float64
        # oracle; tests/test_gpu_headline_precision.py): output mean 4.4e-6 with it, 5.2e-5 without (the reference's own
        # float32 arithmetic: 4.5e-4); +0.12 ms on a 4.34 ms step.
        Kzx64 = Lq64 = None
        can64 = kzx_f64 and fp == 'f32' and W64f is not None and not fuse
        hv = settings.hidden_var_f64.value()
        var64 = can64 and ((x.shape[-2] <= 8192) if hv == 'auto' else bool(hv))
        use_i8 = (fp in ('f32', 'bf16') and not fuse and W64f is not None and settings.whiten_matmul_i8.on()
                  and Z.shape[-1] <= 4 and Z.shape[-2] <= 4096)
        if can64 and not use_i8:
            src = [x.detach(), Z.detach(), ls.detach(), os_.detach()] + ([Lq.detach()] if var64 else [])
            dst = [torch.empty(t.shape, dtype=torch.float64, device=t.device) for t in src]
            torch._foreach_copy_(dst, src)
            Kzx64 = ops.rbf_build(dst[1], dst[0], dst[2], dst[3])
            Lq64 = dst[4] if var64 else None  # C = Lq^T A accumulates in float64 too (the variance's cancellation)
        elif var64:
            Lq64 = ops.cast(Lq.detach(), torch.float64)
        Kzx = None if (fuse or Kzx64 is not None or use_i8) else ops.rbf_build(Z, x, ls, os_)           # (b,M,n)
        # int8 path: the plane-build kernel also writes the float32 Kzx the BACKWARD needs (Wbar = tril(Abar Kzx^T)) when there
        # is going to be one -- its own build launch (28 us at the headline's last layer) disappears
        kzx_keep = [] if (use_i8 and any(ctx.needs_input_grad)) else None
        if x.dtype == torch.float32 and fp in ('bf16', 'bf16_all') and Z.shape[-2] % 8 == 0:
            # BASELINE configs[4]'s "bf16 forward": C = Lq^T A on the bf16 matrix cores; 'bf16_all' also A = W Kzx
            A, C, mean, var = ops.svgp_project_bf16(W, Kzx, Lq, m, os_, base_add=VAR_JITTER, affine=affine, W64f=W64f,
                                                    kernel_inputs=(Z, x, ls, os_) if fp == 'bf16_all' else None,
                                                    i8_inputs=(Z, x, ls, os_) if (use_i8 and fp == 'bf16') else None,
                                                    i8_kzx_out=kzx_keep if (use_i8 and fp == 'bf16') else None)
        else:
            A, C, mean, var = ops.svgp_project(W, Kzx, Lq, m, os_, base_add=VAR_JITTER, affine=affine, W64f=W64f,
                                               kernel_inputs=(Z, x, ls, os_) if fuse else None, Kzx64=Kzx64, Lq64=Lq64,
                                               i8_inputs=(Z, x, ls, os_) if use_i8 else None,
                                               i8_planes=5 if kzx_f64 else 4, i8_kzx_out=kzx_keep)      # 2 GEMMs
        if kzx_keep:
            Kzx = kzx_keep[0]
        ctx.save_for_backward(x, Z, ls, os_, m, Lq, W, Kzx, A, C, mean_w, mean_c)
        ctx.w_dtype = W64.dtype
        return mean, var

    @staticmethod
    def backward(ctx, gmean, gvar):
        x, Z, ls, os_, m, Lq, W, Kzx, A, C, mean_w, mean_c = ctx.saved_tensors
        if Kzx is None:                                  # fused / float64-Kzx forward: built here, once, for the Wbar product
            Kzx = ops.rbf_build(Z, x, ls, os_)
        gmean = gmean.contiguous()
        affine = None if (mean_w is None and mean_c is None) else (x, mean_w, mean_c)
        Abar, Lqbar, mbar, basebar, wbar, cbar = ops.svgp_project_bwd(Lq, m, A, C, gmean, gvar.contiguous(),
                                                                      affine=affine)
        Kzxbar = ops.gemm(W, Abar, ta=True, flags=GEMM_A_UPPER)                  # W^T Abar
        Wbar = ops.gemm(Abar, Kzx, tb=True, flags=GEMM_C_LOWER)                  # tril(Abar Kzx^T)
        need_x = ctx.needs_input_grad[0]
        gZ, gx, gls, gos = ops.rbf_build_bwd(Z, x, ls, os_, Kzxbar, need_x1=True, need_x2=need_x)
        gos = gos + basebar
        if need_x:
            if mean_w is not None:                       # d(prior mean)/dx = w  (deeper layers of a tied stack only)
                gx = gx + gmean.unsqueeze(-1) * mean_w.reshape(-1, 1, x.shape[-1])
            if x.dim() == 2:
                gx = gx[0] if gx.shape[0] == 1 else gx.sum(0)
        return (gx if need_x else None, gZ, gls.reshape(ls.shape), gos.reshape(os_.shape), mbar, Lqbar,
                Wbar if Wbar.dtype == ctx.w_dtype else ops.cast(Wbar, ctx.w_dtype),
                None if mean_w is None else wbar.reshape(mean_w.shape),
                None if mean_c is None else cbar.reshape(mean_c.shape), None, None)


def svgp_marginal(x, Z, ls, os_, m, Lq, jitter=1e-4, chol_bwd_f64=True, W64=None, mean_w=None, mean_c=None, W64f=None,
                  kzx_f64=False):
    """mean and variance of q(f) at x for b whitened SVGPs; the mean excludes the prior mean function unless its
    affine parameters are passed (mean_w: LinearMean weights (D,) / (b,D), mean_c: constant or bias (1,) / (b,)).
    Returns (mean, var, info); pass W64 (from `whiten`) to share one factorisation chain across layers."""
    info = None
    if W64 is None:
        (W64,), info, ((Z, ls, os_),), (W64f,) = whiten([(Z, ls, os_)], jitter, chol_bwd_f64, passthrough=True,
                                                        out_dtype=x.dtype, with_f64=True)
    mean, var = SVGPLayerFn.apply(x, Z, ls, os_, m, Lq, W64, mean_w, mean_c, W64f, kzx_f64)
    return mean, var, info
