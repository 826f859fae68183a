"""GPU parity of the fused whitened-SVGP layer node (nsgp.svgp.SVGPLayerFn) against torch autograd of
the CPU oracle (oracle.svgp.svgp_marginal): marginals and every parameter gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64
from conftest import measured  # noqa: E402

# float32 layer vs float64 oracle at small shapes (kappa(Kzz) ~ 1e4): ~3x the errors measured on MI355X
# measured: values worst diff 2e-5 on |ref| 0.4; gradients <= 3.1e-4 max-norm relative
F32_VALUE_TOL = dict(rtol=1e-4, atol=3e-5)
F32_GRAD_TOL = 1e-3


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _params(b, M, D, n, seed, batched_x):
    g = _g(seed)
    Z = torch.randn(b, M, D, generator=g, dtype=F64)
    ls = torch.rand(b, D, generator=g, dtype=F64) + 0.7
    os_ = torch.rand(b, generator=g, dtype=F64) + 0.5
    m = 0.3 * torch.randn(b, M, generator=g, dtype=F64)
    Lq = torch.tril(0.1 * torch.randn(b, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64)
    x = torch.randn((b, n, D) if batched_x else (n, D), generator=g, dtype=F64)
    gm = torch.randn(b, n, generator=g, dtype=F64)
    gv = torch.randn(b, n, generator=g, dtype=F64)
    return x, Z, ls, os_, m, Lq, gm, gv


def _oracle(x, Z, ls, os_, m, Lq, gm, gv, jitter):
    from oracle import svgp
    ins = [t.clone().requires_grad_() for t in (x, Z, ls, os_, m, Lq)]
    xo, Zo, lso, oso, mo, Lqo = ins
    b = Z.shape[0]
    xin = xo if xo.dim() == 3 else xo.unsqueeze(0).expand(b, *xo.shape)
    p = dict(Z=Zo, lengthscale=lso.unsqueeze(-2), outputscale=oso, m=mo, Lq=Lqo,
             mean=('constant', torch.zeros(b, 1, dtype=F64)))
    mean, var = svgp.svgp_marginal(xin, p, jitter=jitter)
    ((mean * gm).sum() + (var * gv).sum()).backward()
    return mean.detach(), var.detach(), [t.grad for t in ins]


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('b,M,D,n,batched_x', [(2, 50, 3, 315, False), (1, 130, 2, 400, True), (2, 64, 2, 96, True)])
def test_svgp_layer_matches_oracle(dt, b, M, D, n, batched_x):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp.svgp import svgp_marginal
    jitter = 1e-4
    args = _params(b, M, D, n, 40 + M, batched_x)
    mean_r, var_r, grads_r = _oracle(*args, jitter)
    x, Z, ls, os_, m, Lq, gm, gv = args
    cu = [t.to(dt).cuda().requires_grad_() for t in (x, Z, ls, os_, m, Lq)]
    mean, var, info = svgp_marginal(*cu, jitter=jitter)
    assert info.cpu().tolist() == [0] * b
    # float32 layer: W (float64 Cholesky) is rounded to f32 and A = W Kzx runs on f32 MFMA, where the
    # reference rounds A after a float64 solve; with kappa(Kzz) ~ 1e4 here both agree to ~1e-4
    tol = dict(rtol=1e-9, atol=1e-10) if dt == F64 else F32_VALUE_TOL
    assert measured(f'svgp mean {dt} b{b} M{M}', mean, mean_r, **tol)
    assert measured(f'svgp var {dt} b{b} M{M}', var, var_r, **tol)
    ((mean * gm.to(dt).cuda()).sum() + (var * gv.to(dt).cuda()).sum()).backward()
    gtol = dict(rtol=1e-7, atol=1e-8) if dt == F64 else dict(rtol=2e-2, atol=2e-2)
    names = ['x', 'Z', 'ls', 'os', 'm', 'Lq']
    for name, a, r in zip(names, cu, grads_r):
        got = a.grad.cpu().double()
        ref = torch.tril(r) if name == 'Lq' else r
        scale = float(ref.abs().max()) + 1e-30
        err = float((got - ref).abs().max()) / scale
        print(f'[measured] svgp grad {name} {dt} b{b} M{M}: max-norm rel err {err:.3g}')
        assert err < (1e-7 if dt == F64 else F32_GRAD_TOL), (name, err)


def test_svgp_layer_at_init_is_the_prior():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp.svgp import svgp_marginal
    g = _g(50)
    b, M, D, n = 2, 100, 3, 257
    Z = torch.randn(b, M, D, generator=g).cuda()
    x = torch.randn(n, D, generator=g).cuda()
    ls = torch.full((b, D), 0.6931).cuda()
    os_ = torch.full((b,), 0.6931).cuda()
    m = torch.zeros(b, M).cuda()
    Lq = torch.eye(M).expand(b, M, M).contiguous().cuda()
    mean, var, _ = svgp_marginal(x, Z, ls, os_, m, Lq)
    assert float(mean.abs().max()) == 0.0
    assert torch.allclose(var, (os_ + 1e-4)[:, None].expand(b, n), atol=2e-5)


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('b,M,n', [(1, 1, 1), (2, 63, 65), (1, 130, 700), (3, 256, 1000), (1, 1024, 4096)])
def test_fused_projection_ops_match_dense_formulas(dt, b, M, n):
    """nsgp_svgp_tri_gemm_colstats / colstats_finalize / abar / lqbar / rowdot (the GEMMs with fused epilogues
    behind SVGPLayerFn) against the dense float64 formulas they restate, ragged and tile-aligned sizes."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp import ops
    g = _g(1000 + M + n)
    W = torch.tril(torch.randn(b, M, M, generator=g, dtype=F64)) / max(M, 1) ** 0.5
    Lq = torch.tril(0.3 * torch.randn(b, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64)
    K = torch.randn(b, M, n, generator=g, dtype=F64)
    m = torch.randn(b, M, generator=g, dtype=F64)
    base = torch.rand(b, generator=g, dtype=F64) + 0.5
    gm = torch.randn(b, n, generator=g, dtype=F64)
    gv = torch.randn(b, n, generator=g, dtype=F64)
    dev = lambda t: t.to(dt).cuda()
    # garbage in the strict upper triangles must be ignored (only the lower triangles are operands)
    junk = torch.triu(torch.full((M, M), 7.0, dtype=F64), 1)
    A, C, mean, var = ops.svgp_project(dev(W + junk), dev(K), dev(Lq + junk), dev(m), dev(base))
    A_ref = W @ K
    C_ref = Lq.transpose(-1, -2) @ A_ref
    mean_ref = torch.einsum('bkj,bk->bj', A_ref, m)
    var_ref = base[:, None] + (C_ref ** 2).sum(1) - (A_ref ** 2).sum(1)
    tol = dict(rtol=1e-10, atol=1e-10) if dt == F64 else dict(rtol=2e-4, atol=2e-4 * max(1.0, M ** 0.5))
    assert torch.allclose(A.cpu().double(), A_ref, **tol)
    assert torch.allclose(C.cpu().double(), C_ref, **tol)
    assert torch.allclose(mean.cpu().double(), mean_ref, **tol)
    assert torch.allclose(var.cpu().double(), var_ref, **tol)
    Abar, Lqbar, mbar, basebar, wbar, cbar = ops.svgp_project_bwd(dev(Lq + junk), dev(m), A, C, dev(gm), dev(gv))
    assert wbar is None and cbar is None
    A_, C_ = A.cpu().double(), C.cpu().double()           # adjoints evaluated at the device's own A, C
    Abar_ref = 2 * (Lq @ C_) * gv[:, None, :] + m[:, :, None] * gm[:, None, :] - 2 * A_ * gv[:, None, :]
    Lqbar_ref = torch.tril(torch.einsum('bkj,bj,blj->bkl', A_, 2 * gv, C_))
    mbar_ref = torch.einsum('bkj,bj->bk', A_, gm)
    tolb = dict(rtol=1e-9, atol=1e-9) if dt == F64 else dict(rtol=5e-4, atol=5e-4 * max(1.0, (n * 1.0) ** 0.5))
    assert torch.allclose(Abar.cpu().double(), Abar_ref, **(tol if dt == F64 else tolb))
    assert torch.allclose(Lqbar.cpu().double(), Lqbar_ref, **tolb)
    assert torch.allclose(mbar.cpu().double(), mbar_ref, **tolb)
    assert float(torch.triu(Lqbar, 1).abs().max()) == 0.0 if M > 1 else True
    assert torch.allclose(basebar.cpu().double(), gv.sum(-1), **tolb)


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('b,M,n,D,xb,shared,has_w,has_c', [
    (2, 40, 333, 3, False, True, True, True),        # models/dgps.py hidden layer: LinearMean shared by the output GPs
    (1, 64, 1000, 2, False, True, False, True),      # last layer: ConstantMean
    (3, 33, 130, 2, True, False, True, True),        # per-GP inputs, weights and bias
    (2, 16, 65, 4, False, False, False, True),       # ConstantMean(batch_shape=[b])
    (2, 16, 65, 1, True, True, True, False),         # LinearMean(bias=False)
])
def test_affine_prior_mean_is_folded_into_the_projection(dt, b, M, n, D, xb, shared, has_w, has_c):
    """colstats_finalize_affine / rowdot_affine: the prior mean x w + c added while the column statistics are assembled,
    `base_add` on the variance, and the gradients of w, c and `base` from the rowdot launch -- against dense formulas,
    with shared and per-batch parameters and shared / per-batch inputs."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp import ops
    g = _g(77 + n + D)
    W = torch.tril(torch.randn(b, M, M, generator=g, dtype=F64)) / M ** 0.5
    Lq = torch.tril(0.3 * torch.randn(b, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64)
    K = torch.randn(b, M, n, generator=g, dtype=F64)
    m = torch.randn(b, M, generator=g, dtype=F64)
    base = torch.rand(b, generator=g, dtype=F64) + 0.5
    x = torch.randn(*((b,) if xb else ()), n, D, generator=g, dtype=F64)
    nb = 1 if shared else b
    w = torch.randn(nb, D, generator=g, dtype=F64) if has_w else None
    c = torch.randn(nb, generator=g, dtype=F64) if has_c else None
    gm = torch.randn(b, n, generator=g, dtype=F64)
    gv = torch.randn(b, n, generator=g, dtype=F64)
    dev = lambda t: None if t is None else t.to(dt).cuda()
    aff = (dev(x), dev(w), dev(c))
    A, C, mean, var = ops.svgp_project(dev(W), dev(K), dev(Lq), dev(m), dev(base), base_add=1e-4, affine=aff)
    A_ref = W @ K
    C_ref = Lq.transpose(-1, -2) @ A_ref
    xe = x if xb else x.unsqueeze(0).expand(b, n, D)
    prior = torch.zeros(b, n, dtype=F64)
    if has_w:
        prior = prior + torch.einsum('bnd,bd->bn', xe, w.expand(b, D))
    if has_c:
        prior = prior + c.expand(b)[:, None]
    mean_ref = torch.einsum('bkj,bk->bj', A_ref, m) + prior
    var_ref = base[:, None] + 1e-4 + (C_ref ** 2).sum(1) - (A_ref ** 2).sum(1)
    tol = dict(rtol=1e-10, atol=1e-10) if dt == F64 else dict(rtol=2e-4, atol=2e-4 * M ** 0.5)
    assert torch.allclose(mean.cpu().double(), mean_ref, **tol)
    assert torch.allclose(var.cpu().double(), var_ref, **tol)
    _, _, mbar, basebar, wbar, cbar = ops.svgp_project_bwd(dev(Lq), dev(m), A, C, dev(gm), dev(gv), affine=aff)
    tolb = dict(rtol=1e-9, atol=1e-9) if dt == F64 else dict(rtol=5e-4, atol=5e-4 * n ** 0.5)
    assert torch.allclose(mbar.cpu().double(), torch.einsum('bkj,bj->bk', A.cpu().double(), gm), **tolb)
    assert torch.allclose(basebar.cpu().double(), gv.sum(-1), **tolb)
    if has_w:
        wref = torch.einsum('bnd,bn->bd', xe, gm)
        assert wbar.shape == (nb, D)
        assert torch.allclose(wbar.cpu().double(), wref.sum(0, keepdim=True) if shared else wref, **tolb)
    else:
        assert wbar is None
    if has_c:
        cref = gm.sum(-1)
        assert cbar.shape == (nb,)
        assert torch.allclose(cbar.cpu().double(), cref.sum(0, keepdim=True) if shared else cref, **tolb)
    else:
        assert cbar is None


def test_non_positive_definite_kzz_is_reported_when_asked():
    """settings.check_variational_cholesky: a Kzz that is not positive definite leaves the default path sync-free (info
    set, NaNs downstream) and makes the checked path raise like gpytorch's psd_safe_cholesky would; duplicate inducing
    points (a singular Kzz) are repaired by the default jitter.  The non-PD case uses a negative output scale so that
    the first pivot is -1: an exactly singular matrix passes or fails on the sign of a rounding error."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp.gp import settings
    from nsgp.gp.utils.cholesky import NotPSDError
    from nsgp.svgp import whiten
    Z = torch.randn(1, 70, 2, dtype=F32, generator=_g(5)).cuda()
    Z[0, 69] = Z[0, 3]
    ls = torch.ones(1, 2, device='cuda')
    os_ = torch.ones(1, device='cuda')
    Ws, info = whiten([(Z, ls, -os_)], jitter=0.0)
    assert int(info[0]) == 1                                 # LAPACK-style: first failing leading minor
    with settings.check_variational_cholesky(True):
        with pytest.raises(NotPSDError, match='not positive definite'):
            whiten([(Z, ls, -os_)], jitter=0.0)
        Ws, info = whiten([(Z, ls, os_)], jitter=1e-4)       # duplicate points: the default jitter repairs them
        assert int(info[0]) == 0


@pytest.mark.parametrize('M,n,batch,D,shared_x', [(256, 512, 2, 3, True), (1024, 4096, 1, 2, True), (128, 128, 3, 4, False),
                                                   (1024, 640, 2, 1, True)])
def test_projection_with_generated_kzx_equals_the_materialised_one(M, n, batch, D, shared_x):
    """nsgp_svgp_kzx_gemm_colstats_f64acc generates the Kzx tiles inside the loader of A = W Kzx with the arithmetic of the
    RBF build kernel: A, C and the column statistics must equal the materialised path bit for bit (128- and 64-row tile
    variants, shared and per-GP inputs, D = 1..4); shapes outside the whole-tile contract are refused by the query."""
    from nsgp import ops
    g = torch.Generator().manual_seed(M + n)
    Z = torch.randn(batch, M, D, generator=g).cuda()
    x = (1.3 * torch.randn(*( (n, D) if shared_x else (batch, n, D)), generator=g)).cuda()
    ls = (0.6 + torch.rand(batch, D, generator=g)).cuda()
    os_ = (0.5 + torch.rand(batch, generator=g)).cuda()
    W64 = torch.tril(torch.randn(batch, M, M, generator=g, dtype=torch.float64) / M ** 0.5).cuda()
    W = W64.float()
    Lq = torch.tril(torch.randn(batch, M, M, generator=g) / M ** 0.5).cuda()
    m = torch.randn(batch, M, generator=g).cuda()
    assert ops.svgp_kzx_fusable(W64, Z, x, n)
    Kzx = ops.rbf_build(Z, x, ls, os_)
    ref = ops.svgp_project(W, Kzx, Lq, m, os_, base_add=1e-4, W64f=W64)
    got = ops.svgp_project(W, None, Lq, m, os_, base_add=1e-4, W64f=W64, kernel_inputs=(Z, x, ls, os_))
    for a, b, name in zip(ref, got, ('A', 'C', 'mean', 'var')):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
    # not whole tiles / too many input dimensions: the caller keeps the materialised path
    assert not ops.svgp_kzx_fusable(W64[:, :M - 4, :M - 4].contiguous(), Z[:, :M - 4].contiguous(), x, n)
    assert not ops.svgp_kzx_fusable(W64, Z, x, n - 4)
    assert not ops.svgp_kzx_fusable(W64, torch.randn(batch, M, 5).cuda(), x, n)


def test_layer_with_generated_kzx_matches_the_materialised_layer():
    """settings.fuse_kzx: the SVGP layer's outputs and every gradient are those of the default (materialised) data flow,
    bit for bit -- the forward values are identical and the backward runs the same launches on a Kzx built there."""
    from nsgp import ops
    from nsgp.gp import settings
    from nsgp.svgp import svgp_marginal
    g = torch.Generator().manual_seed(77)
    b, M, n, D = 2, 256, 512, 3
    x = torch.randn(n, D, generator=g).cuda().requires_grad_()
    Z = torch.randn(b, M, D, generator=g).cuda().requires_grad_()
    ls = (0.8 + torch.rand(b, D, generator=g)).cuda().requires_grad_()
    os_ = (0.7 + torch.rand(b, generator=g)).cuda().requires_grad_()
    m = (0.3 * torch.randn(b, M, generator=g)).cuda().requires_grad_()
    Lq = (torch.eye(M) + 0.05 * torch.tril(torch.randn(b, M, M, generator=g))).cuda().requires_grad_()
    gm, gv = torch.randn(b, n, generator=g).cuda(), torch.randn(b, n, generator=g).cuda()
    outs = []
    for fuse in (False, True):
        for t in (x, Z, ls, os_, m, Lq):
            t.grad = None
        with settings.fuse_kzx(fuse), settings.whiten_matmul_i8(False):      # (both variants on the float64-accumulating product)
            mean, var, _ = svgp_marginal(x, Z, ls, os_, m, Lq)
            ((mean * gm).sum() + (var * gv).sum()).backward()
        outs.append([mean.detach().clone(), var.detach().clone()] + [t.grad.clone() for t in (x, Z, ls, os_, m, Lq)])
    for a, c in zip(*outs):
        assert torch.equal(a, c)


@pytest.mark.parametrize('b,M,n', [(2, 1024, 4096), (1, 200, 333), (3, 128, 64), (1, 1024, 512)])
def test_float64_kzx_projection_of_layers_that_feed_the_next_layer(b, M, n):
    """settings.hidden_kzx_f64 (ops.svgp_project with Kzx64 / Lq64): A = W Kzx from a float64 Kzx, C = Lq^T A, both
    accumulated in float64 with float64 column-statistic partials -- against the same quantities formed in float64 on the
    host from the SAME inputs.  A and C are float32 (rounded once: 6e-8), mean and variance carry no cancellation loss:
    the variance is checked where it is 1e-3 of the prior variance."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp import ops
    g = torch.Generator().manual_seed(90 + M + n)
    D = 3
    Z = torch.randn(b, M, D, generator=g)
    x = torch.randn(n, D, generator=g)
    ls = torch.rand(b, D, generator=g) + 0.8
    os_ = torch.rand(b, generator=g) + 0.5
    m = torch.randn(b, M, generator=g)
    Kzz = (os_.double().reshape(b, 1, 1) * torch.exp(-0.5 * (((Z.double().unsqueeze(2) - Z.double().unsqueeze(1))
                                                              / ls.double().reshape(b, 1, 1, D)) ** 2).sum(-1))
           + 1e-4 * torch.eye(M, dtype=F64))
    W64 = torch.linalg.inv(torch.linalg.cholesky(Kzz))                       # the whitening chain's output
    Kzx64 = os_.double().reshape(b, 1, 1) * torch.exp(-0.5 * (((Z.double().unsqueeze(2) - x.double().reshape(1, 1, n, D))
                                                               / ls.double().reshape(b, 1, 1, D)) ** 2).sum(-1))
    A_ref = W64 @ Kzx64
    # q(u) close to the exact posterior direction: Lq small, so that var = os + colsum(C^2 - A^2) cancels
    Lq = torch.tril(0.05 * torch.randn(b, M, M, generator=g)) + 0.1 * torch.eye(M)
    C_ref = torch.tril(Lq).double().transpose(-1, -2) @ A_ref.float().double()
    mean_ref = (A_ref * m.double().unsqueeze(-1)).sum(1)
    var_ref = os_.double().reshape(b, 1) + 1e-4 + (C_ref ** 2).sum(1) - (A_ref ** 2).sum(1)
    c = lambda t: t.cuda()
    Kd = ops.rbf_build(c(Z).double(), c(x).double(), c(ls).double(), c(os_).double())
    assert float((Kd.cpu() - Kzx64).abs().max()) < 1e-13
    A, C, mean, var = ops.svgp_project(c(W64).float(), None, c(Lq), c(m), c(os_), base_add=1e-4, W64f=c(W64), Kzx64=Kd,
                                       Lq64=c(Lq).double())
    assert A.dtype == torch.float32 and C.dtype == torch.float32
    assert torch.equal(A.cpu(), A_ref.float()) or float((A.cpu().double() - A_ref).abs().max() / A_ref.abs().max()) < 1.2e-7
    assert float((C.cpu().double() - C_ref).abs().max() / C_ref.abs().max()) < 2e-7
    assert measured(f'f64-Kzx layer mean b{b} M{M} n{n}', mean, mean_ref, rtol=0.0, atol=3e-7 * float(mean_ref.abs().max()))
    # variance: error relative to the PRIOR variance os (what float32 output rounding allows), i.e. no cancellation loss
    assert measured(f'f64-Kzx layer var b{b} M{M} n{n} (min var / os = {float((var_ref / os_.double().reshape(b, 1)).min()):.2g})',
                    var, var_ref, rtol=2e-7, atol=2e-7 * float(os_.max()))
