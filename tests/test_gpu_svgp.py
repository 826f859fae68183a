"""GPU parity of the fused whitened-SVGP layer node (nsgp.svgp.SVGPLayerFn) against torch autograd of
the CPU oracle (oracle.svgp.svgp_marginal): marginals and every parameter gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64


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
    tol = dict(rtol=1e-9, atol=1e-10) if dt == F64 else dict(rtol=2e-3, atol=5e-4)
    assert torch.allclose(mean.detach().cpu().double(), mean_r, **tol)
    assert torch.allclose(var.detach().cpu().double(), var_r, **tol)
    ((mean * gm.to(dt).cuda()).sum() + (var * gv.to(dt).cuda()).sum()).backward()
    gtol = dict(rtol=1e-7, atol=1e-8) if dt == F64 else dict(rtol=2e-2, atol=2e-2)
    names = ['x', 'Z', 'ls', 'os', 'm', 'Lq']
    for name, a, r in zip(names, cu, grads_r):
        got = a.grad.cpu().double()
        ref = torch.tril(r) if name == 'Lq' else r
        scale = float(ref.abs().max()) + 1e-30
        err = float((got - ref).abs().max()) / scale
        assert err < (1e-7 if dt == F64 else 2e-2), (name, err)


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
