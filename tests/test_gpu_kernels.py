"""GPU parity: every C-ABI kernel (through nsgp.ops -> ctypes -> libnsgp_hip.so) against the CPU oracle
on the same seeded inputs.  float64 kernels must agree to ~1e-12, float32 kernels to ~1e-5 relative
(tolerances are written next to each assert).  Full-size cases use size-independent properties."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F32, F64 = torch.float32, torch.float64


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from nsgp import ops as _ops
    return _ops


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _tol(dt):
    return dict(rtol=1e-11, atol=1e-12) if dt == F64 else dict(rtol=2e-5, atol=2e-6)


def _gibbs_inputs(n1, n2, D, dt, seed=0):
    g = _g(seed)
    x1 = torch.randn(n1, D, generator=g, dtype=F64)
    x2 = torch.randn(n2, D, generator=g, dtype=F64)
    e1 = torch.exp(0.3 * torch.randn(D, n1, generator=g, dtype=F64) + math.log(0.5))
    e2 = torch.exp(0.3 * torch.randn(D, n2, generator=g, dtype=F64) + math.log(0.5))
    return [t.to(dt) for t in (x1, x2, e1, e2)]


# ------------------------------------------------------------------------------------------ K1
@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('n1,n2,D', [(316, 78, 2), (1, 1, 1), (5, 700, 3), (130, 257, 5), (64, 64, 1)])
def test_gibbs_fwd(ops, dt, n1, n2, D):
    from oracle import kernels as K
    x1, x2, e1, e2 = _gibbs_inputs(n1, n2, D, dt)
    ref = 0.644 * K.gibbs(x1.double(), x2.double(), e1.double(), e2.double())
    got = ops.gibbs_build(x1.cuda(), x2.cuda(), e1.cuda(), e2.cuda(), outputscale=0.644).cpu()
    assert torch.allclose(got.double(), ref, **_tol(dt))


@pytest.mark.parametrize('dt', [F64, F32])
def test_gibbs_fwd_diag_add_and_device_scalars(ops, dt):
    from oracle import kernels as K
    x1, _, e1, _ = _gibbs_inputs(200, 1, 2, dt)
    os_ = torch.tensor(0.7, dtype=dt, device='cuda')
    noise = torch.tensor([0.011], dtype=dt, device='cuda')
    got = ops.gibbs_build(x1.cuda(), x1.cuda(), e1.cuda(), e1.cuda(), os_, noise).cpu().double()
    ref = 0.7 * K.gibbs(x1.double(), x1.double(), e1.double(), e1.double()) + 0.011 * torch.eye(200, dtype=F64)
    assert torch.allclose(got, ref, **_tol(dt))


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('n1,n2,D', [(316, 78, 2), (70, 300, 3), (33, 65, 1), (40, 50, 5)])
def test_gibbs_bwd_matches_oracle_autograd(ops, dt, n1, n2, D):
    from oracle import kernels as K
    x1, x2, e1, e2 = _gibbs_inputs(n1, n2, D, dt, seed=1)
    G = torch.randn(n1, n2, generator=_g(2), dtype=F64).to(dt)
    ins = [t.detach().clone().double().requires_grad_() for t in (x1, x2, e1, e2)]
    os_ = torch.tensor(0.9, dtype=F64, requires_grad=True)
    (os_ * K.gibbs(*ins) * G.double()).sum().backward()
    cu = [t.detach().clone().cuda().requires_grad_() for t in (x1, x2, e1, e2)]
    os_c = torch.tensor(0.9, dtype=dt, device='cuda', requires_grad=True)
    Kc = ops.gibbs_kernel(cu[0], cu[1], cu[2], cu[3], os_c)
    (Kc * G.cuda()).sum().backward()
    tol = dict(rtol=1e-9, atol=1e-10) if dt == F64 else dict(rtol=2e-3, atol=2e-3)
    for a, b in zip(cu, ins):
        assert torch.allclose(a.grad.cpu().double(), b.grad, **tol)
    assert torch.allclose(os_c.grad.cpu().double(), os_.grad, **tol)


@pytest.mark.parametrize('dt', [F64, F32])
def test_gibbs_and_rbf_exponent_range(ops, dt):
    """The build kernels' own exp (csrc/common.h) over the whole argument range: entries whose exponent runs from 0
    through the float64 denormal range to far below it (-1e6) must agree with the oracle to its rounding, underflow to
    (sub)normal zero without NaNs, and a NaN coordinate must stay a NaN in its row and column."""
    from oracle import kernels as K
    n, D = 96, 2
    g = _g(21)
    # points on a line with geometrically growing gaps: squared distances from 1e-6 to 1e6 lengthscales^2
    t = torch.cat([torch.zeros(1, dtype=F64), torch.logspace(-3, 3, n - 1, dtype=F64)])
    x = torch.stack([t, 0.5 * t], -1)
    e = torch.exp(0.2 * torch.randn(D, n, generator=g, dtype=F64))
    Kd = ops.gibbs_build(x.to(dt).cuda(), x.to(dt).cuda(), e.to(dt).cuda(), e.to(dt).cuda()).cpu().double()
    ref = K.gibbs(x.to(dt).double(), x.to(dt).double(), e.to(dt).double(), e.to(dt).double())
    assert torch.isfinite(Kd).all()
    tol = dict(rtol=1e-11, atol=1e-300) if dt == F64 else dict(rtol=3e-5, atol=1e-37)
    assert torch.allclose(Kd, ref, **tol)
    assert float(Kd[0, -1]) == 0.0 or abs(float(Kd[0, -1])) < 1e-300        # exponent ~ -1e6: underflow, not NaN
    ls = torch.ones(1, D, dtype=F64)
    os_ = torch.ones(1, dtype=F64)
    Kr = ops.rbf_build(x.to(dt).cuda().unsqueeze(0), x.to(dt).cuda().unsqueeze(0), ls.to(dt).cuda(), os_.to(dt).cuda())
    refr = torch.exp(-0.5 * ((x.to(dt).double()[:, None, :] - x.to(dt).double()[None, :, :]) ** 2).sum(-1))
    assert torch.allclose(Kr.cpu().double()[0], refr, **tol)
    xn = x.clone()
    xn[7, 0] = float('nan')
    Kn = ops.gibbs_build(xn.to(dt).cuda(), xn.to(dt).cuda(), e.to(dt).cuda(), e.to(dt).cuda()).cpu()
    assert torch.isnan(Kn[7]).all() and torch.isnan(Kn[:, 7]).all()
    keep = [i for i in range(n) if i != 7]
    assert torch.isfinite(Kn[keep][:, keep]).all()


def test_gibbs_symmetric_roles_sum(ops):
    """K_xx with ell1 is ell2 (training path, models/gibbs_kernels.py:148-149): autograd adds both roles."""
    from oracle import kernels as K
    x, _, e, _ = _gibbs_inputs(150, 1, 2, F64, seed=3)
    G = torch.randn(150, 150, generator=_g(4), dtype=F64)
    eo = e.clone().requires_grad_()
    (K.gibbs(x, x, eo, eo) * G).sum().backward()
    ec = e.cuda().requires_grad_()
    xc = x.cuda()
    (ops.gibbs_kernel(xc, xc, ec, ec) * G.cuda()).sum().backward()
    assert torch.allclose(ec.grad.cpu(), eo.grad, rtol=1e-9, atol=1e-10)


def test_gibbs_full_size_properties(ops):
    """N = 4096 (BASELINE config 2 synthetic): unit diagonal, symmetry, constant-ell == RBF, all on device."""
    n = 4096
    g = _g(5)
    x = torch.randn(n, 2, generator=g, dtype=F32).cuda()
    e = torch.exp(0.3 * torch.randn(2, n, generator=g, dtype=F32) + math.log(0.3)).cuda()
    K = ops.gibbs_build(x, x, e, e)
    assert torch.allclose(torch.diagonal(K), torch.ones(n, device='cuda'), atol=1e-6)
    assert torch.allclose(K, K.T, rtol=0, atol=2e-7)       # symmetric to 1 ulp (FMA contraction order)
    c = torch.full((2, n), 0.37, device='cuda')
    Kc = ops.gibbs_build(x, x, c, c)
    Kr = ops.rbf_build(x, x, torch.full((1, 2), 0.37, device='cuda'), torch.ones(1, device='cuda'))[0]
    assert torch.allclose(Kc, Kr, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------ K2
@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('shared', [True, False])
def test_rbf_fwd_bwd(ops, dt, shared):
    from oracle import kernels as K
    g = _g(6)
    b, n1, n2, D = 2, 130, 300, 3
    x1 = torch.randn(b, n1, D, generator=g, dtype=F64)
    x2 = torch.randn(n2, D, generator=g, dtype=F64) if shared else torch.randn(b, n2, D, generator=g, dtype=F64)
    ls = torch.rand(b, D, generator=g, dtype=F64) + 0.5
    os_ = torch.rand(b, generator=g, dtype=F64) + 0.5
    G = torch.randn(b, n1, n2, generator=g, dtype=F64)
    ins = [t.clone().requires_grad_() for t in (x1, x2, ls, os_)]
    ref = K.rbf_ard(ins[0], ins[1], ins[2].unsqueeze(-2), ins[3])
    (ref * G).sum().backward()
    cu = [t.to(dt).cuda().requires_grad_() for t in (x1, x2, ls, os_)]
    got = ops.rbf_kernel(cu[0], cu[1], cu[2], cu[3])
    assert torch.allclose(got.detach().cpu().double(), ref.detach(), **_tol(dt))
    (got * G.to(dt).cuda()).sum().backward()
    tol = dict(rtol=1e-9, atol=1e-10) if dt == F64 else dict(rtol=2e-3, atol=2e-3)
    for a, r in zip(cu, ins):
        assert torch.allclose(a.grad.cpu().double(), r.grad, **tol)


def test_rbf_diag_add(ops):
    from oracle import kernels as K
    Z = torch.randn(3, 70, 2, generator=_g(7), dtype=F64)
    ls = torch.ones(3, 2, dtype=F64) * 0.8
    os_ = torch.ones(3, dtype=F64) * 0.7
    got = ops.rbf_build(Z.cuda(), Z.cuda(), ls.cuda(), os_.cuda(), diag_add=1e-4).cpu()
    ref = K.rbf_ard(Z, Z, ls.unsqueeze(-2), os_) + 1e-4 * torch.eye(70, dtype=F64)
    assert torch.allclose(got, ref, rtol=1e-12, atol=1e-13)


# ------------------------------------------------------------------------------------------ K3
@pytest.mark.parametrize('dt', [F64, F32])
def test_ps2d_fwd_bwd(ops, dt):
    from oracle import kernels as K
    g = _g(8)
    n1, n2 = 90, 140
    x1 = torch.rand(n1, 2, generator=g, dtype=F64)
    x2 = torch.rand(n2, 2, generator=g, dtype=F64)
    H1, H2 = torch.randn(n1, 2, generator=g, dtype=F64), torch.randn(n2, 2, generator=g, dtype=F64)
    Dm = torch.tensor([[0.7, 0.1], [-0.2, 0.9]], dtype=F64)
    s1 = K.ps_sigma(H1, Dm).requires_grad_()
    s2 = K.ps_sigma(H2, Dm).requires_grad_()
    G = torch.randn(n1, n2, generator=g, dtype=F64)
    ref = K.ps2d(x1, x2, s1, s2, 1e-5)
    (ref * G).sum().backward()
    c1 = s1.detach().to(dt).cuda().requires_grad_()
    c2 = s2.detach().to(dt).cuda().requires_grad_()
    got = ops.ps2d_kernel(x1.to(dt).cuda(), x2.to(dt).cuda(), c1, c2, 1e-5)
    assert torch.allclose(got.detach().cpu().double(), ref.detach(), **_tol(dt))
    (got * G.to(dt).cuda()).sum().backward()
    tol = dict(rtol=1e-9, atol=1e-10) if dt == F64 else dict(rtol=3e-3, atol=3e-3)
    assert torch.allclose(c1.grad.cpu().double(), s1.grad, **tol)
    assert torch.allclose(c2.grad.cpu().double(), s2.grad, **tol)


# ------------------------------------------------------------------------------------------ GEMM
def _gemm_ref(A, B, ta, tb):
    A, B = A.double(), B.double()
    return (A.transpose(-1, -2) if ta else A) @ (B.transpose(-1, -2) if tb else B)


@pytest.mark.parametrize('dt', [F32, F64])
@pytest.mark.parametrize('ta', [False, True])
@pytest.mark.parametrize('tb', [False, True])
@pytest.mark.parametrize('M,N,K', [(250, 315, 130), (128, 128, 64), (1, 7, 3), (300, 1100, 260), (64, 64, 16)])
def test_gemm_all_layouts(ops, dt, ta, tb, M, N, K):
    g = _g(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=dt)
    B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=dt)
    got = ops.gemm(A.cuda(), B.cuda(), ta, tb).cpu().double()
    ref = _gemm_ref(A, B, ta, tb)
    tol = dict(rtol=1e-12, atol=1e-11) if dt == F64 else dict(rtol=1e-4, atol=2e-4 * math.sqrt(K))
    assert torch.allclose(got, ref, **tol)


@pytest.mark.parametrize('dt', [F32, F64])
def test_gemm_mfma_layout_with_asymmetric_integer_data(ops, dt):
    """A = I with an asymmetric B catches any row/col swap of the accumulator map (exact in integers)."""
    n = 192
    A = torch.eye(n, dtype=dt)
    B = (torch.arange(n * n, dtype=F64).reshape(n, n) % 97).to(dt)
    assert torch.equal(ops.gemm(A.cuda(), B.cuda()).cpu(), B)
    assert torch.equal(ops.gemm(B.cuda(), A.cuda()).cpu(), B)
    assert torch.equal(ops.gemm(B.cuda(), A.cuda(), ta=True).cpu(), B.T.contiguous())


@pytest.mark.parametrize('dt', [F32, F64])
def test_gemm_splitk_alpha_beta_batched(ops, dt):
    g = _g(11)
    A = torch.randn(3, 128, 4096, generator=g, dtype=dt)
    B = torch.randn(3, 192, 4096, generator=g, dtype=dt)
    C0 = torch.randn(3, 128, 192, generator=g, dtype=dt)
    out = C0.clone().cuda()
    ops.gemm(A.cuda(), B.cuda(), tb=True, alpha=0.5, beta=-2.0, out=out)
    ref = 0.5 * _gemm_ref(A, B, False, True) - 2.0 * C0.double()
    tol = dict(rtol=1e-12, atol=1e-10) if dt == F64 else dict(rtol=1e-4, atol=2e-2)
    assert torch.allclose(out.cpu().double(), ref, **tol)
    # a 2-D operand broadcasts over the batch
    W = torch.randn(128, 128, generator=g, dtype=dt)
    got = ops.gemm(W.cuda(), A.cuda()[:, :, :300].contiguous()).cpu().double()
    assert torch.allclose(got, W.double() @ A[:, :, :300].double(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize('dt', [F32, F64])
@pytest.mark.parametrize('n', [200, 512])
def test_gemm_triangular_flags_mask_the_other_triangle(ops, dt, n):
    """Lower/upper flags must (a) skip K-tiles and (b) never read the other triangle as data."""
    g = _g(12)
    L = torch.randn(n, n, generator=g, dtype=dt)          # garbage in the strict upper triangle
    X = torch.randn(n, 300, generator=g, dtype=dt)
    Lt = torch.tril(L)
    tol = dict(rtol=1e-12, atol=1e-10) if dt == F64 else dict(rtol=1e-4, atol=5e-3)
    got = ops.gemm(L.cuda(), X.cuda(), flags=ops.GEMM_A_LOWER).cpu().double()
    assert torch.allclose(got, Lt.double() @ X.double(), **tol)
    got = ops.gemm(L.cuda(), X.cuda(), ta=True, flags=ops.GEMM_A_UPPER).cpu().double()
    assert torch.allclose(got, Lt.double().T @ X.double(), **tol)
    Y = torch.randn(300, n, generator=g, dtype=dt)
    got = ops.gemm(Y.cuda(), L.cuda(), flags=ops.GEMM_B_LOWER).cpu().double()
    assert torch.allclose(got, Y.double() @ Lt.double(), **tol)
    got = ops.gemm(Y.cuda(), L.cuda(), tb=True, flags=ops.GEMM_B_UPPER).cpu().double()
    assert torch.allclose(got, Y.double() @ Lt.double().T, **tol)
    got = ops.gemm(X.cuda(), X.cuda(), tb=True, flags=ops.GEMM_C_LOWER).cpu().double()
    assert torch.allclose(got, torch.tril(X.double() @ X.double().T), **tol)


@pytest.mark.parametrize('dt', [F32, F64])
@pytest.mark.parametrize('n,batch', [(200, 1), (512, 3), (1024, 2)])
def test_gemm_combined_triangular_flags_of_the_cholesky_adjoint(ops, dt, n, batch):
    """The three products of WhitenFn.backward keep all their triangular structure: lower triangle of (lower x upper),
    lower x lower (a lower result), upper x lower.  Garbage (incl. NaN for the never-written triangle) sits in every
    triangle the flags declare zero; with C_NOFILL the strict upper triangle of an output is unspecified."""
    g = _g(40 + n)
    Wb = torch.randn(batch, n, n, generator=g, dtype=dt)
    W = torch.randn(batch, n, n, generator=g, dtype=dt) / n ** 0.5
    junk = torch.triu(torch.full((n, n), float('nan'), dtype=dt), 1)
    Wb_in, W_in = (torch.tril(Wb) + junk).cuda(), (torch.tril(W) + junk).cuda()
    Wbl, Wl = torch.tril(Wb).double(), torch.tril(W).double()
    tol = dict(rtol=1e-11, atol=1e-10) if dt == F64 else dict(rtol=2e-4, atol=2e-3)
    fl = ops.GEMM_C_LOWER | ops.GEMM_C_NOFILL
    Phi = torch.full((batch, n, n), 7.0, dtype=dt, device='cuda')
    ops.gemm(Wb_in, W_in, tb=True, flags=ops.GEMM_A_LOWER | ops.GEMM_B_UPPER | fl, out=Phi)
    ref = Wbl @ Wl.transpose(-1, -2)
    assert torch.allclose(torch.tril(Phi.cpu().double()), torch.tril(ref), **tol)
    ops.scale_diag_(Phi, 0.5)
    Phi_ref = torch.tril(ref) - 0.5 * torch.diag_embed(torch.diagonal(ref, dim1=-2, dim2=-1))
    assert torch.allclose(torch.tril(Phi.cpu().double()), Phi_ref, **tol)
    # the same Phi in one launch: the product's epilogue (or its split-K reduce) halves the diagonal
    Phi1 = ops.gemm(Wb_in, W_in, tb=True, flags=ops.GEMM_A_LOWER | ops.GEMM_B_UPPER | fl | ops.GEMM_C_HALFDIAG)
    assert torch.equal(torch.tril(Phi1), torch.tril(Phi))
    Phi_in = torch.tril(Phi) + junk.cuda()
    T = ops.gemm(Phi_in, W_in, flags=ops.GEMM_A_LOWER | ops.GEMM_B_LOWER | fl)
    T_ref = Phi_ref @ Wl
    assert torch.allclose(torch.tril(T.cpu().double()), T_ref, **tol)
    T_in = torch.tril(T) + junk.cuda()
    G = ops.gemm(W_in, T_in, ta=True, alpha=-1.0, flags=ops.GEMM_A_UPPER | ops.GEMM_B_LOWER)
    assert torch.allclose(G.cpu().double(), -(Wl.transpose(-1, -2) @ T_ref), **tol)


def test_matmul_autograd_recursion(ops):
    g = _g(13)
    L = torch.randn(90, 90, generator=g, dtype=F64)           # lower-triangular operand, garbage above
    cases = [  # (A, B, ta, tb, a_lower)
        (L, torch.randn(90, 55, generator=g, dtype=F64), False, False, True),
        (L, torch.randn(90, 55, generator=g, dtype=F64), True, False, True),
        (torch.randn(90, 70, generator=g, dtype=F64), torch.randn(55, 70, generator=g, dtype=F64), False, True, False),
        (torch.randn(70, 90, generator=g, dtype=F64), torch.randn(55, 70, generator=g, dtype=F64), True, True, False),
    ]
    for A0, B0, ta, tb, al in cases:
        a, bm = A0.clone().requires_grad_(), B0.clone().requires_grad_()
        aa = torch.tril(a) if al else a
        ref = (aa.T if ta else aa) @ (bm.T if tb else bm)
        Gm = torch.randn(ref.shape, generator=g, dtype=F64)
        (ref * Gm).sum().backward()
        ac, bc = A0.cuda().requires_grad_(), B0.cuda().requires_grad_()
        got = ops.matmul(ac, bc, ta, tb, a_lower=al)
        assert torch.allclose(got.detach().cpu(), ref.detach(), rtol=1e-11, atol=1e-11)
        (got * Gm.cuda()).sum().backward()
        assert torch.allclose(ac.grad.cpu(), a.grad, rtol=1e-10, atol=1e-10)
        assert torch.allclose(bc.grad.cpu(), bm.grad, rtol=1e-10, atol=1e-10)


# ------------------------------------------------------------------------------------------ K4 / K5
def _spd(n, dt, seed, batch=None):
    g = _g(seed)
    shp = (n, n) if batch is None else (batch, n, n)
    A = torch.randn(shp, generator=g, dtype=F64)
    A = A @ A.transpose(-1, -2) / n + 0.5 * torch.eye(n, dtype=F64)
    return A.to(dt)


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('n', [1, 7, 63, 64, 65, 130, 250, 316, 1024])
def test_potrf_and_trtri(ops, dt, n):
    A = _spd(n, dt, n)
    L, info = ops.potrf(A.cuda())
    assert int(info.item()) == 0
    Lc = L.cpu().double()
    assert torch.equal(torch.triu(Lc, 1), torch.zeros_like(Lc))        # strict upper zeroed
    ref = torch.linalg.cholesky(A.double())
    tol = dict(rtol=1e-10, atol=1e-11) if dt == F64 else dict(rtol=2e-3, atol=2e-4)
    assert torch.allclose(Lc, ref, **tol)
    X = ops.trtri(L).cpu().double()
    assert torch.equal(torch.triu(X, 1), torch.zeros_like(X))
    eye = torch.eye(n, dtype=F64)
    assert torch.allclose(X @ Lc, eye, atol=1e-9 if dt == F64 else 2e-3)


@pytest.mark.parametrize('dt', [F64, F32])
def test_potrf_batched_and_info(ops, dt):
    A = _spd(200, dt, 3, batch=3)
    A[1, 150, 150] = -5.0                                  # breaks positive-definiteness at minor 151
    L, info = ops.potrf(A.cuda())
    assert info.cpu().tolist()[0] == 0 and info.cpu().tolist()[2] == 0
    assert info.cpu().tolist()[1] == 151
    for b in (0, 2):
        ref = torch.linalg.cholesky(A[b].double())
        assert torch.allclose(L[b].cpu().double(), ref, rtol=2e-3, atol=2e-4)
    with pytest.raises(Exception):
        ops.potrf(A.cuda(), check=True)


def test_potrf_full_size_residual(ops):
    """N = 4096 fp64 (BASELINE B2): ||L L^T - A|| / ||A|| at rounding level, all on device."""
    n = 4096
    x = torch.randn(n, 2, generator=_g(9), dtype=F64).cuda()
    e = torch.exp(0.3 * torch.randn(2, n, generator=_g(10), dtype=F64) + math.log(0.3)).cuda()
    A = ops.gibbs_build(x, x, e, e, 0.644, 0.011)
    L, info = ops.potrf(A)
    assert int(info.item()) == 0
    R = ops.gemm(L, L, tb=True) - A
    assert float(R.norm() / A.norm()) < 1e-14
    X = ops.trtri(L)
    E = ops.gemm(X, L) - torch.eye(n, dtype=F64, device='cuda')
    assert float(E.abs().max()) < 1e-9


def test_chol_inv_autograd(ops):
    A = _spd(150, F64, 21)
    Wbar = torch.randn(150, 150, generator=_g(22), dtype=F64)
    a = A.clone().requires_grad_()
    Lr = torch.linalg.cholesky(a)
    Wr = torch.linalg.solve_triangular(Lr, torch.eye(150, dtype=F64), upper=False)
    (Wr * torch.tril(Wbar)).sum().backward()
    ac = A.cuda().requires_grad_()
    W, info = ops.chol_inv(ac)
    assert torch.allclose(W.detach().cpu(), Wr.detach(), rtol=1e-9, atol=1e-10)
    (W * Wbar.cuda()).sum().backward()
    sym = 0.5 * (a.grad + a.grad.T)
    assert torch.allclose(ac.grad.cpu(), sym, rtol=1e-8, atol=1e-9)


# ------------------------------------------------------------------------------------------ K6 / K7
@pytest.mark.parametrize('dt', [F64, F32])
def test_colstats_and_bwd(ops, dt):
    g = _g(30)
    b, M, n = 2, 100, 333
    A = torch.randn(b, M, n, generator=g, dtype=F64)
    C = torch.randn(b, M, n, generator=g, dtype=F64)
    m = torch.randn(b, M, generator=g, dtype=F64)
    base = torch.rand(b, generator=g, dtype=F64)
    gm, gv = torch.randn(b, n, generator=g, dtype=F64), torch.randn(b, n, generator=g, dtype=F64)
    mean_r = torch.einsum('bkj,bk->bj', A, m)
    var_r = base[:, None] + (C * C - A * A).sum(1)
    mean, var = ops.svgp_colstats(A.to(dt).cuda(), C.to(dt).cuda(), m.to(dt).cuda(), base.to(dt).cuda())
    tol = dict(rtol=1e-11, atol=1e-11) if dt == F64 else dict(rtol=1e-4, atol=1e-3)
    assert torch.allclose(mean.cpu().double(), mean_r, **tol)
    assert torch.allclose(var.cpu().double(), var_r, **tol)
    Abar, C2, mbar = ops.svgp_colstats_bwd(A.to(dt).cuda(), C.to(dt).cuda(), m.to(dt).cuda(), gm.to(dt).cuda(),
                                           gv.to(dt).cuda())
    assert torch.allclose(Abar.cpu().double(), m[:, :, None] * gm[:, None, :] - 2 * gv[:, None, :] * A, **tol)
    assert torch.allclose(C2.cpu().double(), 2 * gv[:, None, :] * C, **tol)
    assert torch.allclose(mbar.cpu().double(), torch.einsum('bkj,bj->bk', A, gm), **tol)


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('ns_is_one', [True, False])
def test_dgp_sample_fwd_bwd(ops, dt, ns_is_one):
    g = _g(31)
    S, n, b = 4, 77, 2
    ns = 1 if ns_is_one else S
    mean = torch.randn(b, ns, n, generator=g, dtype=F64).requires_grad_()
    var = (torch.rand(b, ns, n, generator=g, dtype=F64) + 0.1).requires_grad_()
    eps = torch.randn(S, n, b, generator=g, dtype=F64)
    gh = torch.randn(S, n, b, generator=g, dtype=F64)
    h_ref = mean.permute(1, 2, 0) + var.sqrt().permute(1, 2, 0) * eps       # (ns|S, n, b) broadcast
    (h_ref * gh).sum().backward()
    mc = mean.detach().to(dt).cuda().requires_grad_()
    vc = var.detach().to(dt).cuda().requires_grad_()
    h = ops.DgpSampleFn.apply(mc, vc, eps.to(dt).cuda())
    tol = dict(rtol=1e-12, atol=1e-12) if dt == F64 else dict(rtol=1e-5, atol=1e-5)
    assert torch.allclose(h.detach().cpu().double(), h_ref.detach().expand(S, n, b), **tol)
    (h * gh.to(dt).cuda()).sum().backward()
    assert torch.allclose(mc.grad.cpu().double(), mean.grad, **tol)
    assert torch.allclose(vc.grad.cpu().double(), var.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('dt', [F64, F32])
def test_gauss_ell_and_kl(ops, dt):
    from oracle import svgp
    g = _g(32)
    S, n, M = 5, 315, 60
    y = torch.randn(n, generator=g, dtype=F64)
    mu = torch.randn(S, n, generator=g, dtype=F64).requires_grad_()
    v = (torch.rand(S, n, generator=g, dtype=F64) + 0.1).requires_grad_()
    noise = torch.tensor(0.37, dtype=F64, requires_grad=True)
    w = torch.randn(S, generator=g, dtype=F64)                       # upstream gradient per sample
    ref = svgp.gauss_ell(y, mu, v, noise).sum(-1) / n                # (S,)
    (ref * w).sum().backward()
    muc, vc = mu.detach().to(dt).cuda().requires_grad_(), v.detach().to(dt).cuda().requires_grad_()
    nc = noise.detach().to(dt).cuda().requires_grad_()
    got = ops.GaussEllFn.apply(y.to(dt).cuda(), muc, vc, nc, 1.0 / n)
    tol = dict(rtol=1e-11, atol=1e-12) if dt == F64 else dict(rtol=1e-4, atol=1e-5)
    assert got.shape == (S,)
    assert torch.allclose(got.detach().cpu().double(), ref.detach(), **tol)
    (got * w.to(dt).cuda()).sum().backward()
    assert torch.allclose(muc.grad.cpu().double(), mu.grad, **tol)
    assert torch.allclose(vc.grad.cpu().double(), v.grad, **tol)
    assert torch.allclose(nc.grad.cpu().double(), noise.grad, **tol)
    # KL
    m = torch.randn(2, M, generator=g, dtype=F64).requires_grad_()
    Lq = (torch.tril(0.1 * torch.randn(2, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64)
          + torch.triu(torch.randn(2, M, M, generator=g, dtype=F64), 1)).requires_grad_()    # garbage upper
    klr = svgp.kl_whitened(dict(m=m, Lq=Lq))
    klr.backward()
    mc, Lc = m.detach().to(dt).cuda().requires_grad_(), Lq.detach().to(dt).cuda().requires_grad_()
    kl = ops.KlWhitenedFn.apply(mc, Lc)
    assert torch.allclose(kl.detach().cpu().double(), klr.detach(), **tol)
    kl.backward()
    assert torch.allclose(mc.grad.cpu().double(), m.grad, **tol)
    assert torch.allclose(Lc.grad.cpu().double(), Lq.grad, rtol=1e-4, atol=1e-5)


def test_philox_matches_cpu_restatement_and_is_partition_invariant(ops):
    from oracle import philox
    S, n, b = 3, 1000, 2
    ref = philox.normal(173, 7, 0, S, n, b)
    got64 = ops.philox_normal(173, 7, 0, S, n, b, dtype=F64).cpu().numpy()
    assert np.allclose(got64, ref, rtol=0, atol=1e-12)
    got32 = ops.philox_normal(173, 7, 0, S, n, b, dtype=F32).cpu().numpy()
    assert np.allclose(got32, ref, atol=1e-6)
    # two "ranks" drawing disjoint row ranges reproduce the single-GPU draw
    lo = ops.philox_normal(173, 7, 0, S, 400, b, dtype=F64).cpu().numpy()
    hi = ops.philox_normal(173, 7, 400, S, 600, b, dtype=F64).cpu().numpy()
    assert np.array_equal(np.concatenate([lo, hi], axis=1), got64)
    # 5 columns exercise the second counter word
    ref5 = philox.normal(1 << 40, (9 << 32) | 5, 123456789012, 2, 10, 5)
    got5 = ops.philox_normal(1 << 40, (9 << 32) | 5, 123456789012, 2, 10, 5, dtype=F64).cpu().numpy()
    assert np.allclose(got5, ref5, atol=1e-12)


def test_fused_adam_matches_oracle(ops):
    from oracle import svgp
    g = _g(33)
    p0 = torch.randn(10001, generator=g, dtype=F32)
    p = p0.clone().cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    mine, state = [p0.double()], {}
    for step in range(1, 4):
        gr = torch.randn(10001, generator=g, dtype=F32)
        ops.adam_step_(p, gr.cuda(), m, v, 0.01, 0.9, 0.999, 1e-8, step)
        mine = svgp.adam_step(mine, [gr.double()], state)
    assert torch.allclose(p.cpu().double(), mine[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('b,n,D', [(1, 70, 2), (3, 257, 3), (2, 1024, 2)])
def test_rbf_backward_symmetric_mode_sums_both_sides(ops, dt, b, n, D):
    """rbf_build_bwd(x, x, sym=True): one gradient buffer for both arguments receives g_x1 + g_x2 (the Kzz case of the
    whitening chain), the other outputs are unchanged -- against the two-buffer call on the same (non-symmetric) G."""
    g = torch.Generator().manual_seed(11 + n)
    x = torch.randn(b, n, D, generator=g, dtype=F64).to(dt).cuda()
    ls = (torch.rand(b, D, generator=g, dtype=F64) + 0.5).to(dt).cuda()
    os_ = (torch.rand(b, generator=g, dtype=F64) + 0.5).to(dt).cuda()
    G = torch.randn(b, n, n, generator=g, dtype=F64).to(dt).cuda()
    g1, g2, gls, gos = ops.rbf_build_bwd(x, x, ls, os_, G)
    s1, s2, sls, sos = ops.rbf_build_bwd(x, x, ls, os_, G, sym=True)
    assert s1 is s2
    tol = dict(rtol=1e-11, atol=1e-11) if dt == F64 else dict(rtol=2e-5, atol=2e-5)
    assert torch.allclose(s1, g1 + g2, **tol)
    assert torch.equal(sls, gls) and torch.equal(sos, gos)


def test_cpu_tensors_are_rejected_loudly(ops):
    from nsgp import BackendError
    x = torch.randn(4, 2)
    e = torch.ones(2, 4)
    with pytest.raises(BackendError):
        ops.gibbs_build(x, x, e, e)


# ------------------------------------------------------------------------ ABI edge cases
def test_empty_inputs_are_no_ops_and_bad_arguments_report_their_index(ops):
    """The C ABI's error convention (include/nsgp.h): 0 = ok, -k = argument k invalid; zero-sized problems return 0
    without touching memory.  Called straight through ctypes (no Python-side validation in the way)."""
    import ctypes
    from nsgp import _lib
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.randn(4, 2, device='cuda', dtype=F64)
    ell = torch.rand(2, 4, device='cuda', dtype=F64) + 0.5
    K = torch.full((4, 4), 7.0, device='cuda', dtype=F64)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    # empty Gibbs build: n1 == 0 or n2 == 0
    assert lib.nsgp_gibbs_build_fwd_f64(P(x), P(x), P(ell), P(ell), 0, 4, 2, None, None, P(K), 4, st) == 0
    assert lib.nsgp_gibbs_build_fwd_f64(P(x), P(x), P(ell), P(ell), 4, 0, 2, None, None, P(K), 4, st) == 0
    torch.cuda.synchronize()
    assert float(K.min()) == 7.0                                       # untouched
    # bad arguments: null x1 (arg 1), D = 0 (arg 7), ldk < n2 (arg 11)
    assert lib.nsgp_gibbs_build_fwd_f64(None, P(x), P(ell), P(ell), 4, 4, 2, None, None, P(K), 4, st) == -1
    assert lib.nsgp_gibbs_build_fwd_f64(P(x), P(x), P(ell), P(ell), 4, 4, 0, None, None, P(K), 4, st) == -7
    assert lib.nsgp_gibbs_build_fwd_f64(P(x), P(x), P(ell), P(ell), 4, 4, 2, None, None, P(K), 3, st) == -11
    # GEMM: M == 0 / N == 0 are no-ops; K == 0 writes beta * C; contradictory triangle flags are rejected
    A = torch.randn(4, 4, device='cuda', dtype=F32)
    C = torch.full((4, 4), 3.0, device='cuda', dtype=F32)
    g = lambda M, N, Kd, beta, flags: lib.nsgp_gemm_f32(M, N, Kd, 1.0, P(A), 4, 1, 0, 0, P(A), 4, 1, 0, 0, beta, P(C), 4, 0, 0,
                                                       1, 1, flags, None, 0, st)
    assert g(0, 4, 4, 0.0, 0) == 0 and g(4, 0, 4, 0.0, 0) == 0
    torch.cuda.synchronize()
    assert float(C.min()) == 3.0
    assert g(4, 4, 0, 2.0, 0) == 0
    torch.cuda.synchronize()
    assert torch.allclose(C, torch.full_like(C, 6.0))
    assert g(4, 4, 4, 0.0, ops.GEMM_A_LOWER | ops.GEMM_A_UPPER) == -22
    assert g(-1, 4, 4, 0.0, 0) == -1
    # potrf / trtri with n == 0 or batch == 0
    info = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert lib.nsgp_potrf_f64(P(K), 0, 4, 16, 1, P(info), None, 0, st) == 0
    assert lib.nsgp_potrf_f64(P(K), 4, 4, 16, 0, P(info), None, 0, st) == 0
    assert lib.nsgp_potrf_f64(P(K), 4, 3, 16, 1, P(info), None, 0, st) == -3          # lda < n
    assert lib.nsgp_potrf_f64(P(K), 4, 4, 16, 1, P(info), None, 0, st) == -7          # workspace missing
    assert lib.nsgp_trtri_f64(P(K), 0, 4, 16, P(K), 4, 16, 1, None, 0, st) == 0
    # fused SVGP GEMM: empty batch
    assert lib.nsgp_svgp_tri_gemm_colstats_f32(P(A), 0, P(A), None, 0, 4, 4, P(C), None, P(C), st) == 0
    assert lib.nsgp_svgp_tri_gemm_colstats_f32(P(A), 2, P(A), None, 1, 4, 4, P(C), None, P(C), st) == -2   # trans not 0/1
    # affine finalize / rowdot: empty problems, linear part without x (arg 9), D beyond NSGP_MAX_DIM (arg 11 / 6)
    v = torch.zeros(8, device='cuda', dtype=F32)
    fin = lambda batch, n, x, D, w: lib.nsgp_svgp_colstats_finalize_affine_f32(
        P(A), P(A), P(A), P(v), 0.0, batch, 1, n, x, 0, D, w, 0, None, 0, P(v), P(C), st)
    assert fin(0, 4, None, 0, None) == 0 and fin(1, 0, None, 0, None) == 0
    assert fin(1, 4, None, 2, P(v)) == -9
    assert fin(1, 4, P(A), 1000, P(v)) == -11
    rd = lambda batch, D, x, ox: lib.nsgp_rowdot_affine_f32(P(A), P(v), None, x, 0, D, 1, batch, 2, 2, P(C), None, ox,
                                                          None, st)
    assert rd(0, 0, None, None) == 0
    assert rd(1, 1000, None, None) == -6
    assert rd(1, 2, None, P(v)) == -13                                  # weight gradient asked for without x
    # a non-positive-definite matrix is reported LAPACK-style in info (1-based index of the failing minor)
    Bad = torch.eye(70, device='cuda', dtype=F64)
    Bad[65, 65] = -1.0
    L, info = ops.potrf(Bad)
    assert int(info[0]) == 66


@pytest.mark.parametrize('dt', [F64, F32])
def test_scalar_total_elbo_terms_match_autograd_of_the_vector_forms(ops, dt):
    """GaussEllTotalFn / KlWhitenedTotalFn (one scalar, device-resident upstream gradient) against torch autograd of
    the float64 closed forms, including a non-trivial upstream factor."""
    g = _g(77)
    S, n, b, M = 5, 333, 2, 70
    y = torch.randn(n, generator=g, dtype=F64)
    mu = torch.randn(S, n, generator=g, dtype=F64)
    v = torch.rand(S, n, generator=g, dtype=F64) + 0.1
    noise = torch.tensor([0.37], dtype=F64)
    m = 0.3 * torch.randn(b, M, generator=g, dtype=F64)
    Lq = torch.tril(0.1 * torch.randn(b, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64)
    ref_in = [t.clone().requires_grad_() for t in (mu, v, noise, m, Lq)]
    mu_r, v_r, n_r, m_r, L_r = ref_in
    ell = (-0.5 * (((y - mu_r) ** 2 + v_r) / n_r + torch.log(n_r) + math.log(2 * math.pi))).sum()
    Lt = torch.tril(L_r)
    kl = 0.5 * ((Lt ** 2).sum() + (m_r ** 2).sum() - b * M - 2 * torch.log(torch.diagonal(Lt, dim1=-1, dim2=-2).abs()).sum())
    c1, c2 = 1.0 / (S * n), 0.013
    ref = 3.0 * (c1 * ell - c2 * kl)
    ref.backward()
    dev = [t.detach().to(dt).cuda().requires_grad_() for t in (mu, v, noise, m, Lq)]
    mu_d, v_d, n_d, m_d, L_d = dev
    out = 3.0 * (ops.GaussEllTotalFn.apply(y.to(dt).cuda(), mu_d, v_d, n_d, c1) + ops.KlWhitenedTotalFn.apply(m_d, L_d, -c2))
    out.backward()
    tol = dict(rtol=1e-10, atol=1e-12) if dt == F64 else dict(rtol=2e-4, atol=2e-6)
    assert abs(float(out) - float(ref)) < (1e-10 if dt == F64 else 2e-4) * abs(float(ref))
    for got, want in zip(dev, ref_in):
        w = want.grad if want is not L_r else torch.tril(want.grad)
        assert torch.allclose(got.grad.cpu().double(), w, **tol), got.shape


@pytest.mark.parametrize('N', [4096, 5120, 4100])
@pytest.mark.parametrize('ta,flag', [(False, 'A_LOWER'), (True, 'A_UPPER')])
def test_gemm_narrow_tiles_for_single_round_triangular_launches(ops, N, ta, flag):
    """M = K = 1024 with a triangular A operand and a grid that fits one round of the 512 slots takes the 128x64 tile
    variant (three workgroups per CU): the hidden layer at one GPU (n = 4096), a rank's last layer at eight (n = 5120),
    and a ragged width.  float32 against a float64 product of the same (masked) operands."""
    g = _g(5 + N)
    M = 1024
    A = torch.randn(M, M, generator=g)
    B = torch.randn(M, N, generator=g)
    junk = torch.triu(torch.full((M, M), 3.0), 1)              # the ignored triangle holds garbage
    Ain = torch.tril(A) + junk                                 # stored lower; with ta its transpose (upper) is the operand
    ref = (torch.tril(A).double().t() if ta else torch.tril(A).double()) @ B.double()
    out = ops.gemm(Ain.cuda(), B.cuda(), ta=ta, flags=getattr(ops, 'GEMM_' + flag))
    err = float((out.cpu().double() - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-6, err



@pytest.mark.parametrize('dt', [F32, F64])
@pytest.mark.parametrize('n,batch', [(64, 1), (128, 2), (200, 2), (1024, 3), (1100, 1), (2048, 1)])
def test_potrf_trtri_fused_equals_potrf_then_trtri(ops, dt, n, batch, monkeypatch):
    """nsgp_potrf_trtri (the whitening chain's entry).  n a multiple of 64 up to 2048: the inverse is accumulated inside
    the factorisation's panel launches (row-block solves + rank-64 updates of W, no separate inverse) -- equal to potrf
    followed by trtri to round-off, strict upper triangle exactly zero; other sizes (and NSGP_POTRF_INV=0): potrf without
    the write-back pass + the recursive inverse, bit for bit the two-call result.  info as by potrf (LAPACK convention)."""
    g = _g(60 + n)
    A = torch.randn(batch, n, n, generator=g, dtype=torch.float64)
    K = (A @ A.transpose(-1, -2) / n + torch.eye(n, dtype=torch.float64)).to(dt).cuda()
    L, info0 = ops.potrf(K)
    X0 = ops.trtri(L)
    X1, info1 = ops.potrf_trtri_(K.clone())
    assert info0.tolist() == [0] * batch and info1.tolist() == [0] * batch
    tol = 1e-13 if dt == F64 else 2e-5
    assert float((torch.tril(X1) - torch.tril(X0)).abs().max() / X0.abs().max()) < tol
    if n % 64 == 0:
        assert bool((torch.triu(X1, 1) == 0).all())
    eye = torch.eye(n, dtype=torch.float64)
    res = (torch.tril(X1).cpu().double() @ torch.tril(L).cpu().double() - eye).abs().max()
    assert float(res) < (1e-11 if dt == F64 else 2e-3)
    monkeypatch.setenv('NSGP_POTRF_INV', '0')
    X2, _ = ops.potrf_trtri_(K.clone())
    assert torch.equal(torch.tril(X2), torch.tril(X0))
    monkeypatch.delenv('NSGP_POTRF_INV')
    bad = K.clone()
    bad[0, 5, 5] = -1.0
    _, info2 = ops.potrf_trtri_(bad)
    assert int(info2[0]) == 6


@pytest.mark.parametrize('n,batch', [(64, 1), (128, 2), (200, 2), (1024, 3), (1100, 1), (2048, 1), (2112, 1)])
def test_potrf_trtri_w32_variant_writes_the_rounded_inverse(ops, n, batch):
    """nsgp_potrf_trtri_f64_w32 -- what every float32 model's whitening chain calls (WhitenFn, want_f32=True): the float32
    copy of W is written by the panel launches themselves (n a multiple of 64 up to 2048), strict upper triangle included,
    and by a cast otherwise (n = 200, 1100, 2112: the wrote32 = 0 fallback).  Must equal the float64 result rounded once."""
    g = _g(70 + n)
    A = torch.randn(batch, n, n, generator=g, dtype=torch.float64)
    K = (A @ A.transpose(-1, -2) / n + torch.eye(n, dtype=torch.float64)).cuda()
    X, info, X32 = ops.potrf_trtri_(K.clone(), want_f32=True)
    assert info.tolist() == [0] * batch
    assert X32.dtype == torch.float32 and X32.shape == X.shape
    assert torch.equal(torch.tril(X32), torch.tril(X).float())
    if n % 64 == 0 and n <= 2048:
        assert bool((torch.triu(X32, 1) == 0).all()) and bool((torch.triu(X, 1) == 0).all())
    X0, info0 = ops.potrf_trtri_(K.clone())
    assert torch.equal(torch.tril(X0), torch.tril(X))
