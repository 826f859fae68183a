"""The int8 digit-plane projection A = W Kzx (csrc/gemm_i8.hip; settings.whiten_matmul_i8): exact int32 accumulation of 14
plane products of the float64 W and of Kzx evaluated in float64, against the float64 product formed on the host --
gpytorch's float64 triangular solve behind models/dgps.py:44-51."""
import math

import pytest
import torch

from conftest import measured

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _case(b, M, n, D, seed, shared_x=True):
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(b, M, D, generator=g)
    x = torch.randn((n, D) if shared_x else (b, n, D), generator=g)
    ls = torch.rand(b, D, generator=g) + 0.6
    os_ = torch.rand(b, generator=g) + 0.5
    m = torch.randn(b, M, generator=g)
    Zd, lsd, osd = Z.double(), ls.double(), os_.double()
    xd = x.double() if x.dim() == 3 else x.double().unsqueeze(0).expand(b, n, D)
    Kzz = osd.reshape(b, 1, 1) * torch.exp(-0.5 * (((Zd.unsqueeze(2) - Zd.unsqueeze(1)) / lsd.reshape(b, 1, 1, D)) ** 2).sum(-1)) \
        + 1e-4 * torch.eye(M, dtype=F64)
    W64 = torch.linalg.inv(torch.linalg.cholesky(Kzz))
    Kzx = osd.reshape(b, 1, 1) * torch.exp(-0.5 * (((Zd.unsqueeze(2) - xd.unsqueeze(1)) / lsd.reshape(b, 1, 1, D)) ** 2).sum(-1))
    return Z, x, ls, os_, m, W64, Kzx


@pytest.mark.parametrize('b,M,n,D,shared_x', [(1, 1024, 4096, 2, True), (2, 1024, 1000, 3, True), (3, 200, 333, 3, False),
                                               (1, 96, 50, 1, True), (2, 128, 64, 4, True)])
def test_i8_projection_matches_the_float64_product(b, M, n, D, shared_x):
    _need_gpu()
    from nsgp import ops
    Z, x, ls, os_, m, W64, Kzx = _case(b, M, n, D, 7 + M + n, shared_x)
    A_ref = W64 @ Kzx
    c = lambda t: t.cuda()
    Lq = torch.tril(0.05 * torch.randn(b, M, M, generator=torch.Generator().manual_seed(1))) + 0.3 * torch.eye(M)
    A, C, mean, var = ops.svgp_project(c(W64).float(), None, c(Lq), c(m), c(os_), base_add=1e-4, W64f=c(W64),
                                       i8_inputs=(c(Z), c(x), c(ls), c(os_)))
    print('kappa-free scale: max|W| %.3g, max sum|W||K| %.3g, max|A| %.3g' % (
        float(W64.abs().max()), float((W64.abs() @ Kzx.abs()).max()), float(A_ref.abs().max())))
    # 35-bit W digits, 28-bit K digits, pairs with a + b >= 5 dropped: a few 1e-7 of the PRODUCT SCALE sum|W||K|, which at
    # kappa ~ 1e6 is ~1e2 x max|A|; measured 9.5e-7 of max|A| at the headline shape
    scale = float((W64.abs() @ Kzx.abs()).max())
    assert measured(f'i8 A b{b} M{M} n{n} D{D}', A, A_ref, rtol=0.0, atol=2e-8 * scale + 1.2e-7 * float(A_ref.abs().max()))
    mean_ref = (A_ref * m.double().unsqueeze(-1)).sum(1)
    assert measured('i8 mean', mean, mean_ref, rtol=0.0, atol=3e-8 * scale * math.sqrt(M) + 3e-7 * float(mean_ref.abs().max()))
    # the second projection and the variance follow from the kernel's own A (float32 path here)
    C_ref = torch.tril(Lq).double().transpose(-1, -2) @ A.cpu().double()
    assert float((C.cpu().double() - C_ref).abs().max() / C_ref.abs().max()) < 2e-5
    var_ref = os_.double().reshape(b, 1) + 1e-4 + (C_ref ** 2).sum(1) - (A.cpu().double() ** 2).sum(1)
    assert measured('i8 var', var, var_ref, rtol=2e-5, atol=2e-5 * float(os_.max()))


def test_i8_projection_with_float64_second_projection_and_partials():
    """Layers that feed the next layer (settings.hidden_var_f64): C = Lq^T A on the float64-accumulating kernel and float64
    partials -- the variance carries no cancellation loss."""
    _need_gpu()
    from nsgp import ops
    b, M, n, D = 2, 1024, 4096, 3
    Z, x, ls, os_, m, W64, Kzx = _case(b, M, n, D, 99)
    c = lambda t: t.cuda()
    Lq = torch.tril(0.05 * torch.randn(b, M, M, generator=torch.Generator().manual_seed(2))) + 0.1 * torch.eye(M)
    A, C, mean, var = ops.svgp_project(c(W64).float(), None, c(Lq), c(m), c(os_), base_add=1e-4, W64f=c(W64),
                                       i8_inputs=(c(Z), c(x), c(ls), c(os_)), Lq64=c(Lq).double(), i8_planes=5)
    A_ref = W64 @ Kzx
    # five Kzx planes (35 bits below os): what is left is W's own 35-bit digits and the dropped pairs a + b >= 5 -- measured
    # 3.7e-7 of max|A| here (4 planes: 1.3e-6)
    assert measured('i8 (5 planes) A', A, A_ref, rtol=0.0, atol=8e-7 * float(A_ref.abs().max()))
    C_ref = torch.tril(Lq).double().transpose(-1, -2) @ A.cpu().double()
    var_ref = os_.double().reshape(b, 1) + 1e-4 + (C_ref ** 2).sum(1) - (A_ref ** 2).sum(1)
    scale = float((W64.abs() @ Kzx.abs()).max())
    assert measured('i8+f64 C', C, C_ref, rtol=0.0, atol=2e-7 * float(C_ref.abs().max()))
    assert measured(f'i8+f64 var (min var / os = {float((var_ref / os_.double().reshape(b, 1)).min()):.2g})', var, var_ref,
                    rtol=3e-7, atol=1e-7 * scale)


def test_model_level_switch_and_backward():
    """settings.whiten_matmul_i8 on / off through the layer: same values to float32 accuracy, gradients alike."""
    _need_gpu()
    from nsgp.gp import settings
    from nsgp.svgp import svgp_marginal
    g = torch.Generator().manual_seed(3)
    b, M, n, D = 2, 256, 500, 3
    mk = lambda *s: torch.randn(*s, generator=g).cuda()
    x, Z = mk(n, D), mk(b, M, D)
    ls, os_ = (torch.rand(b, D, generator=g) + 0.7).cuda(), (torch.rand(b, generator=g) + 0.5).cuda()
    m = mk(b, M)
    Lq = (torch.tril(0.1 * torch.randn(b, M, M, generator=g)) + torch.eye(M)).cuda()
    gm, gv = mk(b, n), mk(b, n)
    res = {}
    for on in (True, False):
        leaves = [t.clone().requires_grad_() for t in (Z, ls, os_, m, Lq)]
        with settings.whiten_matmul_i8(on):
            mean, var, _ = svgp_marginal(x, *leaves)
        ((mean * gm).sum() + (var * gv).sum()).backward()
        res[on] = (mean.detach(), var.detach(), [t.grad for t in leaves])
    # (the float64-accumulating path multiplies a float32-ROUNDED Kzx: its own error, ~4e-5 here, dominates the difference)
    assert measured('mean i8 vs f64acc', res[True][0], res[False][0], rtol=0.0, atol=2e-4 * float(res[False][0].abs().max()))
    assert measured('var i8 vs f64acc', res[True][1], res[False][1], rtol=0.0, atol=2e-4 * float(res[False][1].abs().max()))
    for a, r in zip(res[True][2], res[False][2]):
        assert float((a - r).abs().max()) < 2e-3 * float(r.abs().max()) + 1e-6


def test_plane_build_kernel_also_writes_the_float32_kzx_the_backward_reads():
    """nsgp_i8_rbf_build_f32(..., Kzx_f32): the float32 Kzx rounded from the float64 values the digits are cut from -- equal to
    the float64 oracle kernel rounded to float32, to one float32 ulp (the kernel's own exp is good to ~1 ulp of float64).  The
    layer keeps it for its backward pass (Wbar = tril(Abar Kzx^T)) instead of launching a float32 build there; the gradients of
    that path are held to the oracle by test_model_level_switch_and_backward above."""
    import torch
    from nsgp import _lib, ops
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    g = torch.Generator().manual_seed(4)
    b, M, n, D = 2, 200, 333, 3
    Z, x = torch.randn(b, M, D, generator=g), torch.randn(n, D, generator=g)
    ls, os_ = torch.rand(b, D, generator=g) + 0.6, torch.rand(b, generator=g) + 0.5
    lib = _lib.load()
    dZ, dx, dls, dos = Z.cuda(), x.cuda(), ls.cuda(), os_.cuda()
    Kd = torch.empty(int(lib.nsgp_i8_k_planes_bytes(b, M, n, 4)), dtype=torch.uint8, device='cuda')
    ksc = torch.empty(b, dtype=torch.float64, device='cuda')
    K32 = torch.full((b, M, n), float('nan'), device='cuda')
    _lib.call('nsgp_i8_rbf_build_f32', ops._p(dZ), ops._p(dx), 0, ops._p(dls), ops._p(dos), b, M, n, D, 4, ops._p(Kd), ops._p(ksc),
              ops._p(K32), ops._stream())
    d2 = (((Z.double().unsqueeze(2) - x.double().unsqueeze(0).unsqueeze(1)) / ls.double().reshape(b, 1, 1, D)) ** 2).sum(-1)
    ref = (os_.double().reshape(b, 1, 1) * torch.exp(-0.5 * d2)).float()
    assert torch.isfinite(K32).all()
    assert float((K32.cpu() - ref).abs().max()) <= 1.2e-7 * float(ref.abs().max())
