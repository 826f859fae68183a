"""BASELINE configs[4]'s "bf16 forward" (settings.forward_precision('bf16'); csrc/gemm_bf16.hip): the two forward
projections A = W Kzx, C = Lq^T A of a whitened SVGP layer (models/dgps.py:44-51,92-98 through gpytorch's
VariationalStrategy.forward) on the bf16 matrix cores.

* kernel level: against the SAME products formed in float64 from the bf16-rounded operands (so only the float32
  accumulation order differs), ragged sizes included;
* model level at the configs[4] shape (3-layer = tied hidden layer twice + last, M = 2048, three output dims, reduced
  minibatch): ELBO and last-layer mean against the float64 oracle -- the ACHIEVED error of the bf16 mode is printed and
  bounded loosely (it is a throughput mode: operands carry 8 bits of mantissa, and W = L^-1 has entries ~1e2 whose
  products cancel); gradients (float32 backward on the bf16 forward's A / C) must stay finite and close to float32's."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


@pytest.mark.parametrize('b,M,n,D,shared_x', [(1, 256, 512, 2, True), (2, 200, 333, 3, True), (3, 128, 64, 3, False),
                                               (1, 1024, 4096, 2, True)])
def test_bf16_projection_kernels_match_products_of_the_rounded_operands(b, M, n, D, shared_x):
    _need_gpu()
    from nsgp import ops
    g = torch.Generator().manual_seed(5 + M + n)
    Z = torch.randn(b, M, D, generator=g)
    x = torch.randn((n, D) if shared_x else (b, n, D), generator=g)
    ls = torch.rand(b, D, generator=g) + 0.6
    os_ = torch.rand(b, generator=g) + 0.5
    W = torch.tril(torch.randn(b, M, M, generator=g)) / math.sqrt(M)
    Lq = torch.tril(torch.randn(b, M, M, generator=g)) / math.sqrt(M) + torch.eye(M)
    m = torch.randn(b, M, generator=g)
    c = lambda t: t.cuda()
    Kzx32 = ops.rbf_build(c(Z), c(x), c(ls), c(os_))
    A, C, mean, var = ops.svgp_project_bf16(c(W), Kzx32, c(Lq), c(m), c(os_), base_add=1e-4,
                                            kernel_inputs=(c(Z), c(x), c(ls), c(os_)))
    # reference: float64 products of the bf16-rounded operands
    r = lambda t: t.to(torch.bfloat16).double()
    xb = x if x.dim() == 3 else x.unsqueeze(0).expand(b, n, D)
    d2 = (((Z.unsqueeze(2) - xb.unsqueeze(1)) / ls.reshape(b, 1, 1, D)) ** 2).sum(-1)          # (b, M, n)
    Kzx = r((os_.reshape(b, 1, 1) * torch.exp(-0.5 * d2)).float())
    A_ref = r(W) @ Kzx
    # the kernel's exp is the hardware exp2 path: a bf16 ulp (4e-3) on a few entries of Kzx is expected
    sa = float(A_ref.abs().max())
    assert float((A.cpu().double() - A_ref).abs().max()) < 2e-2 * sa
    # product 2 consumes the bf16 copy of the kernel's OWN A: rebuild it from the returned A
    C_ref = r(torch.tril(Lq)).transpose(-1, -2) @ r(A.cpu())
    sc = float(C_ref.abs().max())
    assert float((C.cpu().double() - C_ref).abs().max()) < 2e-4 * sc + 1e-5
    mean_ref = (A.cpu().double() * m.double().unsqueeze(-1)).sum(1)
    var_ref = os_.double().reshape(b, 1) + 1e-4 + (C.cpu().double() ** 2).sum(1) - (A.cpu().double() ** 2).sum(1)
    assert torch.allclose(mean.cpu().double(), mean_ref, rtol=1e-4, atol=1e-4 * float(mean_ref.abs().max()))
    assert torch.allclose(var.cpu().double(), var_ref, rtol=1e-4, atol=1e-4 * float(var_ref.abs().max()))
    # mode 'bf16' (product 2 only): A and the mean are the float32 path's, C is the bf16 product of the rounded operands
    A32, C32, mean32, var32 = ops.svgp_project(c(W), Kzx32, c(Lq), c(m), c(os_), base_add=1e-4)
    A2, C2, mean2, var2 = ops.svgp_project_bf16(c(W), Kzx32, c(Lq), c(m), c(os_), base_add=1e-4)
    assert torch.equal(A2, A32) and torch.allclose(mean2, mean32, rtol=1e-6, atol=1e-6)
    C2_ref = r(torch.tril(Lq)).transpose(-1, -2) @ r(A32.cpu())
    assert float((C2.cpu().double() - C2_ref).abs().max()) < 2e-4 * float(C2_ref.abs().max()) + 1e-5
    assert float((C2 - C32).abs().max()) < 3e-2 * float(C32.abs().max())           # bf16 rounding of both operands


def test_bf16_forward_at_the_cfg5_shape_states_its_error():
    _need_gpu()
    import models.dgps as m
    from oracle import svgp
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from test_gpu_dgp import _FixedEps, _build, _model_params, _oracle_layers
    old = m.num_output_dims
    m.num_output_dims = 3
    try:
        model, settings = _build(2, 3, 2048, 2048)
    finally:
        m.num_output_dims = old
    B, S, N = 192, 3, 1_000_000
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(B, 3, generator=g), torch.randn(B, generator=g)
    eps = [torch.randn(S, B, 3, generator=g) for _ in range(2)]
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
    model.train()
    res = {}
    for mode in ('f32', 'bf16', 'bf16_all'):
        model.zero_grad()
        with settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)), settings.forward_precision(mode):
            out = model(x.cuda())
            elbo = mll(out, y.cuda())
            elbo.backward()
        res[mode] = (float(elbo.detach()), out.mean.detach().cpu().double(),
                     {k: p.grad.detach().cpu().double().clone() for k, p in _model_params(model).items()})
    hidden, last, noise, _ = _oracle_layers(model)
    with torch.no_grad():
        om, ov = svgp.dgp_forward(x.double(), hidden, last, 2, [e.double() for e in eps], S)
        ref = svgp.dsvi_elbo(x.double(), y.double(), hidden, last, 2, [e.double() for e in eps], S, noise, N)
    err = {k: float((res[k][1] - om).abs().max() / om.abs().max()) for k in res}
    eerr = {k: abs(res[k][0] - float(ref)) / abs(float(ref)) for k in res}
    gerr = {k: max(float((res[k][2][q] - res['f32'][2][q]).abs().max() / (res['f32'][2][q].abs().max() + 1e-12))
                   for q in res['f32'][2]) for k in ('bf16', 'bf16_all')}
    print('configs[4] shape (M=2048, 3 layers), output mean max-norm rel err vs float64 oracle: '
          + ', '.join('%s %.2e' % (k, err[k]) for k in res) + '; ELBO rel err: '
          + ', '.join('%s %.2e' % (k, eerr[k]) for k in res) + '; gradient max-norm rel diff vs f32: '
          + ', '.join('%s %.2e' % (k, gerr[k]) for k in gerr))
    assert err['f32'] < 1e-3
    # 'bf16' (C = Lq^T A on the bf16 cores): the mean of every layer is computed from the full-precision A; what moves
    # is the variance, hence the samples fed to the next layer
    assert err['bf16'] < 5e-2 and eerr['bf16'] < 2e-2, (err, eerr)
    assert all(bool(torch.isfinite(v).all()) for v in res['bf16'][2].values())
    # 'bf16_all' is a throughput figure: stated above, only required to be finite
    assert math.isfinite(res['bf16_all'][0])


@pytest.mark.parametrize('b,M,n', [(2, 1024, 1950), (1, 1024, 4032), (1, 1024, 512)])
def test_bf16_mode_with_f64_whitening_at_shapes_where_the_kernels_tile_rows_differ(b, M, n):
    """forward_precision('bf16') with the float64-accumulating whitening product.  At these shapes the float32
    plan uses 128-row tiles (8 tile rows) but the float64-accumulating kernel picks 64-row tiles (16 tile rows): the
    partial buffers must hold the larger count (the call used to fail with -10 / BackendError)."""
    _need_gpu()
    from nsgp import ops
    g = torch.Generator().manual_seed(M + n + b)
    D = 3
    Z = torch.randn(b, M, D, generator=g)
    x = torch.randn(n, D, generator=g)
    ls = torch.rand(b, D, generator=g) + 0.6
    os_ = torch.rand(b, generator=g) + 0.5
    W = torch.tril(torch.randn(b, M, M, generator=g)) / math.sqrt(M)
    Lq = torch.tril(torch.randn(b, M, M, generator=g)) / math.sqrt(M) + torch.eye(M)
    m = torch.randn(b, M, generator=g)
    c = lambda t: t.cuda()
    Kzx = ops.rbf_build(c(Z), c(x), c(ls), c(os_))
    W64 = c(W).double()
    A0, C0, mean0, var0 = ops.svgp_project(c(W), Kzx, c(Lq), c(m), c(os_), base_add=1e-4, W64f=W64)
    A1, C1, mean1, var1 = ops.svgp_project_bf16(c(W), Kzx, c(Lq), c(m), c(os_), base_add=1e-4, W64f=W64)
    assert torch.equal(A1, A0)                                       # product 1 is the same launch
    assert torch.allclose(mean1, mean0, rtol=1e-6, atol=1e-6)
    r = lambda t: t.to(torch.bfloat16).double()
    C_ref = r(torch.tril(Lq)).transpose(-1, -2) @ r(A0.cpu())
    assert float((C1.cpu().double() - C_ref).abs().max()) < 2e-4 * float(C_ref.abs().max()) + 1e-5
    var_ref = os_.double().reshape(b, 1) + 1e-4 + (C1.cpu().double() ** 2).sum(1) - (A1.cpu().double() ** 2).sum(1)
    assert torch.allclose(var1.cpu().double(), var_ref, rtol=1e-4, atol=1e-4 * float(var_ref.abs().max()))
    # and without W64f (float32 product 1)
    A2, C2, mean2, var2 = ops.svgp_project_bf16(c(W), Kzx, c(Lq), c(m), c(os_), base_add=1e-4)
    A3, _, mean3, _ = ops.svgp_project(c(W), Kzx, c(Lq), c(m), c(os_), base_add=1e-4)
    assert torch.equal(A2, A3) and torch.allclose(mean2, mean3, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('forward', ['f32', 'bf16'])
def test_configs4_full_shape_step_is_finite_decreases_the_loss_and_fits(forward):
    """BASELINE configs[4] at its FULL per-GPU shape (3-layer, M = 2048, minibatch 4096, S = 10,
    N = 1e6) through `bench.py --config cfg5`'s own runner (tools/dsvi_cfg5_probe.py: hipGraph-captured
    fwd + ELBO + bwd + Adam).  The oracle cannot run this size in test time, so: size-independent properties -- every
    loss finite, Adam decreases the objective over a few steps, peak HBM stays far below the 288 GB of the card."""
    _need_gpu()
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tools'))
    import dsvi_cfg5_probe as probe
    import models.dgps as m
    old = m.num_output_dims
    try:
        r = probe.run(probe.parser().parse_args(['--steps', '8', '--warmup', '2', '--forward', forward]))
    finally:
        m.num_output_dims = old
    print('configs[4] full shape, forward=%s: %.1f ms/step, loss %.4f -> %.4f, peak HBM %.2f GB, f32 GEMMs %.1f TFLOP/s'
          % (forward, r['ms_per_step'], r['loss_first'], r['loss_last'], r['hbm_peak_allocated_GB'], r['f32_gemm_TFLOPs']))
    assert math.isfinite(r['loss_first']) and math.isfinite(r['loss_last'])
    assert r['loss_last'] < r['loss_first']
    assert 1.0 < r['hbm_peak_allocated_GB'] < 40.0
    assert r['f32_gemm_launches'] > 0 and (r['f64acc_gemm_launches'] + r['i8_gemm_launches'] > 0)
    if forward == 'bf16':
        assert r['bf16_gemm_launches'] > 0
