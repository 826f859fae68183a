"""GPU parity of the DSVI deep GP behind models.dgps (the reference's class surface) against the CPU
oracle: ELBO value, every parameter gradient (through the softplus constraints), predict() outputs.
The float32 model is compared with the float64 oracle; tolerances are stated at each assert."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64
# float32 DSVI step vs float64 oracle, per-parameter max-norm relative gradient error: ~3x the measured worst case
DSVI_GRAD_TOL = 6e-4          # measured worst case 1.8e-4 (fp32 Cholesky adjoint), 9e-5 with the float64 adjoint


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


class _FixedEps:
    """eps provider returning pre-drawn tensors (the same ones the oracle gets)."""

    def __init__(self, eps_list):
        self.eps, self.k = eps_list, 0

    def __call__(self, shape, dtype, device):
        e = self.eps[self.k % len(self.eps)]
        self.k += 1
        assert tuple(e.shape) == tuple(shape)
        return e.to(device=device, dtype=dtype)


def _oracle_layers(model):
    """Clone the model's raw parameters to float64 CPU leaves and build the oracle's layer dicts."""
    sp = torch.nn.functional.softplus
    leaves = {}

    def leaf(name, t):
        leaves[name] = t.detach().cpu().double().clone().requires_grad_()
        return leaves[name]

    def layer(prefix, mod, linear):
        vs = mod.variational_strategy
        p = dict(Z=leaf(prefix + 'Z', vs.inducing_points),
                 lengthscale=sp(leaf(prefix + 'raw_ls', mod.covar_module.base_kernel.raw_lengthscale)),
                 outputscale=sp(leaf(prefix + 'raw_os', mod.covar_module.raw_outputscale)),
                 m=leaf(prefix + 'm', vs._variational_distribution.variational_mean),
                 Lq=leaf(prefix + 'Lq', vs._variational_distribution.chol_variational_covar))
        if linear:
            p['mean'] = ('linear', leaf(prefix + 'w', mod.mean_module.weights), leaf(prefix + 'b', mod.mean_module.bias))
        else:
            p['mean'] = ('constant', leaf(prefix + 'c', mod.mean_module.constant))
        return p
    hidden = layer('h.', model.layers[0], True)
    last = layer('l.', model.last_layer, False)
    noise = sp(leaf('raw_noise', model.likelihood.noise_covar.raw_noise)) + 1e-4
    return hidden, last, noise, leaves


def _model_params(model):
    h, l_ = model.layers[0], model.last_layer
    return {
        'h.Z': h.variational_strategy.inducing_points,
        'h.raw_ls': h.covar_module.base_kernel.raw_lengthscale, 'h.raw_os': h.covar_module.raw_outputscale,
        'h.m': h.variational_strategy._variational_distribution.variational_mean,
        'h.Lq': h.variational_strategy._variational_distribution.chol_variational_covar,
        'h.w': h.mean_module.weights, 'h.b': h.mean_module.bias,
        'l.Z': l_.variational_strategy.inducing_points,
        'l.raw_ls': l_.covar_module.base_kernel.raw_lengthscale, 'l.raw_os': l_.covar_module.raw_outputscale,
        'l.m': l_.variational_strategy._variational_distribution.variational_mean,
        'l.Lq': l_.variational_strategy._variational_distribution.chol_variational_covar,
        'l.c': l_.mean_module.constant, 'raw_noise': model.likelihood.noise_covar.raw_noise,
    }


def _build(num_layers, D, M, seed):
    import models.dgps as m
    from nsgp.gp import settings
    torch.manual_seed(seed)
    model = m.DeepGP(num_layers, (1000, D), num_inducing=M).cuda()
    # move away from the trivial initialisation so that every gradient path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in (model.layers[0], model.last_layer):
            vd = mod.variational_strategy._variational_distribution
            vd.variational_mean.copy_(0.3 * torch.randn(vd.variational_mean.shape, generator=g))
            Lq = torch.tril(0.1 * torch.randn(vd.chol_variational_covar.shape, generator=g)) + torch.eye(M)
            vd.chol_variational_covar.copy_(Lq)
            mod.variational_strategy.variational_params_initialized.fill_(1)
            mod.covar_module.base_kernel.raw_lengthscale.add_(0.3 * torch.randn(
                mod.covar_module.base_kernel.raw_lengthscale.shape, generator=g).cuda())
    return model, settings


@pytest.mark.parametrize('chol_bwd_f64', [True, False])
@pytest.mark.parametrize('num_layers,D,M,B,S', [(1, 3, 40, 315, 3), (2, 2, 64, 128, 4), (1, 2, 130, 200, 10)])
def test_dsvi_elbo_and_gradients_match_oracle(num_layers, D, M, B, S, chol_bwd_f64):
    _need_gpu()
    from oracle import svgp
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    model, settings = _build(num_layers, D, M, 100 + M)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, D, generator=g)
    y = torch.randn(B, generator=g)
    eps = [torch.randn(S, B, 2, generator=g) for _ in range(num_layers)]
    N = 5000
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
    model.train()
    with settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)), \
            settings.chol_bwd_f64(chol_bwd_f64):
        out = model(x.cuda())
        elbo = mll(out, y.cuda())
        assert out.mean.shape == (S, B) and out.variance.shape == (S, B)
        elbo.backward()

    hidden, last, noise, leaves = _oracle_layers(model)
    ref = svgp.dsvi_elbo(x.double(), y.double(), hidden, last, num_layers, [e.double() for e in eps], S, noise, N)
    ref.backward()
    # float32 pipeline vs float64 oracle: ELBO (measured <= 1.6e-7 relative)
    assert abs(float(elbo) - float(ref)) < 2e-6 * abs(float(ref)) + 1e-7
    errs = {}
    for name, p in _model_params(model).items():
        got, want = p.grad.detach().cpu().double(), leaves[name].grad
        if name.endswith('Lq'):
            want = torch.tril(want)
        scale = float(want.abs().max()) + 1e-12
        err = float((got - want).abs().max()) / scale
        errs[name] = err
        assert err < DSVI_GRAD_TOL, (name, err, chol_bwd_f64)         # per-parameter max-norm relative error
    print('[measured] dsvi max grad rel err', (num_layers, D, M, B, S, chol_bwd_f64), '%.3g' % max(errs.values()), max(errs, key=errs.get),
          '| elbo rel err %.3g' % (abs(float(elbo) - float(ref)) / abs(float(ref))))


def test_negated_data_parallel_objective_is_minus_the_objective():
    """nsgp.dist.dp_objective(negate=True) (what bench.py back-propagates) == -dp_objective(), value and gradients."""
    _need_gpu()
    from nsgp.dist import dp_objective
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    model, settings = _build(1, 3, 40, 140)
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(100, 3, generator=g).cuda(), torch.randn(100, generator=g).cuda()
    eps = [torch.randn(3, 100, 2, generator=g)]
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, 1234))
    model.train()
    vals, grads = [], []
    for negate in (False, True):
        model.zero_grad()
        with settings.num_likelihood_samples(3), settings.eps_provider(_FixedEps(eps)):
            obj = dp_objective(mll, model(x), y, 100, 1, negate=negate)
            obj.backward()
        vals.append(float(obj.detach()))
        grads.append([p.grad.detach().clone() for p in model.parameters()])
    assert vals[1] == pytest.approx(-vals[0], rel=1e-6)
    for a, b in zip(*grads):
        assert torch.allclose(a, -b, rtol=1e-5, atol=1e-7)


def test_cfg5_three_layer_m2048_three_output_dims_matches_oracle():
    """BASELINE configs[4] shape at a reduced minibatch: DeepGP(num_layers=2) = the tied hidden layer applied twice
    + the last layer, M=2048 inducing points, 3-D inputs, so `num_output_dims` has to be 3 (the tied layer feeds
    itself).  Exercises the two-level potrf / trtri (n >= 2048) and three hidden GPs inside the DSVI step.
    float32 pipeline vs float64 oracle; 2048 random inducing points in 3-D make Kzz far worse conditioned than the
    small cases above; tolerances are stated at the asserts."""
    _need_gpu()
    import models.dgps as m
    from oracle import svgp
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    old = m.num_output_dims
    m.num_output_dims = 3
    try:
        model, settings = _build(2, 3, 2048, 2048)
    finally:
        m.num_output_dims = old
    assert model.layers[0] is model.layers[1] and model.layers[0].output_dims == 3
    B, S, N = 192, 3, 1_000_000
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, generator=g)
    y = torch.randn(B, generator=g)
    eps = [torch.randn(S, B, 3, generator=g) for _ in range(2)]
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, N))
    model.train()
    with settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)):
        out = model(x.cuda())
        elbo = mll(out, y.cuda())
        assert out.mean.shape == (S, B)
        elbo.backward()
    hidden, last, noise, leaves = _oracle_layers(model)
    ref = svgp.dsvi_elbo(x.double(), y.double(), hidden, last, 2, [e.double() for e in eps], S, noise, N)
    ref.backward()
    rel = abs(float(elbo.detach()) - float(ref.detach())) / abs(float(ref.detach()))
    errs = {}
    for name, p in _model_params(model).items():
        got, want = p.grad.detach().cpu().double(), leaves[name].grad
        if name.endswith('Lq'):
            want = torch.tril(want)
        errs[name] = float((got - want).abs().max()) / (float(want.abs().max()) + 1e-12)
    print('cfg5 shape: ELBO rel err %.2e, grad rel errs' % rel, {k: '%.1e' % v for k, v in errs.items()})
    assert rel < 2e-5                                      # ELBO: 2e-5 relative (measured 8e-7)
    assert max(errs.values()) < 5e-3, errs                 # per-parameter max-norm relative gradient error (measured 9e-4)


def test_predict_matches_oracle_and_full_covariance_is_consistent():
    _need_gpu()
    from oracle import svgp
    model, settings = _build(1, 2, 48, 321)
    g = torch.Generator().manual_seed(9)
    n, S = 78, 4
    x = torch.randn(n, 2, generator=g)
    y = torch.randn(n, generator=g)
    eps = [torch.randn(S, n, 2, generator=g)]
    model.eval()
    loader = [(x.cuda(), y.cuda())]
    with settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)):
        preds, mus, variances, lls = model.predict(loader)
        cov = preds.covariance_matrix
        lp = preds.log_prob(y.cuda())
    hidden, last, noise, _ = _oracle_layers(model)
    with torch.no_grad():
        m_ref, v_ref, ll_ref = svgp.dgp_predict(x.double(), y.double(), hidden, last, 1, [e.double() for e in eps], S,
                                                noise)
        mean_f, cov_f = svgp.dgp_forward(x.double(), hidden, last, 1, [e.double() for e in eps], S,
                                         full_cov_last=True)
    tol = dict(rtol=2e-3, atol=2e-4)
    assert torch.allclose(mus.cpu().double(), m_ref, **tol)
    assert torch.allclose(variances.cpu().double(), v_ref, **tol)
    assert torch.allclose(lls.cpu().double(), ll_ref, rtol=5e-3, atol=5e-3)
    cov_ref = cov_f + float(noise) * torch.eye(n, dtype=F64)
    assert torch.allclose(cov.cpu().double(), cov_ref, rtol=5e-3, atol=5e-4)
    assert torch.allclose(torch.diagonal(cov, dim1=-1, dim2=-2).cpu().double(), v_ref, **tol)
    lp_ref = torch.distributions.MultivariateNormal(m_ref, covariance_matrix=cov_ref).log_prob(y.double())
    assert torch.allclose(lp.cpu().double(), lp_ref, rtol=5e-3, atol=5e-2)


@pytest.mark.parametrize('grads_as_views', [True, False])
def test_fused_adam_and_flat_bucket_train_the_dgp_like_torch_adam(grads_as_views):
    _need_gpu()
    import copy
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from nsgp.optim import FusedAdam
    model_a, settings = _build(1, 3, 32, 555)
    model_b = copy.deepcopy(model_a)
    g = torch.Generator().manual_seed(11)
    B, S = 256, 3
    x, y = torch.randn(B, 3, generator=g).cuda(), torch.randn(B, generator=g).cuda()
    eps = [torch.randn(S, B, 2, generator=g)]
    opt_a = FusedAdam(model_a.parameters(), lr=0.01, grads_as_views=grads_as_views)
    opt_b = torch.optim.Adam(model_b.parameters(), lr=0.01)
    losses = []
    for model, opt in ((model_a, opt_a), (model_b, opt_b)):
        mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, 1000))
        model.train()
        ls = []
        for it in range(5):
            with settings.num_likelihood_samples(S), settings.eps_provider(_FixedEps(eps)):
                opt.zero_grad()
                loss = -mll(model(x), y)
                loss.backward()
                opt.step()
            ls.append(float(loss))
        losses.append(ls)
    assert losses[0][-1] < losses[0][0]                       # it trains
    for a, b in zip(*losses):
        assert abs(a - b) < 1e-3 * abs(b) + 1e-4              # fused flat Adam == torch.optim.Adam
    for pa, pb in zip(model_a.parameters(), model_b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('dt', [torch.float32, torch.float64])
def test_fused_dsvi_objective_equals_the_chain_of_per_term_kernels(dt):
    """nsgp_dsvi_objective_{fwd,bwd} (ops.DsviObjectiveFn: likelihood term + every layer's KL as one scalar, two launches
    each way) against the per-term kernels it replaces (GaussEllTotalFn + KlWhitenedTotalFn, themselves held to the oracle by
    the DSVI tests above): value and every gradient, two groups of different batch size, upstream gradient != 1."""
    _need_gpu()
    from nsgp import ops
    g = torch.Generator().manual_seed(21)
    S, n, M = 5, 777, 96
    tol = dict(rtol=2e-5, atol=1e-6) if dt == torch.float32 else dict(rtol=1e-12, atol=1e-13)
    mk = lambda *shp: torch.randn(*shp, generator=g, dtype=torch.float64).to(dt).cuda()
    y, mu = mk(n), mk(S, n)
    v = (mk(S, n).abs() + 0.1)
    noise = torch.tensor([0.3], dtype=dt, device='cuda')
    groups = []
    for b in (2, 1):
        m = mk(b, M) if b > 1 else mk(M)
        Lq = (torch.tril(0.1 * mk(b, M, M)) + torch.eye(M, dtype=dt, device='cuda')) if b > 1 else \
            (torch.tril(0.1 * mk(M, M)) + torch.eye(M, dtype=dt, device='cuda'))
        groups.append((m, Lq))

    def run(fused):
        leaves = [t.clone().requires_grad_() for t in (mu, v, noise)] + [t.clone().requires_grad_() for pr in groups for t in pr]
        mu_, v_, nz_, *mL = leaves
        if fused:
            tot = ops.DsviObjectiveFn.apply(y, mu_, v_, nz_, 0.37, -0.011, *mL)
        else:
            tot = ops.GaussEllTotalFn.apply(y, mu_, v_, nz_, 0.37)
            for m_, L_ in zip(mL[0::2], mL[1::2]):
                tot = ops.KlWhitenedTotalFn.apply(m_, L_, -0.011, tot)
        (tot * 1.7).backward()
        return tot.detach(), [t.grad for t in leaves]
    a, ga = run(True)
    b_, gb = run(False)
    assert torch.allclose(a, b_, **tol), (float(a), float(b_))
    for x, z in zip(ga, gb):
        assert x.shape == z.shape and torch.allclose(x, z, **tol)
    assert bool((torch.triu(ga[4], 1) == 0).all())             # strict upper triangle of the Lq gradients is zero
