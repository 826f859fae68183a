"""Round-2 parity holes:

* the PRODUCT utils.functional on CUDA tensors against the only reference-generated goldens
  (tests/golden/ref_functional.npz, written by running /root/reference/utils/functional.py:14-64);
* MatrixVariateNormalPrior.log_prob / sample_n values against oracle.psgibbs (models/latent_priors.py:27-64);
* InducingGibbsKernelST + SparseSpatioTemporal_Nonstationary objective, gradients and predict against an oracle
  restatement (models/gibbs_kernels.py:268-363, models/spatio_temporal_models.py:35-126);
* the Gibbs exact GP beyond gpytorch's max_cholesky_size (N = 1024, 2048): objective, gradient and predict
  against the float64 oracle (models/nonstationary_models.py:40-62; experiments/spatial_exp.py:99,199 is where
  the reference raises the CG iteration cap for this regime -- here it is always a Cholesky).
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


# ------------------------------------------------------------------------------------------ a2: utils.functional
def test_product_functional_on_gpu_matches_reference_goldens(golden_dir):
    """dot / t / mv / mv(invert=True) / op of the PRODUCT's utils.functional, fed CUDA tensors, against the values
    the reference's own utils/functional.py produced (float64).  mv(invert=True) runs the MI355X Cholesky + MFMA
    triangular products where the reference calls torch.linalg.solve: 1e-12 relative on a kappa ~ 10 system."""
    _need_gpu()
    import utils.functional as fn
    z = np.load(os.path.join(golden_dir, 'ref_functional.npz'))
    T = {k: torch.from_numpy(z[k]).cuda() for k in z.files}
    assert torch.allclose(fn.dot(T['v1'], T['v2']), T['dot'], rtol=1e-14, atol=1e-14)    # device summation order
    assert torch.equal(fn.t(T['A']), T['t'])
    assert torch.allclose(fn.mv(T['A'], T['b']), T['mv'], rtol=1e-13, atol=1e-13)       # MFMA f64 summation order
    assert torch.allclose(fn.mv(T['A'], T['b'], invert=True), T['mv_inv'], rtol=1e-12, atol=1e-13)
    assert torch.equal(fn.op(T['e1'], T['e2']), T['op'])
    assert torch.equal(fn.op(T['e1']), T['op_self'])
    # float32 inputs: the reference's own arithmetic for the DGP scripts
    A32, b32 = T['A'].float(), T['b'].float()
    assert torch.allclose(fn.mv(A32, b32, invert=True).double(), T['mv_inv'], rtol=2e-5, atol=1e-6)


def test_functional_mv_invert_rejects_a_non_spd_matrix():
    """The reference's torch.linalg.solve (utils/functional.py:33) accepts any invertible matrix; this path is a
    Cholesky and only ever sees K + noise I.  A matrix that is not positive definite must raise, not return garbage."""
    _need_gpu()
    import utils.functional as fn
    from nsgp.gp.utils.cholesky import NotPSDError
    A = torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=F64, device='cuda')          # indefinite
    with pytest.raises(NotPSDError):
        fn.mv(A, torch.ones(2, dtype=F64, device='cuda'), invert=True)


# ------------------------------------------------------------------------------------- a10: matrix-normal prior
def test_matrix_variate_normal_prior_values_match_oracle():
    _need_gpu()
    from models.latent_priors import MatrixVariateNormalPrior
    from oracle import kernels, psgibbs
    g = torch.Generator().manual_seed(18)
    n, d = 40, 2
    x = torch.rand(n, 2, generator=g)
    row = kernels.rbf_ard(x, x, torch.tensor([[0.6931, 0.6931]]))
    col = torch.tensor([[5.0, 0.3], [0.3, 4.0]])
    ref = psgibbs.MatrixNormalPrior(torch.zeros(n, d), row, col)
    pr = MatrixVariateNormalPrior(torch.zeros(n, d).cuda(), row.cuda(), col.cuda())
    H = torch.randn(n, d, generator=g)
    lp = pr.log_prob(H.cuda())
    lp_ref = ref.log_prob(H)
    assert abs(float(lp) - float(lp_ref)) < 1e-8 * abs(float(lp_ref)), (float(lp), float(lp_ref))
    # column-stacking order of log_prob vs the row-major Kronecker covariance (the reference's inconsistency, kept):
    # a matrix and its "other vec order" twin get different densities
    assert abs(float(pr.log_prob(H.cuda())) - float(pr.log_prob(H.T.reshape(n, d).cuda()))) > 1e-3
    assert torch.allclose(pr.kron_cov.cpu().double(), ref.kron_cov, rtol=1e-6, atol=1e-7)
    # kron_cov_inv = kron(col^-1, (row + 1e-5 I)^-1): the reference inverts the float32 row covariance in float32
    # (kappa ~ 1e5: a few 1e-3 of its own largest entry off); this path inverts it with the float64 MFMA Cholesky and
    # rounds once.  Held to the float64 inverse at float32 round-off, and to the reference-style float32 inverse at that
    # inverse's own accuracy.
    rowj64 = (row + 1e-5 * torch.eye(n)).double()       # the jitter is added in float32, as the reference does (:45)
    inv64 = torch.kron(torch.linalg.inv(col.double()), torch.linalg.inv(rowj64))
    scale = float(inv64.abs().max())
    assert float((pr.kron_cov_inv.cpu().double() - inv64).abs().max()) < 1e-5 * scale
    assert float((ref.kron_cov_inv.double() - inv64).abs().max()) < 2e-2 * scale
    # sample_n: shape, and first two moments of many draws against the Kronecker covariance
    torch.manual_seed(0)
    S = torch.stack([pr.sample_n(1).reshape(-1) for _ in range(64)])
    assert pr.sample_n(1).shape == (n, d) and bool(torch.isfinite(S).all())
    draws = pr.rsample(torch.Size((20000,))).cpu().double()                  # parent-class sampler on vec(X)
    emp = torch.cov(draws.T)
    assert float((emp - ref.kron_cov).abs().max()) < 0.08 * float(ref.kron_cov.abs().max())


# ------------------------------------------------------------------------------- a5: InducingGibbsKernelST
def _uib_st_subset(data_dir):
    """The 215-point subset of experiments/spatio_temporal_exp.py:36-56 (year 2000, months 1-5; months 1-4 train)."""
    import pandas as pd
    d = pd.read_csv(os.path.join(data_dir, 'uib_spatio_temporal.csv'))
    d = d[d['time'] < 2001].copy()
    d['month'] = d['time'].rank(method='dense').astype('int')
    t = d[d['month'] < 6]
    x = torch.tensor(np.array(t)[:, 1:4], dtype=F64)
    y = torch.tensor(np.array(t)[:, -2], dtype=F64)
    stdx, meanx = torch.std_mean(x, dim=-2)
    stdy, meany = torch.std_mean(y)
    xn, yn = (x - meanx) / stdx, (y - meany) / stdy
    k = int((t['month'] < 5).sum())
    return xn[:k], yn[:k], xn[k:], yn[k:]


def test_sparse_spatiotemporal_nonstationary_matches_oracle(data_dir):
    """Objective value, every gradient and predict() of SparseSpatioTemporal_Nonstationary (InducingGibbsKernelST
    spatial part + SGPR temporal part sharing the inducing points) against oracle.spatiotemporal.st_ns_* (float64)."""
    _need_gpu()
    import nsgp.gp as gpytorch
    from models.gibbs_kernels import LogNormalPriorProcess
    from models.spatio_temporal_models import SparseSpatioTemporal_Nonstationary
    from oracle import exact, spatiotemporal as st
    xtr, ytr, xte, yte = _uib_st_subset(data_dir)
    g = torch.Generator().manual_seed(11)
    # 30 inducing points with distinct times / cells (a plain subset repeats time stamps: singular temporal Kzz)
    z = xtr[torch.randperm(len(xtr), generator=g)[:30]].clone() + 0.15 * torch.randn(30, 3, generator=g, dtype=F64)
    # the subset holds 4 distinct (train) time stamps: spread the inducing times so that the temporal Kzz (a smooth
    # RBF x Periodic kernel on 30 times) is factorable without psd_safe_cholesky's jitter ladder
    z[:, 0] = torch.linspace(-1.6, 1.6, 30, dtype=F64)[torch.randperm(30, generator=g)] + 0.01 * torch.randn(30, generator=g, dtype=F64)
    prior = LogNormalPriorProcess(input_dim=2, active_dims=(0, 1)).double()
    prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
    prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
    for p_ in prior.parameters():
        p_.requires_grad = False
    oprior = exact.LogNormalPrior(torch.full((2,), math.log(0.3), dtype=F64), torch.full((2, 2), 1.3, dtype=F64),
                                  torch.full((2,), math.log(2.0), dtype=F64))            # outputscale softplus(0)
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SparseSpatioTemporal_Nonstationary(xtr, ytr, lik, prior, z, num_dim=2).double().cuda()
    with torch.no_grad():
        model.log_ell_z.add_(0.1 * torch.randn(2, 30, generator=g, dtype=F64).cuda())
        lik.noise = 0.2
        tkk = model.temporal_covar_module.base_kernel.base_kernel.kernels
        tkk[0].lengthscale = 0.12 * torch.ones_like(tkk[0].lengthscale)        # kappa(temporal Kzz) ~ 1e3
        tkk[1].lengthscale = 0.8 * torch.ones_like(tkk[1].lengthscale)
        tkk[1].period_length = 0.9 * torch.ones_like(tkk[1].period_length)
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()

    sp = torch.nn.functional.softplus
    tk, sk = model.temporal_covar_module.base_kernel, model.spatial_covar_module
    raw = dict(os_t=tk.raw_outputscale, ls_t=tk.base_kernel.kernels[0].raw_lengthscale,
               ls_p=tk.base_kernel.kernels[1].raw_lengthscale, period=tk.base_kernel.kernels[1].raw_period_length,
               os_s=sk.raw_outputscale, noise=lik.noise_covar.raw_noise)
    lv = {k: v.detach().cpu().double().clone().requires_grad_() for k, v in raw.items()}
    p = dict(os_t=sp(lv['os_t']) + 7.0, ls_t=sp(lv['ls_t']).reshape(()), ls_p=sp(lv['ls_p']).reshape(()),
             period=sp(lv['period']).reshape(()), os_s=sp(lv['os_s']))
    noise = sp(lv['noise']).reshape(()) + 1e-4
    zo = sk.base_kernel.inducing_points.detach().cpu().double().clone().requires_grad_()
    le = model.log_ell_z.detach().cpu().clone().requires_grad_()
    ref = st.st_ns_mll(xtr, ytr, zo, le, p, noise, oprior)
    ref.backward()
    assert abs(float(val) - float(ref)) < 1e-8 * abs(float(ref)) + 1e-10, (float(val), float(ref))
    assert torch.allclose(model.log_ell_z.grad.cpu(), le.grad, rtol=1e-5, atol=1e-8)
    # the spatial kernel's inducing points are the trainable copy; the temporal kernel's are frozen (:43), so the
    # oracle's gradient through the temporal columns must be excluded: compare the (lon, lat) columns, and the time
    # column against the prior's contribution only (the prior sees columns (time, lon))
    gz = sk.base_kernel.inducing_points.grad.cpu()
    zo2 = zo.detach().clone().requires_grad_()
    p_det = {k: v.detach() for k, v in p.items()}
    # gradient with the temporal inducing locations held fixed
    zt_fixed = zo.detach()[:, 0:1]

    def mll_frozen_t(zvar):
        n = xtr.shape[-2]
        from oracle.sparse import sgpr_root
        xt = xtr[:, 0:1]
        Kt_zz = st._temporal_kernel(zt_fixed, zt_fixed, p_det)
        root_t = st._temporal_kernel(xt, zt_fixed, p_det) @ st._inv_root(Kt_zz)
        root_s, _ = sgpr_root(xtr[:, 1:3], zvar[:, 1:3], torch.exp(le.detach()), oprior)
        Qt, Qs = root_t @ root_t.T, root_s @ root_s.T
        cov = Qt + p_det['os_s'] * Qs + noise.detach() * torch.eye(n, dtype=F64)
        lp = exact.mvn_log_prob(ytr, torch.zeros_like(ytr), cov)
        lp = lp - 0.5 * ((p_det['os_t'] - torch.diagonal(Qt)) / noise.detach()).sum()
        lp = lp - 0.5 * ((1.0 - torch.diagonal(Qs)) / noise.detach()).sum()
        lp = lp + oprior.log_prob(zvar[:, 0:2], le.detach()).sum()
        return lp / n
    mll_frozen_t(zo2).backward()
    assert torch.allclose(gz, zo2.grad, rtol=1e-5, atol=1e-8), float((gz - zo2.grad).abs().max())
    assert model.temporal_covar_module.inducing_points.grad is None
    for k, v in raw.items():
        got, want = v.grad.detach().cpu().double().reshape(-1), lv[k].grad.reshape(-1)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-8), (k, got, want)

    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model.predict(xte.cuda())
        m_ref, c_ref = st.st_ns_predict(xtr, ytr, zo.detach(), le.detach(), p_det, noise.detach(), oprior, xte)
    assert pred.loc.shape == (43,)
    assert torch.allclose(pred.loc.cpu(), m_ref, rtol=1e-6, atol=1e-7 * float(m_ref.abs().max()))
    assert torch.allclose(pred.covariance_matrix.cpu(), c_ref, rtol=1e-5, atol=1e-6 * float(c_ref.abs().max()))


# ------------------------------------------------------------------------------------ f.2: exact GP at N > 800
def _lattice(n_side, n_test, seed):
    g = torch.Generator().manual_seed(seed)
    u = torch.linspace(-1.7, 1.7, n_side, dtype=F64)
    x = torch.stack(torch.meshgrid(u, u, indexing='ij'), -1).reshape(-1, 2)
    x = x + 0.01 * torch.randn(x.shape, generator=g, dtype=F64)
    f = torch.sin(2.0 * x[:, 0]) * torch.cos(1.5 * x[:, 1]) + 0.3 * x[:, 0]
    y = f + 0.1 * torch.randn(len(x), generator=g, dtype=F64)
    xs = 3.4 * torch.rand(n_test, 2, generator=g, dtype=F64) - 1.7
    return x, y, xs


@pytest.mark.parametrize('n_side', [32, 45])                        # N = 1024, 2025
def test_gibbs_exact_gp_beyond_max_cholesky_size_matches_oracle(n_side):
    """DiagonalExactGP at N = 1024 / 2025 (> gpytorch's max_cholesky_size 800, where the reference switches to CG +
    stochastic Lanczos): MLL, d MLL / d log ell and predict() against the float64 oracle, which takes the Cholesky
    route like this path.  kappa(K + 0.011 I) ~ 1e4-1e5; float64 MFMA Cholesky: 1e-7 relative on the objective,
    1e-7 on the posterior mean."""
    _need_gpu()
    import nsgp.gp as gpytorch
    from models.gibbs_kernels import LogNormalPriorProcess
    from models.nonstationary_models import DiagonalExactGP
    from oracle import exact
    x, y, xs = _lattice(n_side, 64, seed=n_side)
    N = len(x)
    prior = LogNormalPriorProcess(input_dim=2)
    prior.covar_module.outputscale = 1.0 * torch.ones_like(prior.covar_module.outputscale)
    prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
    prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
    for p_ in prior.parameters():
        p_.requires_grad = False
    oprior = exact.LogNormalPrior(torch.full((2,), math.log(0.3), dtype=F64), torch.full((2, 2), 1.3, dtype=F64),
                                  torch.ones(2, dtype=F64))
    lik = gpytorch.likelihoods.GaussianLikelihood().double()
    model = DiagonalExactGP(x, y, lik, prior, num_dim=2).to('cuda').double()
    model.likelihood.noise = 0.011
    model.covar_module.outputscale = 0.644
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        model.log_ell_train_x.add_(0.2 * torch.randn(2, N, generator=g, dtype=F64).cuda())
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    log_ell = model.log_ell_train_x.detach().cpu().clone().requires_grad_()
    ref = exact.gibbs_exact_mll(x, y, log_ell, 0.644, 0.011, oprior)
    ref.backward()
    assert abs(float(val) - float(ref)) < 1e-7 * abs(float(ref)), (float(val), float(ref))
    gerr = float((model.log_ell_train_x.grad.cpu() - log_ell.grad).abs().max() / log_ell.grad.abs().max())
    assert gerr < 1e-6, gerr
    model.eval()
    with torch.no_grad():
        pred = model.predict(xs.cuda())
        mu_r, cov_r, _ = exact.gibbs_exact_predict(x, y, log_ell.detach(), 0.644, 0.011, oprior, xs)
    rel = float((pred.loc.cpu() - mu_r).norm() / mu_r.norm())
    assert rel < 1e-7, rel
    assert torch.allclose(torch.diagonal(pred.covariance_matrix).cpu(), torch.diagonal(cov_r), rtol=1e-5, atol=1e-8)


# ------------------------------------------------------------------------------------------ ADVICE r1 (medium)
def test_non_positive_definite_covariance_raises_instead_of_nan():
    """MultivariateNormal.log_prob (under ExactMarginalLogLikelihood, the Gibbs MAP models and
    LogNormalPriorProcess.log_prob) reads potrf's `info`: jitter retries with a NumericalWarning like
    psd_safe_cholesky, then NotPSDError -- never a silent NaN objective."""
    _need_gpu()
    import warnings
    from nsgp.gp.distributions import MultivariateNormal
    from nsgp.gp.utils.cholesky import NotPSDError, NumericalWarning
    g = torch.Generator().manual_seed(0)
    A = torch.randn(50, 50, generator=g, dtype=F64)
    K = (A @ A.T + 50 * torch.eye(50, dtype=F64)).cuda()
    y = torch.randn(50, generator=g, dtype=F64).cuda()
    zero = torch.zeros(50, dtype=F64, device='cuda')
    with pytest.raises(NotPSDError):                                 # "negative outputscale": -K is negative definite
        MultivariateNormal(zero, -K).log_prob(y)
    # barely singular: rank-deficient Gram matrix -> fixed by the 1e-8..1e-6 jitter ladder, with a warning
    R = torch.randn(50, 10, generator=g, dtype=F64).cuda()
    Ks = R @ R.T
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        lp = MultivariateNormal(zero, Ks).log_prob(R @ torch.ones(10, dtype=F64, device='cuda'))
    assert bool(torch.isfinite(lp)) and any(issubclass(x.category, NumericalWarning) for x in w)
    # a positive definite covariance: unchanged value
    lp_ok = MultivariateNormal(zero, K).log_prob(y)
    ref = torch.distributions.MultivariateNormal(zero.cpu(), K.cpu()).log_prob(y.cpu())
    assert abs(float(lp_ok) - float(ref)) < 1e-9 * abs(float(ref))


def test_harness_fit_checkpoints_and_resumes_with_fused_adam(tmp_path):
    """nsgp.harness.fit(logdir=...) with the DSVI path's FusedAdam: best.tar / final.tar carry the optimiser state
    (weights_only-loadable), and a resumed run continues bit-for-bit like an uninterrupted one."""
    _need_gpu()
    import models.dgps as dgps
    from nsgp.gp import settings
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    from nsgp.harness import fit, load_checkpoint
    from nsgp.optim import FusedAdam
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(200, 3, generator=g).cuda(), torch.randn(200, generator=g).cuda()
    eps = torch.randn(3, 200, 2, generator=g).cuda()

    def make():
        torch.manual_seed(9)
        model = dgps.DeepGP(1, (200, 3), num_inducing=32).cuda()
        with torch.no_grad():
            model(x)                                   # draws the N(0, 1e-3^2) variational-mean init once
        mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, 200))
        opt = FusedAdam(model.parameters(), lr=0.01)
        model.train()

        def loss_fn():
            with settings.num_likelihood_samples(3), settings.eps_provider(lambda s, dt, dev: eps.to(dev, dt)):
                return -mll(model(x), y)
        return model, opt, loss_fn
    model, opt, loss_fn = make()
    res = fit(model, loss_fn, opt, max_iters=6, threshold=0.0, logdir=str(tmp_path))
    assert res['iterations'] == 6
    final = torch.load(os.path.join(str(tmp_path), 'final.tar'), weights_only=True)
    assert final['optim_state']['step'] == 6 and final['i'] == 5
    ref_losses = []
    fit(model, loss_fn, opt, max_iters=3, threshold=0.0, callback=lambda i, l: ref_losses.append(l))
    model2, opt2, loss_fn2 = make()
    load_checkpoint(os.path.join(str(tmp_path), 'final.tar'), model2, opt2)
    opt2.bucket.check_homed()                          # load_state_dict copied INTO the flat buffer
    got = []
    fit(model2, loss_fn2, opt2, max_iters=3, threshold=0.0, callback=lambda i, l: got.append(l))
    assert got == pytest.approx(ref_losses, rel=1e-6)


def test_ops_refuse_tensors_of_a_device_that_is_not_current():
    _need_gpu()
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two visible GPUs')
    from nsgp import ops
    x = torch.randn(8, 2, device='cuda:1')
    with pytest.raises(ops.BackendError, match='current device'):
        ops.rbf_build(x, x, torch.ones(1, 2, device='cuda:1'), torch.ones(1, device='cuda:1'))
