"""Pins the CPU oracle: reference-generated goldens where the reference imports here
(utils/functional.py, utils/dataprep.py), known-answer identities + sklearn/scipy elsewhere
(gpytorch-dependent parts are "parity unpinned" by reference artefacts -- oracle/__init__.py)."""
import math
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import kernels as K, exact, sparse, svgp, psgibbs, functional as fn, dataprep as dp

torch.set_default_dtype(torch.float32)
F64 = torch.float64


def _g(seed=173):
    return torch.Generator().manual_seed(seed)


# ---------------------------------------------------------------- reference-generated goldens
def test_functional_matches_reference_goldens(golden_dir):
    z = np.load(os.path.join(golden_dir, 'ref_functional.npz'))
    T = {k: torch.from_numpy(z[k]) for k in z.files}
    assert torch.equal(fn.dot(T['v1'], T['v2']), T['dot'])
    assert torch.equal(fn.t(T['A']), T['t'])
    assert torch.equal(fn.mv(T['A'], T['b']), T['mv'])
    assert torch.allclose(fn.mv(T['A'], T['b'], invert=True), T['mv_inv'], rtol=0, atol=1e-15)
    assert torch.equal(fn.op(T['e1'], T['e2']), T['op'])
    assert torch.equal(fn.op(T['e1']), T['op_self'])


@pytest.mark.parametrize('name', ['uib_spatial', 'khyber_time_series'])
def test_dataprep_matches_reference_goldens(golden_dir, data_dir, name):
    z = np.load(os.path.join(golden_dir, 'ref_dataprep.npz'))
    data = dp.download_data(os.path.join(data_dir, name + '.csv'))
    x, y, mx, sx, my, sy = dp.whitening_transform(data)
    for got, key in ((x, 'x'), (y, 'y'), (mx, 'meanx'), (sx, 'stdx'), (my, 'meany'), (sy, 'stdy')):
        assert torch.equal(got, torch.from_numpy(z[f'{name}_{key}'])), key
    trx, try_, tex, tey = dp.train_test_split(x, y, 0.8)
    assert len(trx) == int(z[f'{name}_ntrain'])
    assert torch.equal(trx[-3:], torch.from_numpy(z[f'{name}_train_x_tail']))
    assert torch.equal(tey[:3], torch.from_numpy(z[f'{name}_test_y_head']))


def test_bundled_data_shapes(data_dir):
    # 342 / 394 / 5676 rows
    assert dp.download_data(os.path.join(data_dir, 'khyber_time_series.csv')).shape == (342, 2)
    assert dp.download_data(os.path.join(data_dir, 'uib_spatial.csv')).shape == (394, 3)
    assert dp.download_data(os.path.join(data_dir, 'uib_spatio_temporal.csv')).shape == (5676, 5)


# ---------------------------------------------------------------- kernels: known answers
def test_gibbs_matches_scalar_formula_and_limits():
    g = _g()
    x1 = torch.randn(13, 2, generator=g, dtype=F64)
    x2 = torch.randn(9, 2, generator=g, dtype=F64)
    e1 = torch.exp(0.3 * torch.randn(2, 13, generator=g, dtype=F64) + math.log(0.3))
    e2 = torch.exp(0.3 * torch.randn(2, 9, generator=g, dtype=F64) + math.log(0.3))
    Kg = K.gibbs(x1, x2, e1, e2)
    for i, j in ((0, 0), (3, 7), (12, 8)):
        assert abs(float(Kg[i, j]) - K.gibbs_scalar(x1, x2, e1, e2, i, j)) < 1e-15
    # self-covariance: unit diagonal, symmetric, PSD
    Kxx = K.gibbs(x1, x1, e1, e1)
    assert torch.allclose(torch.diagonal(Kxx), torch.ones(13, dtype=F64), atol=1e-15)
    assert torch.allclose(Kxx, Kxx.T, atol=1e-15)
    assert torch.linalg.eigvalsh(Kxx).min() > -1e-10
    # constant lengthscale == RBF with that lengthscale
    l0 = 0.37
    c1 = torch.full((2, 13), l0, dtype=F64)
    c2 = torch.full((2, 9), l0, dtype=F64)
    ref = K.rbf_ard(x1, x2, torch.full((1, 2), l0, dtype=F64))
    assert torch.allclose(K.gibbs(x1, x2, c1, c2), ref, atol=1e-14)


def test_rbf_matches_sklearn():
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel
    g = _g(1)
    x1 = torch.randn(11, 3, generator=g, dtype=F64)
    x2 = torch.randn(6, 3, generator=g, dtype=F64)
    ls = torch.tensor([[0.5, 1.3, 0.9]], dtype=F64)
    ref = (ConstantKernel(0.644) * RBF(ls[0].numpy()))(x1.numpy(), x2.numpy())
    assert np.allclose(K.rbf_ard(x1, x2, ls, 0.644).numpy(), ref, atol=1e-14)


def test_ps2d_constant_sigma_is_anisotropic_rbf_without_half():
    g = _g(2)
    x1 = torch.randn(7, 2, generator=g, dtype=F64)
    x2 = torch.randn(5, 2, generator=g, dtype=F64)
    S = torch.tensor([[0.8, 0.2], [0.2, 0.5]], dtype=F64)
    s1, s2 = S.expand(7, 2, 2), S.expand(5, 2, 2)
    got = K.ps2d(x1, x2, s1, s2, jitter=1e-5)
    inv = torch.inverse(S + 1e-5 * torch.eye(2, dtype=F64))
    d = x1[:, None, :] - x2[None, :, :]
    ref = torch.exp(-torch.einsum('ijk,kl,ijl->ij', d, inv, d))
    assert torch.allclose(got, ref, atol=1e-14)


def test_ps_sigma_matches_reference_python_loop():
    # models/multivariate_gibbs_kernel.py:98 builds Sigma_i row by row through numpy
    g = _g(3)
    H = torch.randn(6, 2, generator=g)
    D = torch.diag(torch.randn(2, generator=g))
    loop = torch.nn.functional.softplus(
        torch.Tensor(np.array([np.outer(h, h.T) ** 2 for h in H.numpy()]))) + D ** 2
    assert torch.allclose(K.ps_sigma(H, D), loop, atol=1e-7)


# ---------------------------------------------------------------- lognormal prior process
def _prior(D=2, Din=2, dtype=F64):
    return exact.LogNormalPrior(torch.full((D,), math.log(0.3), dtype=dtype),
                                torch.full((D, Din), 1.3, dtype=dtype),
                                torch.full((D,), 1.0, dtype=dtype))


def test_conditional_mean_at_given_points_returns_given_values():
    g = _g(4)
    xg = torch.randn(20, 2, generator=g, dtype=F64)
    ell_g = torch.exp(0.2 * torch.randn(2, 20, generator=g, dtype=F64) + math.log(0.3))
    pr = exact.LogNormalPrior(torch.full((2,), math.log(0.3), dtype=F64),
                              torch.full((2, 2), 0.3, dtype=F64), torch.ones(2, dtype=F64))
    got = pr.conditional_mean_ell(xg, xg, ell_g)
    assert got.shape == (2, 20)
    assert torch.allclose(got, ell_g, rtol=2e-3)       # exact up to the 1e-4 jitter


def test_prior_log_prob_matches_torch_mvn():
    g = _g(5)
    x = torch.randn(15, 2, generator=g, dtype=F64)
    le = torch.randn(2, 15, generator=g, dtype=F64) * 0.1 + math.log(0.3)
    pr = _prior()
    cov = pr.cov(x, x) + 1e-4 * torch.eye(15, dtype=F64)
    ref = torch.distributions.MultivariateNormal(pr.mean(x), covariance_matrix=cov).log_prob(le) / 15
    assert torch.allclose(pr.log_prob(x, le), ref, atol=1e-10)


# ---------------------------------------------------------------- exact GP vs sklearn / scipy
def test_seard_exact_gp_matches_sklearn(data_dir):
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, WhiteKernel
    data = dp.download_data(os.path.join(data_dir, 'uib_spatial.csv')).double()
    x, y, *_ = dp.whitening_transform(data)
    trx, try_, tex, tey = dp.train_test_split(x, y, 0.8)
    ls = torch.tensor([[0.7, 0.9]], dtype=F64)
    os_, noise = 0.644, 0.05
    kern = ConstantKernel(os_, 'fixed') * RBF(ls[0].numpy(), 'fixed') + WhiteKernel(noise, 'fixed')
    gpr = GaussianProcessRegressor(kern, alpha=0.0, optimizer=None).fit(trx.numpy(), try_.numpy())
    c0 = torch.zeros(1, dtype=F64)
    mll = exact.seard_mll(trx, try_, ls, os_, noise, c0)
    assert abs(float(mll) * len(trx) - gpr.log_marginal_likelihood_value_) < 1e-7
    mean, cov = exact.seard_predict(trx, try_, ls, os_, noise, c0, tex, with_noise=True)
    m_ref, c_ref = gpr.predict(tex.numpy(), return_cov=True)
    assert np.allclose(mean.numpy(), m_ref, atol=1e-8)
    assert np.allclose(cov.numpy(), c_ref, atol=1e-8)


def test_mvn_log_prob_matches_scipy():
    from scipy.stats import multivariate_normal
    g = _g(6)
    A = torch.randn(12, 12, generator=g, dtype=F64)
    cov = A @ A.T + 0.5 * torch.eye(12, dtype=F64)
    mu = torch.randn(12, generator=g, dtype=F64)
    y = torch.randn(12, generator=g, dtype=F64)
    ref = multivariate_normal(mu.numpy(), cov.numpy()).logpdf(y.numpy())
    assert abs(float(exact.mvn_log_prob(y, mu, cov)) - ref) < 1e-10


def test_gibbs_exact_constant_ell_equals_seard():
    g = _g(7)
    x = torch.randn(25, 2, generator=g, dtype=F64)
    y = torch.randn(25, generator=g, dtype=F64)
    xs = torch.randn(6, 2, generator=g, dtype=F64)
    l0 = 0.6
    log_ell = torch.full((2, 25), math.log(l0), dtype=F64)
    # a prior whose conditional mean is constant: mean = log l0 and given values == mean
    pr = exact.LogNormalPrior(torch.full((2,), math.log(l0), dtype=F64),
                              torch.full((2, 2), 1.3, dtype=F64), torch.ones(2, dtype=F64))
    mu, cov, ell2 = exact.gibbs_exact_predict(x, y, log_ell, 0.644, 0.011, pr, xs)
    assert torch.allclose(ell2, torch.full((2, 6), l0, dtype=F64), atol=1e-12)
    m_ref, c_ref = exact.seard_predict(x, y, torch.full((1, 2), l0, dtype=F64), 0.644, 0.011,
                                       torch.zeros(1, dtype=F64), xs, with_noise=False)
    assert torch.allclose(mu, m_ref, atol=1e-9)
    assert torch.allclose(cov, c_ref + 1e-4 * torch.eye(6, dtype=F64), atol=1e-9)
    mll = exact.gibbs_exact_mll(x, y, log_ell, 0.644, 0.011, pr)
    ref = exact.seard_mll(x, y, torch.full((1, 2), l0, dtype=F64), 0.644, 0.011,
                          torch.zeros(1, dtype=F64)) + pr.log_prob(x, log_ell).sum() / 25
    assert abs(float(mll - ref)) < 1e-12


def test_gibbs_exact_mll_gradcheck():
    g = _g(8)
    x = torch.randn(8, 2, generator=g, dtype=F64)
    y = torch.randn(8, generator=g, dtype=F64)
    le = (0.1 * torch.randn(2, 8, generator=g, dtype=F64) + math.log(0.4)).requires_grad_()
    pr = _prior()
    assert torch.autograd.gradcheck(lambda t: exact.gibbs_exact_mll(x, y, t, 0.644, 0.011, pr), (le,),
                                    atol=1e-6)


# ---------------------------------------------------------------- SGPR
def test_sgpr_with_Z_equal_X_is_exact_gp():
    g = _g(9)
    x = torch.rand(18, 2, generator=g, dtype=F64) * 3
    y = torch.randn(18, generator=g, dtype=F64)
    xs = torch.rand(5, 2, generator=g, dtype=F64) * 3
    le = 0.1 * torch.randn(2, 18, generator=g, dtype=F64) + math.log(0.5)
    # a rough prior (short lengthscale) so that the 1e-4 jitter does not smooth the conditional mean
    pr = exact.LogNormalPrior(torch.full((2,), math.log(0.5), dtype=F64),
                              torch.full((2, 2), 0.05, dtype=F64), torch.ones(2, dtype=F64))
    os_, noise = 0.644, 0.05
    # Z == X: conditional lengthscale at X equals the given (up to jitter) and Q == K
    mll_s = sparse.sgpr_mll(x, y, x, le, os_, noise, pr)
    mll_e = exact.gibbs_exact_mll(x, y, le, os_, noise, pr)
    assert abs(float(mll_s - mll_e)) < 5e-3
    mean_s, cov_s = sparse.sgpr_predict(x, y, x, le, os_, noise, pr, xs)
    mean_e, cov_e, _ = exact.gibbs_exact_predict(x, y, le, os_, noise, pr, xs)
    assert torch.allclose(mean_s, mean_e, atol=5e-3)
    assert torch.allclose(torch.diagonal(cov_s), torch.diagonal(cov_e) - 1e-4, atol=5e-3)


# ---------------------------------------------------------------- SVGP / DSVI
def _layer(b, M, D, g, mean='constant', dtype=F64):
    shp = (b,) if b else ()
    p = dict(Z=torch.randn((*shp, M, D), generator=g, dtype=dtype),
             lengthscale=torch.rand((*shp, 1, D), generator=g, dtype=dtype) + 0.5,
             outputscale=(torch.rand(shp, generator=g, dtype=dtype) + 0.5),
             m=0.3 * torch.randn((*shp, M), generator=g, dtype=dtype),
             Lq=torch.tril(0.1 * torch.randn((*shp, M, M), generator=g, dtype=dtype))
             + torch.eye(M, dtype=dtype))
    if mean == 'constant':
        p['mean'] = ('constant', 0.1 * torch.randn((*shp, 1), generator=g, dtype=dtype))
    else:
        p['mean'] = ('linear', torch.randn(D, 1, generator=g, dtype=dtype),
                     torch.randn(1, generator=g, dtype=dtype))
    return p


def test_svgp_at_init_is_the_prior():
    g = _g(10)
    p = _layer(2, 16, 3, g, 'linear')
    p['m'] = torch.zeros_like(p['m'])
    p['Lq'] = torch.eye(16, dtype=F64).expand(2, 16, 16).clone()
    x = torch.randn(9, 3, generator=g, dtype=F64)
    xin = x.unsqueeze(0).expand(2, 9, 3)
    mean, var = svgp.svgp_marginal(xin, p)
    assert torch.allclose(mean, ((x @ p['mean'][1]).squeeze(-1) + p['mean'][2]).expand(2, 9))
    assert torch.allclose(var, (p['outputscale'] + 1e-4).unsqueeze(-1).expand(2, 9))
    assert abs(float(svgp.kl_whitened(p))) < 1e-12


def test_kl_whitened_matches_torch_distributions():
    g = _g(11)
    p = _layer(2, 10, 2, g)
    q = torch.distributions.MultivariateNormal(p['m'], scale_tril=torch.tril(p['Lq']))
    pr = torch.distributions.MultivariateNormal(torch.zeros(2, 10, dtype=F64),
                                                torch.eye(10, dtype=F64).expand(2, 10, 10))
    assert abs(float(svgp.kl_whitened(p) - torch.distributions.kl_divergence(q, pr).sum())) < 1e-10


def test_svgp_with_Z_equal_X_and_optimal_q_is_exact_gp():
    g = _g(12)
    n, D = 14, 2
    X = torch.randn(n, D, generator=g, dtype=F64)
    y = torch.randn(n, generator=g, dtype=F64)
    xs = torch.randn(6, D, generator=g, dtype=F64)
    ls = torch.tensor([[0.8, 1.1]], dtype=F64)
    os_, noise, jit = torch.tensor(0.7, dtype=F64), 0.1, 1e-10
    Kzz = K.rbf_ard(X, X, ls, os_) + jit * torch.eye(n, dtype=F64)
    L = torch.linalg.cholesky(Kzz)
    A = L.T                                              # L^{-1} Kzz
    S = torch.inverse(torch.eye(n, dtype=F64) + A @ A.T / noise)
    m = S @ A @ y / noise
    p = dict(Z=X, lengthscale=ls, outputscale=os_, m=m, Lq=torch.linalg.cholesky(S),
             mean=('constant', torch.zeros(1, dtype=F64)))
    mean, cov = svgp.svgp_marginal(xs, p, jitter=jit, full_cov=True)
    m_ref, c_ref = exact.seard_predict(X, y, ls, os_, noise, torch.zeros(1, dtype=F64), xs,
                                       with_noise=False)
    assert torch.allclose(mean, m_ref, atol=1e-6)
    assert torch.allclose(cov - 1e-4 * torch.eye(6, dtype=F64), c_ref, atol=1e-6)


def _dgp(g, D=3, M=12, B=10, S=4, dtype=F64):
    hidden = _layer(2, M, D, g, 'linear', dtype)
    last = _layer(0, M, 2, g, 'constant', dtype)
    x = torch.randn(B, D, generator=g, dtype=dtype)
    y = torch.randn(B, generator=g, dtype=dtype)
    eps = [torch.randn(S, B, 2, generator=g, dtype=dtype)]
    return hidden, last, x, y, eps


def test_dgp_mirror_mode_is_numerically_the_same_computation():
    g = _g(13)
    hidden, last, x, y, eps = _dgp(g)
    a = svgp.dsvi_elbo(x, y, hidden, last, 1, eps, 4, 0.5, 100)
    b = svgp.dsvi_elbo(x, y, hidden, last, 1, eps, 4, 0.5, 100, mirror=True)
    assert abs(float(a - b)) < 1e-12


def test_dsvi_elbo_decomposition_and_gradcheck():
    g = _g(14)
    hidden, last, x, y, eps = _dgp(g, D=2, M=6, B=5, S=3)
    mean, var = svgp.dgp_forward(x, hidden, last, 1, eps, 3)
    assert mean.shape == (3, 5) and var.shape == (3, 5) and bool((var > 0).all())
    ell = svgp.gauss_ell(y, mean, var, 0.5).sum(-1) / 5
    kl = svgp.kl_whitened(hidden) + svgp.kl_whitened(last)
    assert abs(float(svgp.dsvi_elbo(x, y, hidden, last, 1, eps, 3, 0.5, 50) - (ell - kl / 50).mean())) < 1e-14
    # tied hidden layer applied twice (DeepGP(num_layers=2), D == 2): KL still counted once
    eps2 = eps + [torch.randn(3, 5, 2, generator=g, dtype=F64)]
    e2 = svgp.dsvi_elbo(x, y, hidden, last, 2, eps2, 3, 0.5, 50)
    assert torch.isfinite(e2)

    Z = hidden['Z'].clone().requires_grad_()
    m = last['m'].clone().requires_grad_()

    def f(Z_, m_):
        h = dict(hidden, Z=Z_)
        l_ = dict(last, m=m_)
        return svgp.dsvi_elbo(x, y, h, l_, 1, eps, 3, 0.5, 50)
    assert torch.autograd.gradcheck(f, (Z, m), atol=1e-6)


def test_gauss_ell_is_the_expected_log_density():
    # closed form vs Gauss-Hermite quadrature of E_{f~N(mu,v)} log N(y | f, s2)
    mu, v, y, s2 = 0.3, 0.7, -0.4, 0.25
    t, w = np.polynomial.hermite_e.hermegauss(40)
    f = mu + math.sqrt(v) * t
    quad = float((w * (-0.5 * math.log(2 * math.pi * s2) - 0.5 * (y - f) ** 2 / s2)).sum() / math.sqrt(2 * math.pi))
    got = float(svgp.gauss_ell(torch.tensor(y, dtype=F64), torch.tensor(mu, dtype=F64),
                               torch.tensor(v, dtype=F64), s2))
    assert abs(got - quad) < 1e-12


def test_adam_step_matches_torch_optim():
    g = _g(15)
    p0 = [torch.randn(5, generator=g, dtype=F64), torch.randn(3, 2, generator=g, dtype=F64)]
    tp = [p.clone().requires_grad_() for p in p0]
    opt = torch.optim.Adam(tp, lr=0.01)
    mine, state = [p.clone() for p in p0], {}
    for it in range(3):
        grads = [torch.randn(p.shape, generator=g, dtype=F64) for p in p0]
        for p, gr in zip(tp, grads):
            p.grad = gr.clone()
        opt.step()
        mine = svgp.adam_step(mine, grads, state)
    for a, b in zip(mine, tp):
        assert torch.allclose(a, b.detach(), atol=1e-14)


# ---------------------------------------------------------------- matrix-normal prior / P-S kernel
def test_conditional_H_kron_form_is_row_regression():
    g = _g(16)
    x = torch.rand(9, 2, generator=g) * 2
    xs = torch.rand(4, 2, generator=g) * 2
    H = torch.randn(9, 2, generator=g)
    ls = torch.tensor([[0.6931, 0.6931]])
    col = torch.tensor([[5.0, 0.0], [0.0, 5.0]])
    got = psgibbs.conditional_H(xs, x, H, ls, col)
    R = K.rbf_ard(x, x, ls) + 1e-5 * torch.eye(9)
    ref = K.rbf_ard(xs, x, ls) @ torch.linalg.solve(R.double(), H.double()).float()
    assert got.shape == (4, 2)
    assert torch.allclose(got, ref, atol=2e-2, rtol=2e-2)    # fp32 explicit inverse of an RBF Gram


def test_mv_gibbs_forward_branches():
    g = _g(17)
    x = torch.rand(8, 2, generator=g, dtype=F64)
    xs = torch.rand(5, 2, generator=g, dtype=F64)
    H = torch.randn(8, 2, generator=g, dtype=F64)
    Dm = torch.diag(torch.randn(2, generator=g, dtype=F64))
    ls = torch.tensor([[0.6931, 0.6931]], dtype=F64)
    col = 5.0 * torch.eye(2, dtype=F64)
    Kxx = psgibbs.mv_gibbs_forward(x, x, x, H, Dm, ls, col)
    assert torch.allclose(Kxx, Kxx.T, atol=1e-12)
    # diagonal: |S|^{1/2} |S|^{-1/2} exp(0) = 1
    assert torch.allclose(torch.diagonal(Kxx), torch.ones(8, dtype=F64), atol=1e-12)
    Ksx = psgibbs.mv_gibbs_forward(xs, x, x, H, Dm, ls, col)
    Kxs = psgibbs.mv_gibbs_forward(x, xs, x, H, Dm, ls, col)
    assert Ksx.shape == (5, 8) and torch.allclose(Ksx, Kxs.T, atol=1e-12)
    Kss = psgibbs.mv_gibbs_forward(xs, xs, x, H, Dm, ls, col)
    assert Kss.shape == (5, 5)
    joint = torch.cat([torch.cat([Kxx, Kxs], 1), torch.cat([Ksx, Kss], 1)], 0)
    assert torch.linalg.eigvalsh(joint).min() > -1e-8


def test_matrix_normal_prior_vec_orders_preserved():
    g = _g(18)
    x = torch.rand(6, 2, generator=g)
    row = K.rbf_ard(x, x, torch.tensor([[0.6931, 0.6931]]))
    col = torch.tensor([[5.0, 0.0], [0.0, 5.0]])
    pr = psgibbs.MatrixNormalPrior(torch.zeros(6, 2), row, col)
    assert pr.kron_cov.shape == (12, 12) and pr.kron_cov.dtype == F64
    H = pr.sample_from_eps(torch.randn(12, generator=g))
    assert H.shape == (6, 2)
    ref = torch.distributions.MultivariateNormal(
        pr.vec_loc, covariance_matrix=pr.kron_cov).log_prob(H.T.flatten())
    assert abs(float(pr.log_prob(H) - ref)) < 1e-8


def test_periodic_kernel_known_answers_and_spatiotemporal_restatement():
    """oracle.kernels.periodic (gpytorch's pre-ARD PeriodicKernel as recalled: division by ell, not ell^2) equals
    scikit-learn's ExpSineSquared with length_scale = sqrt(ell); it is periodic in the period; the spatio-temporal
    exact-GP log marginal likelihood of oracle.spatiotemporal equals scipy's multivariate-normal log-density under
    the same composite kernel assembled from scikit-learn kernels; torch.autograd.gradcheck passes.  (Formula pins only: gpytorch itself is absent.)"""
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, ExpSineSquared
    from oracle import kernels, spatiotemporal as st
    g = torch.Generator().manual_seed(3)
    t = torch.randn(40, 1, generator=g, dtype=torch.float64)
    ell, per = torch.tensor(0.7, dtype=torch.float64), torch.tensor(1.3, dtype=torch.float64)
    K = kernels.periodic(t, t, ell, per)
    K_sk = ExpSineSquared(length_scale=float(ell) ** 0.5, periodicity=float(per))(t.numpy())
    assert np.allclose(K.numpy(), K_sk, rtol=1e-12, atol=1e-12)
    assert torch.allclose(torch.diagonal(kernels.periodic(t, t + per, ell, per)), torch.ones(40, dtype=torch.float64), atol=1e-12)
    x = torch.randn(30, 3, generator=g, dtype=torch.float64)
    y = torch.randn(30, generator=g, dtype=torch.float64)
    f64 = lambda v: torch.tensor(v, dtype=torch.float64)
    p = dict(os_t=f64(7.5), ls_t=f64(0.9), ls_p=f64(0.6), period=f64(1.1), os_s=f64(0.8), ls_s=f64([0.7, 0.7]))
    noise = 0.2
    # scikit-learn kernels act on all columns; restrict by giving the unused columns a huge lengthscale / period
    kt = ConstantKernel(7.5) * RBF([0.9, 1e9, 1e9])
    Kt = kt(x.numpy()) * ExpSineSquared(length_scale=0.6 ** 0.5, periodicity=1.1)(x[:, :1].numpy())
    Ks = (ConstantKernel(0.8) * RBF([1e9, 0.7, 0.7]))(x.numpy())
    K_ref = Kt + Ks
    assert np.allclose(st.st_kernel(x, x, p).numpy(), K_ref, rtol=1e-9, atol=1e-10)
    from scipy.stats import multivariate_normal
    lml = multivariate_normal(np.zeros(30), K_ref + noise * np.eye(30)).logpdf(y.numpy())
    assert abs(float(st.st_exact_mll(x, y, p, noise)) * 30 - lml) < 1e-8
    leaves = [v.clone().requires_grad_() for v in (p['ls_t'], p['ls_p'], p['period'])]
    assert torch.autograd.gradcheck(
        lambda a, b, c: st.st_exact_mll(x, y, dict(p, ls_t=a, ls_p=b, period=c), noise), leaves, atol=1e-6)


def test_oracle_reproduces_its_committed_golden_fixtures():
    """tests/golden/oracle_cfg2_gibbs.npz / oracle_dgp.npz were written by tests/golden/make_oracle_goldens.py from
    this oracle; re-deriving them guards the checker against drift between rounds (SURVEY 8c fixtures)."""
    import importlib.util
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('make_oracle_goldens', os.path.join(here, 'make_oracle_goldens.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for name, fresh in (('oracle_cfg2_gibbs.npz', mod.cfg2()), ('oracle_dgp.npz', mod.dgp())):
        gold = np.load(os.path.join(here, name))
        assert set(gold.files) == set(fresh)
        for k in gold.files:
            assert np.allclose(np.asarray(fresh[k], dtype=np.float64), gold[k], rtol=1e-9, atol=1e-11), (name, k)


def test_sparse_spatiotemporal_nonstationary_restatement():
    """oracle.spatiotemporal.st_ns_*: with ALL training points as inducing points (distinct times) both SGPR components
    are exact, so the objective equals the exact-GP log marginal likelihood of  K_t + os_s Gibbs  plus the prior term
    (trace terms vanish); gradcheck passes; predict() reproduces the reference's dense-branch algebra."""
    from oracle import spatiotemporal as st
    g = _g(23)
    n = 14
    x = torch.randn(n, 3, generator=g, dtype=F64)
    y = torch.randn(n, generator=g, dtype=F64)
    f64 = lambda v: torch.tensor(v, dtype=F64)
    p = dict(os_t=f64(7.4), ls_t=f64(0.9), ls_p=f64(0.8), period=f64(1.3), os_s=f64(0.7))
    prior = exact.LogNormalPrior(torch.full((2,), math.log(0.5), dtype=F64), torch.full((2, 2), 1.3, dtype=F64),
                                 torch.full((2,), 0.6931, dtype=F64))
    le = 0.1 * torch.randn(2, n, generator=g, dtype=F64) + math.log(0.5)
    noise = f64(0.3)
    val = st.st_ns_mll(x, y, x.clone(), le, p, noise, prior)
    ell = torch.exp(le)
    # at x == z the conditional mean of log ell reproduces log ell up to the 1e-4 jitter of the conditional
    ell_x = prior.conditional_mean_ell(x[:, 1:3], x[:, 1:3], ell)
    Kfull = st._temporal_kernel(x[:, :1], x[:, :1], p) + p['os_s'] * K.gibbs(x[:, 1:3], x[:, 1:3], ell_x, ell_x)
    dense = (exact.mvn_log_prob(y, torch.zeros_like(y), Kfull + noise * torch.eye(n, dtype=F64))
             + prior.log_prob(x[:, 0:2], le).sum()) / n
    assert abs(float(val) - float(dense)) < 2e-3 * abs(float(dense))         # Q_s uses ell(x) | ell_z: jitter-level gap
    z = x[:6].clone() + 0.1 * torch.randn(6, 3, generator=g, dtype=F64)
    lez = le[:, :6].clone()
    assert torch.autograd.gradcheck(lambda a, b: st.st_ns_mll(x, y, a, b, p, noise, prior),
                                    [z.requires_grad_(), lez.requires_grad_()], atol=1e-6)
    xs = torch.randn(5, 3, generator=g, dtype=F64)
    m, c = st.st_ns_predict(x, y, z.detach(), lez.detach(), p, noise, prior, xs)
    assert m.shape == (5,) and c.shape == (5, 5) and bool(torch.isfinite(m).all()) and bool(torch.isfinite(c).all())


def test_generic_inducing_point_kernel_restatement_reduces_to_the_exact_gp_at_z_equal_x():
    """oracle.sparse.ipk_mll / ipk_predict (gpytorch InducingPointKernel over any base kernel): with Z = X the low-rank
    covariance is exact, so the objective is the exact-GP log marginal likelihood / N (trace term 0) and the
    prediction is the exact-GP posterior."""
    g = _g(31)
    n, ns = 25, 7
    x = torch.randn(n, 2, generator=g, dtype=F64)
    xs = torch.randn(ns, 2, generator=g, dtype=F64)
    y = torch.randn(n, generator=g, dtype=F64)
    ls = torch.tensor([[0.8, 1.1]], dtype=F64)
    Kxx = K.rbf_ard(x, x, ls, 0.9) + 1e-9 * torch.eye(n, dtype=F64)
    Ksx, Kss = K.rbf_ard(xs, x, ls, 0.9), K.rbf_ard(xs, xs, ls, 0.9)
    noise = 0.2
    val = sparse.ipk_mll(Kxx, Kxx, torch.diagonal(Kxx), y, noise)
    ref = exact.mvn_log_prob(y, torch.zeros_like(y), Kxx + noise * torch.eye(n, dtype=F64)) / n
    assert abs(float(val) - float(ref)) < 1e-6 * abs(float(ref))
    m, c = sparse.ipk_predict(Kxx, Kxx, torch.diagonal(Kxx), Ksx, torch.diagonal(Kss), y, noise, with_noise=False)
    Kn = Kxx + noise * torch.eye(n, dtype=F64)
    m_ref = Ksx @ torch.linalg.solve(Kn, y)
    v_ref = torch.diagonal(Kss - Ksx @ torch.linalg.solve(Kn, Ksx.T))
    assert torch.allclose(m, m_ref, rtol=1e-6, atol=1e-8)
    assert torch.allclose(torch.diagonal(c), v_ref, rtol=1e-5, atol=1e-7)
