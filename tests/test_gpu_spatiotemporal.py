"""SURVEY 8f.1 -- spatio-temporal additive models (models/spatio_temporal_models.py of the reference):
the fused RBF x Periodic build kernel and the SpatioTemporal_* models against the CPU oracle
(oracle/kernels.periodic, oracle/spatiotemporal).  gpytorch's PeriodicKernel is restated from memory
(pre-ARD form, division by the lengthscale): parity unpinned, like every gpytorch-backed piece."""
import math
import os

import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


@pytest.mark.parametrize('dt', [F64, F32])
@pytest.mark.parametrize('b,n1,n2,D,with_rbf,with_os,shared', [
    (1, 215, 215, 1, True, True, True), (2, 65, 130, 1, True, True, False), (1, 1, 7, 1, False, False, True),
    (3, 100, 257, 2, True, False, True), (1, 300, 64, 2, False, True, True)])
def test_rbf_periodic_build_forward_backward(dt, b, n1, n2, D, with_rbf, with_os, shared):
    _need_gpu()
    from nsgp import ops
    from oracle import kernels
    g = torch.Generator().manual_seed(31 + n1 + 7 * D)
    x1 = torch.randn((n1, D) if shared else (b, n1, D), generator=g, dtype=F64)
    x2 = torch.randn((n2, D) if shared else (b, n2, D), generator=g, dtype=F64)
    lr = torch.rand(b, D, generator=g, dtype=F64) + 0.6
    lp = torch.rand(b, generator=g, dtype=F64) + 0.5
    pe = torch.rand(b, generator=g, dtype=F64) + 0.8
    os_ = torch.rand(b, generator=g, dtype=F64) + 0.5
    G = torch.randn(b, n1, n2, generator=g, dtype=F64)
    leaves = [t.clone().requires_grad_() for t in (x1, x2, lr, lp, pe, os_)]
    xo1, xo2, lro, lpo, peo, oso = leaves
    a1 = xo1 if xo1.dim() == 3 else xo1.unsqueeze(0).expand(b, n1, D)
    a2 = xo2 if xo2.dim() == 3 else xo2.unsqueeze(0).expand(b, n2, D)
    K_ref = kernels.periodic(a1, a2, lpo.reshape(b, 1, 1), peo.reshape(b, 1, 1))
    if with_rbf:
        K_ref = K_ref * kernels.rbf_ard(a1, a2, lro.reshape(b, 1, D))
    if with_os:
        K_ref = K_ref * oso.reshape(b, 1, 1)
    (K_ref * G).sum().backward()
    dev = [t.detach().to(dt).cuda().requires_grad_() for t in (x1, x2, lr, lp, pe, os_)]
    d1, d2, dlr, dlp, dpe, dos = dev
    K = ops.rbf_periodic_kernel(d1, d2, dlr if with_rbf else None, dlp, dpe, dos if with_os else None)
    (K * G.to(dt).cuda()).sum().backward()
    tol = dict(rtol=1e-10, atol=1e-11) if dt == F64 else dict(rtol=2e-4, atol=2e-5)
    assert torch.allclose(K.detach().cpu().double(), K_ref.detach(), **tol)
    gtol = dict(rtol=1e-8, atol=1e-8) if dt == F64 else dict(rtol=5e-3, atol=5e-3 * max(1.0, math.sqrt(n1 * n2) / 10))
    pairs = [(d1, xo1), (d2, xo2), (dlp, lpo), (dpe, peo)]
    if with_rbf:
        pairs.append((dlr, lro))
    if with_os:
        pairs.append((dos, oso))
    for got, want in pairs:
        assert torch.allclose(got.grad.cpu().double(), want.grad, **gtol), (got.shape, (got.grad.cpu().double() - want.grad).abs().max())
    # periodic in the period along one axis (no RBF factor): k(x, x + period e_0) == os
    if not with_rbf and D == 1:
        Kp = ops.rbf_periodic_kernel(d1.detach(), (d1 + dpe.reshape(-1)[0]).detach(), None, dlp.detach(), dpe.detach(), None)
        assert torch.allclose(torch.diagonal(Kp[0]), torch.ones(n1, dtype=dt, device='cuda'), atol=1e-5 if dt == F32 else 1e-12)


def _uib_subset():
    """The 215-point subset of experiments/spatio_temporal_exp.py:36-56: year 2000, months 1-5; months 1-4 train."""
    d = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'data', 'uib_spatio_temporal.csv'))
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


def _model_params(model):
    sp = torch.nn.functional.softplus
    tk, sk = model.temporal_covar_module, model.spatial_covar_module
    raw = dict(os_t=tk.raw_outputscale, ls_t=tk.base_kernel.kernels[0].raw_lengthscale,
               ls_p=tk.base_kernel.kernels[1].raw_lengthscale, period=tk.base_kernel.kernels[1].raw_period_length,
               os_s=sk.raw_outputscale, ls_s=sk.base_kernel.raw_lengthscale, noise=model.likelihood.noise_covar.raw_noise)
    leaves = {k: v.detach().cpu().double().clone().requires_grad_() for k, v in raw.items()}
    p = dict(os_t=sp(leaves['os_t']) + 7.0, ls_t=sp(leaves['ls_t']).reshape(()), ls_p=sp(leaves['ls_p']).reshape(()),
             period=sp(leaves['period']).reshape(()), os_s=sp(leaves['os_s']),
             ls_s=sp(leaves['ls_s']).reshape(1).expand(2))       # RBFKernel(active_dims=(1,2)): ONE shared lengthscale
    noise = sp(leaves['noise']).reshape(()) + 1e-4
    return raw, leaves, p, noise


@pytest.mark.parametrize('dt', [F64, F32])
def test_stationary_spatiotemporal_exact_gp_matches_oracle(dt):
    _need_gpu()
    import nsgp.gp as gpytorch
    from models.spatio_temporal_models import SpatioTemporal_Stationary
    from oracle import spatiotemporal as st
    xtr, ytr, xte, yte = _uib_subset()
    assert xtr.shape == (172, 3) and xte.shape == (43, 3)
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SpatioTemporal_Stationary(xtr.to(dt), ytr.to(dt), lik).to(dt).cuda()
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.add_(0.3 * torch.randn(p_.shape, generator=g, dtype=F64).to(p_))
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    xd, yd = model.train_inputs[0], model.train_targets
    val = mll(model(xd), yd)
    val.backward()
    raw, leaves, p, noise = _model_params(model)
    ref = st.st_exact_mll(xtr, ytr, p, noise)
    ref.backward()
    rel = 1e-9 if dt == F64 else 2e-3
    assert abs(float(val) - float(ref)) < rel * abs(float(ref)) + (1e-10 if dt == F64 else 1e-4)
    for k, v in raw.items():
        got, want = v.grad.detach().cpu().double().reshape(-1), leaves[k].grad.reshape(-1)
        assert torch.allclose(got, want, rtol=1e-6 if dt == F64 else 5e-2, atol=1e-8 if dt == F64 else 2e-3), (k, got, want)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(xte.to(dt).cuda()))
        m_ref, c_ref = st.st_exact_predict(xtr, ytr, {k: v.detach() for k, v in p.items()}, noise.detach(), xte)
    tol = dict(rtol=1e-7, atol=1e-8) if dt == F64 else dict(rtol=2e-3, atol=2e-3)
    assert torch.allclose(pred.loc.cpu().double(), m_ref, **tol)
    assert torch.allclose(torch.diagonal(pred.covariance_matrix).cpu().double(), torch.diagonal(c_ref), **tol)


def test_stationary_spatiotemporal_sgpr_matches_oracle():
    _need_gpu()
    import nsgp.gp as gpytorch
    from models.spatio_temporal_models import SpatioTemporal_Stationary
    from oracle import spatiotemporal as st
    xtr, ytr, xte, yte = _uib_subset()
    z = xtr[::6].clone()
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SpatioTemporal_Stationary(xtr, ytr, lik, z).double().cuda()
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    raw, leaves, p, noise = _model_params(model)
    zl = model.covar_module.inducing_points.detach().cpu().double().clone().requires_grad_()
    ref = st.st_sgpr_mll(xtr, ytr, zl, p, noise)
    ref.backward()
    assert abs(float(val) - float(ref)) < 1e-7 * abs(float(ref)) + 1e-9
    assert torch.allclose(model.covar_module.inducing_points.grad.cpu().double(), zl.grad, rtol=1e-5, atol=1e-7)
    for k, v in raw.items():
        assert torch.allclose(v.grad.detach().cpu().double().reshape(-1), leaves[k].grad.reshape(-1), rtol=1e-5, atol=1e-7), k


def test_sparse_nonstationary_spatiotemporal_trains_and_predicts():
    _need_gpu()
    import nsgp.gp as gpytorch
    from models.gibbs_kernels import LogNormalPriorProcess
    from models.spatio_temporal_models import SparseSpatioTemporal_Nonstationary
    xtr, ytr, xte, yte = _uib_subset()
    z = xtr[::5].clone()
    prior = LogNormalPriorProcess(input_dim=2, active_dims=(0, 1)).double()
    prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
    prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
    for p_ in prior.parameters():
        p_.requires_grad = False
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SparseSpatioTemporal_Nonstationary(xtr, ytr, lik, prior, z, num_dim=2).double().cuda()
    model.train(); lik.train()
    opt = torch.optim.Adam([p_ for p_ in model.parameters() if p_.requires_grad], lr=0.015)
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    losses = []
    for _ in range(25):
        opt.zero_grad()
        loss = -mll(model(model.train_inputs[0]), model.train_targets)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0]
    assert model.temporal_covar_module.inducing_points.grad is None         # frozen (spatio_temporal_models.py:43)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model.predict(xte.double().cuda()))
    assert pred.loc.shape == (43,) and bool(torch.isfinite(pred.loc).all())
    # The reference's predict() takes its dense branch here (the summed covariance is not a LowRankRootLazyTensor,
    # spatio_temporal_models.py:101-113): it uses rows of the dense joint covariance where the SGPR algebra expects
    # a low-rank root, so the returned covariance is not a valid predictive covariance (its own docstring warns);
    # the restatement reproduces the arithmetic, the test checks shape and finiteness only.
    assert pred.covariance_matrix.shape == (43, 43) and bool(torch.isfinite(pred.covariance_matrix).all())
    # cross-check the arithmetic against the same formulas evaluated densely in float64 torch
    with torch.no_grad():
        xall = torch.cat([model.train_inputs[0], xte.double().cuda()], dim=-2)
        C = gpytorch.lazy.delazify(model.forward(xall).lazy_covariance_matrix).cpu()
        sig = float(lik.noise) ** 0.5
        n = xtr.shape[0]
        L, At = C[n:, :], C[:n, :] / sig
        B = torch.eye(C.shape[-1], dtype=F64) + At.T @ At
        mean_ref = L @ torch.linalg.solve(B, At.T @ ytr) / sig
        cov_ref = C[n:, n:] - L @ ((torch.eye(C.shape[-1], dtype=F64) - torch.inverse(B)) @ L.T)
        f = model.predict(xte.double().cuda())
    assert torch.allclose(f.loc.cpu(), mean_ref, rtol=1e-6, atol=1e-6)
    assert torch.allclose(f.covariance_matrix.cpu(), cov_ref, rtol=1e-5, atol=1e-4 * float(cov_ref.abs().max()))


def test_khyber_temporal_rbf_times_periodic_exact_gp_matches_oracle():
    """BASELINE configs[0] with the reference's own model for it: experiments/temporal_exp.py:34-44
    KhyberTemporalStat = ExactGP(ConstantMean, ScaleKernel(RBFKernel() * PeriodicKernel(), outputscale > 7)) on
    khyber_time_series.csv (first 80 % train, no shuffle, :59-67; Box-Cox targets).  Objective, gradients and
    eval-mode predictions against the CPU oracle (float64)."""
    _need_gpu()
    import scipy.stats
    import nsgp.gp as gpytorch
    from nsgp.gp.constraints import GreaterThan
    from nsgp.gp.kernels import PeriodicKernel, RBFKernel, ScaleKernel
    from oracle import kernels
    from oracle.exact import mvn_log_prob
    d = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'data', 'khyber_time_series.csv'))
    x = torch.tensor(np.array(d)[:, 0], dtype=F64)
    y = torch.tensor(scipy.stats.boxcox(np.array(d)[:, -1])[0], dtype=F64)
    stdx, meanx = torch.std_mean(x)
    xn = (x - meanx) / stdx
    k = math.ceil(0.8 * y.shape[0])
    xtr, ytr, xte = xn[:k], y[:k], xn[k:]

    class KhyberTemporalStat(gpytorch.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = gpytorch.means.ConstantMean()
            self.covar_module = ScaleKernel(RBFKernel() * PeriodicKernel(), outputscale_constraint=GreaterThan(7))

        def forward(self, x):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = KhyberTemporalStat(xtr, ytr, lik).double().cuda()
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p_ in model.parameters():
            p_.add_(0.25 * torch.randn(p_.shape, generator=g, dtype=F64).to(p_))
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    sp = torch.nn.functional.softplus
    raw = dict(os=model.covar_module.raw_outputscale, lr=model.covar_module.base_kernel.kernels[0].raw_lengthscale,
               lp=model.covar_module.base_kernel.kernels[1].raw_lengthscale,
               pe=model.covar_module.base_kernel.kernels[1].raw_period_length, c=model.mean_module.constant,
               noise=lik.noise_covar.raw_noise)
    lv = {k_: v.detach().cpu().double().clone().requires_grad_() for k_, v in raw.items()}

    def kern(a, b):
        a, b = a.unsqueeze(-1), b.unsqueeze(-1)
        return (sp(lv['os']) + 7.0) * kernels.rbf_ard(a, b, sp(lv['lr']).reshape(1, 1)) * \
            kernels.periodic(a, b, sp(lv['lp']).reshape(()), sp(lv['pe']).reshape(()))
    noise = sp(lv['noise']).reshape(()) + 1e-4
    n = xtr.shape[0]
    mean = lv['c'].reshape(()).expand(n)
    ref = mvn_log_prob(ytr, mean, kern(xtr, xtr) + noise * torch.eye(n, dtype=F64)) / n
    ref.backward()
    assert abs(float(val) - float(ref)) < 1e-9 * abs(float(ref)) + 1e-10
    for k_, v in raw.items():
        assert torch.allclose(v.grad.detach().cpu().double().reshape(-1), lv[k_].grad.reshape(-1), rtol=1e-6, atol=1e-8), k_
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(xte.cuda()))
        Kxx = kern(xtr, xtr) + noise * torch.eye(n, dtype=F64)
        Ksx = kern(xte, xtr)
        m_ref = lv['c'].reshape(()) + Ksx @ torch.linalg.solve(Kxx, ytr - mean)
        v_ref = torch.diagonal(kern(xte, xte) - Ksx @ torch.linalg.solve(Kxx, Ksx.T)) + noise
    assert torch.allclose(pred.loc.cpu(), m_ref, rtol=1e-7, atol=1e-8)
    assert torch.allclose(torch.diagonal(pred.covariance_matrix).cpu(), v_ref, rtol=1e-6, atol=1e-8)
