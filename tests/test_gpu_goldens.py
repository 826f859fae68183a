"""The HIP path against the COMMITTED golden fixtures (tests/golden/oracle_*.npz, written by the CPU oracle with
tests/golden/make_oracle_goldens.py): BASELINE configs[1] -- Gibbs exact GP on uib_spatial.csv, 316 train /
78 test, float64 -- and the 2-layer DSVI DeepGP on a 315-point batch, float32 model vs float64 fixtures."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def test_gibbs_exact_gp_meets_the_cfg2_fixture():
    _need_gpu()
    import importlib.util
    import nsgp.gp as gpytorch
    from models.gibbs_kernels import LogNormalPriorProcess
    from models.nonstationary_models import DiagonalExactGP
    spec = importlib.util.spec_from_file_location('mk', os.path.join(GOLD, 'make_oracle_goldens.py'))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    xtr, ytr, xte, yte = mk.uib_spatial()
    z = np.load(os.path.join(GOLD, 'oracle_cfg2_gibbs.npz'))
    prior = LogNormalPriorProcess(input_dim=2).double().cuda()
    prior.covar_module.outputscale = torch.ones_like(prior.covar_module.outputscale)
    prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
    prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
    for p in prior.parameters():
        p.requires_grad = False
    lik = gpytorch.likelihoods.GaussianLikelihood().double()
    model = DiagonalExactGP(xtr, ytr, lik, prior, num_dim=2).double().cuda()
    model.likelihood.noise = float(z['noise'])
    model.covar_module.outputscale = float(z['outputscale'])
    with torch.no_grad():
        model.log_ell_train_x.copy_(torch.tensor(z['log_ell']))
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    # noise / outputscale go through the softplus constraint and back (setter -> raw -> value): ~1e-9 relative
    assert abs(float(val.detach()) - float(z['mll'])) < 2e-8 * abs(float(z['mll']))
    assert np.allclose(model.log_ell_train_x.grad.cpu().numpy(), z['grad_log_ell'], rtol=1e-6, atol=1e-9)
    with torch.no_grad():
        K = gpytorch.lazy.delazify(model.covar_module(model.train_inputs[0], ell1=torch.exp(model.log_ell_train_x))).cpu()
    ii = torch.tensor([0, 1, 17, 100, 200, 315])
    assert np.allclose(K[ii][:, ii].numpy(), z['K_sample'], rtol=1e-8, atol=1e-10)
    assert abs(float(K.sum()) - float(z['K_sum'])) < 1e-8 * abs(float(z['K_sum']))
    assert abs(float(torch.trace(K)) - float(z['K_trace'])) < 1e-8 * float(z['K_trace'])
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = model.predict(xte.cuda())
    # posterior mean within the north star's 1e-4 relative bound (float64 path: ~1e-8 in practice)
    assert np.allclose(pred.loc.cpu().numpy(), z['pred_mean'], rtol=1e-7, atol=1e-8)
    assert np.allclose(torch.diagonal(pred.covariance_matrix).cpu().numpy(), z['pred_var'], rtol=1e-6, atol=1e-8)


def test_dsvi_deepgp_meets_the_dgp_fixture():
    _need_gpu()
    import models.dgps as m
    from nsgp.gp import settings
    from nsgp.gp.constraints import inv_softplus
    from nsgp.gp.mlls import DeepApproximateMLL, VariationalELBO
    z = np.load(os.path.join(GOLD, 'oracle_dgp.npz'))
    T = lambda k: torch.tensor(z[k])
    B, D = z['x'].shape
    M = z['h_Z'].shape[-2]
    model = m.DeepGP(1, (int(z['num_data']), D), num_inducing=M).cuda()
    h, l_ = model.layers[0], model.last_layer
    with torch.no_grad():
        for mod, tag in ((h, 'h'), (l_, 'l')):
            vs = mod.variational_strategy
            vs.inducing_points.copy_(T(f'{tag}_Z'))
            vs._variational_distribution.variational_mean.copy_(T(f'{tag}_m'))
            vs._variational_distribution.chol_variational_covar.copy_(T(f'{tag}_Lq'))
            vs.variational_params_initialized.fill_(1)
            mod.covar_module.base_kernel.raw_lengthscale.copy_(inv_softplus(T(f'{tag}_lengthscale')))
            mod.covar_module.raw_outputscale.copy_(inv_softplus(T(f'{tag}_outputscale')))
        h.mean_module.weights.copy_(T('h_w'))
        h.mean_module.bias.copy_(T('h_b'))
        l_.mean_module.constant.zero_()
        model.likelihood.noise = float(z['noise'])
    eps = T('eps')
    S = eps.shape[0]
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, int(z['num_data'])))
    model.train()
    with settings.num_likelihood_samples(S), settings.eps_provider(lambda shape, dt, dev: eps.to(dev, dt)):
        out = model(T('x').float().cuda())
        elbo = mll(out, T('y').float().cuda())
    elbo.backward()
    assert abs(float(elbo) - float(z['elbo'])) < 2e-4 * abs(float(z['elbo'])) + 1e-5
    sp = torch.nn.functional.softplus

    def chain(raw, gold_grad):            # d/d raw = d/d value * sigmoid(raw)
        return gold_grad * torch.sigmoid(raw.detach().cpu().double())
    checks = []
    for mod, tag in ((h, 'h'), (l_, 'l')):
        vs = mod.variational_strategy
        checks += [(vs.inducing_points.grad, T(f'{tag}_grad_Z')),
                   (vs._variational_distribution.variational_mean.grad, T(f'{tag}_grad_m')),
                   (vs._variational_distribution.chol_variational_covar.grad, T(f'{tag}_grad_Lq')),
                   (mod.covar_module.base_kernel.raw_lengthscale.grad,
                    chain(mod.covar_module.base_kernel.raw_lengthscale, T(f'{tag}_grad_lengthscale'))),
                   (mod.covar_module.raw_outputscale.grad, chain(mod.covar_module.raw_outputscale, T(f'{tag}_grad_outputscale')))]
    for got, want in checks:
        got = got.detach().cpu().double()
        scale = float(want.abs().max()) + 1e-12
        assert float((got - want.reshape(got.shape)).abs().max()) / scale < 2e-2      # float32 model vs float64 fixture
    with torch.no_grad(), settings.num_likelihood_samples(S):
        mean1, var1 = h.variational_strategy.marginals(T('x').float().cuda())
    # posterior mean of layer 1 within 1e-4 relative (max-norm), the north star's bound
    ref_m, ref_v = T('h_mean'), T('h_var')
    assert float((mean1.cpu().double() - ref_m).abs().max()) < 1e-4 * float(ref_m.abs().max())
    assert float((var1.cpu().double() - ref_v).abs().max()) < 2e-4 * float(ref_v.abs().max())
