"""Host-side logic of the drop-in surface that needs no GPU: the gpytorch-shaped namespace --
parameter names / shapes / initial values the reference's scripts and checkpoints rely on, constraint transforms and
setters, settings contexts, the `gpytorch` alias, the scoped transform cache -- and the planner queries of the C ABI.
No kernel is launched (constructing the models only allocates CPU tensors)."""
import math

import pytest
import torch


def test_gpytorch_names_used_by_the_reference_scripts_resolve():
    import models  # noqa: F401  (installs the alias when the real gpytorch is absent)
    import gpytorch
    for path in ('models.ExactGP', 'models.deep_gps.DeepGPLayer', 'models.deep_gps.DeepGP', 'means.ZeroMean',
                 'means.ConstantMean', 'means.LinearMean', 'kernels.Kernel', 'kernels.ScaleKernel', 'kernels.RBFKernel',
                 'kernels.InducingPointKernel', 'kernels.PeriodicKernel', 'kernels.MaternKernel',
                 'likelihoods.GaussianLikelihood', 'likelihoods.Likelihood', 'mlls.ExactMarginalLogLikelihood',
                 'mlls.VariationalELBO', 'mlls.DeepApproximateMLL', 'mlls.AddedLossTerm',
                 'mlls.InducingPointKernelAddedLossTerm', 'variational.VariationalStrategy',
                 'variational.CholeskyVariationalDistribution', 'distributions.MultivariateNormal',
                 'distributions.MultitaskMultivariateNormal', 'constraints.GreaterThan', 'priors.MultivariateNormalPrior',
                 'settings.num_likelihood_samples', 'settings.max_cg_iterations', 'settings.cholesky_jitter',
                 'settings.sgpr_diagonal_correction', 'lazy.delazify', 'lazy.LowRankRootLazyTensor',
                 'lazy.LowRankRootAddedDiagLazyTensor', 'lazy.DiagLazyTensor', 'lazy.MatmulLazyTensor',
                 'utils.cholesky.psd_safe_cholesky', 'utils.broadcasting._mul_broadcast_shape', '__version__'):
        obj = gpytorch
        for part in path.split('.'):
            obj = getattr(obj, part)
        assert obj is not None, path


def test_deepgp_state_dict_keys_shapes_and_initial_values():
    import models.dgps as m
    torch.manual_seed(0)
    model = m.DeepGP(1, (1000, 3))                         # reference defaults: num_inducing 250, tied hidden layer
    sd = model.state_dict()
    want = {
        'layers.0.variational_strategy.inducing_points': (2, 250, 3),
        'layers.0.variational_strategy._variational_distribution.variational_mean': (2, 250),
        'layers.0.variational_strategy._variational_distribution.chol_variational_covar': (2, 250, 250),
        'layers.0.mean_module.weights': (3, 1), 'layers.0.mean_module.bias': (1,),
        'layers.0.covar_module.raw_outputscale': (2,), 'layers.0.covar_module.base_kernel.raw_lengthscale': (2, 1, 3),
        'last_layer.variational_strategy.inducing_points': (250, 2),
        'last_layer.variational_strategy._variational_distribution.variational_mean': (250,),
        'last_layer.variational_strategy._variational_distribution.chol_variational_covar': (250, 250),
        'last_layer.mean_module.constant': (1,), 'last_layer.covar_module.raw_outputscale': (),
        'last_layer.covar_module.base_kernel.raw_lengthscale': (1, 2), 'likelihood.noise_covar.raw_noise': (1,),
    }
    for k, shp in want.items():
        assert k in sd and tuple(sd[k].shape) == shp, (k, tuple(sd[k].shape) if k in sd else None)
    sp0 = math.log(2.0)                                    # softplus(0)
    assert abs(float(model.likelihood.noise) - (sp0 + 1e-4)) < 1e-6           # GreaterThan(1e-4), raw init 0
    assert abs(float(model.last_layer.covar_module.outputscale) - sp0) < 1e-6
    assert torch.allclose(model.last_layer.covar_module.base_kernel.lengthscale, torch.full((1, 2), sp0))
    assert torch.equal(torch.tril(sd['last_layer.variational_strategy._variational_distribution.chol_variational_covar']),
                       torch.eye(250))
    # num_layers repeats ONE hidden layer (reference quirk F5): 3 "layers" are the same module
    deep = m.DeepGP(3, (1000, 2))
    assert len({id(l) for l in deep.layers}) == 1


def test_constraints_setters_and_fixed_hyperparameters():
    import nsgp.gp as gp
    from nsgp.gp.constraints import GreaterThan, Positive, inv_softplus
    lik = gp.likelihoods.GaussianLikelihood()
    lik.noise = 0.011                                       # spatial_exp.py:177 style
    assert abs(float(lik.noise) - 0.011) < 1e-7
    k = gp.kernels.ScaleKernel(gp.kernels.RBFKernel(ard_num_dims=2), outputscale_constraint=GreaterThan(7))
    assert abs(float(k.outputscale) - (7 + math.log(2.0))) < 1e-6
    k.outputscale = 7.5
    assert abs(float(k.outputscale) - 7.5) < 1e-6
    k.base_kernel.lengthscale = torch.tensor([[0.3, 1.3]])
    assert torch.allclose(k.base_kernel.lengthscale, torch.tensor([[0.3, 1.3]]), atol=1e-6)
    c = Positive()
    x = torch.tensor([1e-3, 0.5, 30.0])
    assert torch.allclose(c.transform(c.inverse_transform(x)), x, rtol=1e-5)
    assert torch.allclose(inv_softplus(torch.nn.functional.softplus(torch.tensor([-3.0, 0.0, 4.0]))),
                          torch.tensor([-3.0, 0.0, 4.0]), atol=1e-5)
    k.initialize(**{'base_kernel.lengthscale': 0.9})
    assert torch.allclose(k.base_kernel.lengthscale, torch.full((1, 2), 0.9), atol=1e-6)
    with pytest.raises(AttributeError):
        k.initialize(no_such_parameter=1.0)


def test_settings_contexts_and_transform_cache_scope():
    from nsgp.gp import settings
    from nsgp.gp.module import transform_cache
    import nsgp.gp as gp
    assert settings.num_likelihood_samples.value() == 10
    # defaults the precision policy of nsgp/svgp.py reads (a missing default silently switched the float64 hidden path off)
    assert settings.hidden_var_f64.value() == 'auto' and settings.hidden_kzx_f64.on() and settings.whiten_matmul_i8.on()
    assert settings.forward_precision.value() == 'f32'
    with settings.hidden_var_f64(False):
        assert settings.hidden_var_f64.value() is False
    assert settings.hidden_var_f64.value() == 'auto'
    with settings.num_likelihood_samples(3):
        assert settings.num_likelihood_samples.value() == 3
        with settings.num_likelihood_samples(5):
            assert settings.num_likelihood_samples.value() == 5
        assert settings.num_likelihood_samples.value() == 3
    assert settings.num_likelihood_samples.value() == 10
    assert settings.variational_cholesky_jitter.value(torch.float32) == 1e-4
    assert settings.cholesky_jitter.value(torch.float64) == 1e-8 and settings.cholesky_jitter.value(torch.float32) == 1e-6
    k = gp.kernels.RBFKernel()
    a, b = k.lengthscale, k.lengthscale
    assert a is not b                                       # no caching outside a scope (no stale autograd graphs)
    with transform_cache():
        c, d = k.lengthscale, k.lengthscale
        assert c is d
        with torch.no_grad():
            k.raw_lengthscale.add_(1.0)                     # an optimiser step bumps the version: new value
        assert k.lengthscale is not c
    assert k.lengthscale is not d


def test_model_errors_match_the_reference():
    import nsgp.gp as gp
    from models.multivariate_gibbs_kernel import MultivariateGibbsKernel
    with pytest.raises(ValueError, match='Use gibbs 1d kernel for dim 1'):
        MultivariateGibbsKernel(torch.randn(5, 1), 1)
    lik = gp.likelihoods.GaussianLikelihood()
    from models.dgps import ExactGPModel
    model = ExactGPModel(torch.randn(6, 2), torch.randn(6), lik, gp.kernels.ScaleKernel(gp.kernels.RBFKernel()))
    model.train_inputs = None
    model.train()
    with pytest.raises(RuntimeError, match='train_inputs, train_targets cannot be None in training mode'):
        model(torch.randn(6, 2))


def test_planner_queries_of_the_c_abi():
    import nsgp
    lib = nsgp.load_library()
    # row tiles of the fused projection GEMM: 128-row tiles at the headline sizes, 64-row tiles for small problems
    assert lib.nsgp_svgp_colstats_tiles(1024, 40960, 1, 4) == 8
    assert lib.nsgp_svgp_colstats_tiles(1024, 4096, 2, 4) == 8
    assert lib.nsgp_svgp_colstats_tiles(130, 400, 1, 4) == 3
    assert lib.nsgp_svgp_colstats_tiles(1024, 4096, 2, 8) == 16            # float64: 64-row tiles
    assert lib.nsgp_svgp_colstats_tiles(0, 10, 1, 4) == 0
    # split-K workspace only for long inner dimensions with few output tiles
    assert lib.nsgp_svgp_lqbar_workspace(1, 1024, 40960, 4) > 0
    assert lib.nsgp_svgp_lqbar_workspace(1, 64, 64, 4) == 0
    assert lib.nsgp_trtri_workspace(64, 3, 8) == 0 and lib.nsgp_trtri_workspace(1024, 3, 8) == 3 * 1024 * 1024 * 8
    assert lib.nsgp_rbf_periodic_build_bwd_workspace(1, 215, 215, 1, 8) > 0


@pytest.mark.parametrize('name', ['uib_spatial', 'khyber_time_series'])
def test_product_dataprep_matches_reference_goldens(golden_dir, data_dir, name):
    """the PRODUCT's utils.dataprep (not the oracle's copy) against tests/golden/ref_dataprep.npz, which
    was written by running the reference's own utils/dataprep.py:9-52 (tests/golden/make_reference_goldens.py)."""
    import os
    import numpy as np
    import utils.dataprep as dp
    z = np.load(os.path.join(golden_dir, 'ref_dataprep.npz'))
    data = dp.download_data(os.path.join(data_dir, name + '.csv'))
    x, y, mx, sx, my, sy = dp.whitening_transform(data)
    for got, key in ((x, 'x'), (y, 'y'), (mx, 'meanx'), (sx, 'stdx'), (my, 'meany'), (sy, 'stdy')):
        assert torch.equal(got, torch.from_numpy(z[f'{name}_{key}'])), key
    trx, try_, tex, tey = dp.train_test_split(x, y, 0.8)
    assert len(trx) == int(z[f'{name}_ntrain'])
    assert torch.equal(trx[-3:], torch.from_numpy(z[f'{name}_train_x_tail']))
    assert torch.equal(tey[:3], torch.from_numpy(z[f'{name}_test_y_head']))


def test_product_functional_elementwise_helpers_match_reference_goldens(golden_dir):
    """dot / t / op / non-inverting mv of the product's utils.functional on CPU tensors (device-agnostic torch) against
    the reference-generated ref_functional.npz; mv(invert=True) is GPU-only (tests/test_gpu_parity_r2.py)."""
    import os
    import numpy as np
    import utils.functional as fn
    z = np.load(os.path.join(golden_dir, 'ref_functional.npz'))
    T = {k: torch.from_numpy(z[k]) for k in z.files}
    assert torch.equal(fn.dot(T['v1'], T['v2']), T['dot'])
    assert torch.equal(fn.t(T['A']), T['t'])
    assert torch.allclose(fn.mv(T['A'], T['b']), T['mv'], rtol=1e-14, atol=1e-14)
    assert torch.equal(fn.op(T['e1'], T['e2']), T['op']) and torch.equal(fn.op(T['e1']), T['op_self'])
