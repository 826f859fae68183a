"""BASELINE configs[2] end to end: GP inference with `SparseMultivariateGibbsKernel` (M = 512 inducing locations) on
data/uib_spatio_temporal.csv (kernel on (lon, lat)).  The reference kernel is defined for x1 == x2 and
for pairs where one side is the M inducing locations (models/sparse_multivariate_gibbs_kernel.py:84-154; any other pair
leaves its locals unbound), which is what an inducing-point GP needs: the model is an ExactGP whose covar_module is
gpytorch's InducingPointKernel over ScaleKernel(SparseMultivariateGibbsKernel), inducing points = the kernel's own
inducing locations (SGPR / Titsias objective).

* all 5,676 rows, M = 512, float32: objective + gradients finite, a few Adam steps decrease it, predictions finite --
  the oracle's (N, N, 2, 2) temporaries do not fit at this size;
* a 900-row training subset + 150 test rows against the CPU oracle (oracle.psgibbs + oracle.sparse.ipk_*), float32 model
  vs float64 oracle, tolerances at the asserts.  Here the inducing locations are a 6 x 6 lattice: with 512 locations on
  this smooth kernel Kzz is numerically singular and BOTH the reference (psd_safe_cholesky) and this path only factor
  it after adding jitter at a precision-dependent retry level, so a 512-point comparison would test the jitter ladder,
  not the arithmetic."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F64 = torch.float64
from conftest import measured  # noqa: E402

# float32 model vs float64 oracle on the 900-row subset (36 inducing locations): ~3x the errors measured on MI355X
# measured: objective 1.3e-8, mean 2.4e-5 (2-norm), variance 1.5e-6 absolute on 0.74
OBJ_TOL, MEAN_TOL, VAR_TOL = 1e-6, 1e-4, 2e-5


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _data(data_dir):
    import pandas as pd
    d = pd.read_csv(os.path.join(data_dir, 'uib_spatio_temporal.csv'))
    xy = torch.tensor(d[['lon', 'lat']].values, dtype=torch.float32)
    y = torch.tensor(d['tp'].values, dtype=torch.float32)
    sx, mx = torch.std_mean(xy, dim=0)
    sy, my = torch.std_mean(y)
    return (xy - mx) / sx, (y - my) / sy


def _inducing(x, M=512, seed=173):
    from sklearn.cluster import KMeans
    Z = torch.tensor(KMeans(M, n_init=1, random_state=seed).fit(x.numpy()).cluster_centers_, dtype=torch.float32)
    # only 43 distinct cells exist: k-means returns duplicate centres; spread them like a user would
    return Z + 0.05 * torch.randn(Z.shape, generator=torch.Generator().manual_seed(0))


def _model(x, y, Z):
    import nsgp.gp as gpytorch
    from models.sparse_multivariate_gibbs_kernel import SparseMultivariateGibbsKernel

    class SparsePSGP(gpytorch.models.ExactGP):
        def __init__(self, train_x, train_y, likelihood, Z):
            super().__init__(train_x, train_y, likelihood)
            self.mean_module = gpytorch.means.ZeroMean()
            base = gpytorch.kernels.ScaleKernel(SparseMultivariateGibbsKernel(Z, 2, Z.clone()))
            self.covar_module = gpytorch.kernels.InducingPointKernel(base, inducing_points=Z.clone(), likelihood=likelihood)
            self.covar_module.inducing_points.requires_grad = False      # they ARE the kernel's inducing locations

        def forward(self, x):
            return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    torch.manual_seed(3)
    lik = gpytorch.likelihoods.GaussianLikelihood()
    model = SparsePSGP(x.cuda(), y.cuda(), lik, Z.cuda()).cuda()
    with torch.no_grad():
        lik.noise = 0.3
        model.covar_module.base_kernel.outputscale = 0.8
    return model, lik


def test_sparse_multivariate_gibbs_gp_on_all_rows_trains_and_predicts(data_dir):
    _need_gpu()
    import nsgp.gp as gpytorch
    x, y = _data(data_dir)
    assert x.shape == (5676, 2)
    Z = _inducing(x)
    model, lik = _model(x, y, Z)
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=0.02)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = -mll(model(model.train_inputs[0]), model.train_targets)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(math.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(x[:200].cuda() + 0.01))
    assert pred.loc.shape == (200,) and bool(torch.isfinite(pred.loc).all())
    v = torch.diagonal(pred.covariance_matrix)
    assert bool(torch.isfinite(v).all()) and float(v.min()) > 0


def test_sparse_multivariate_gibbs_gp_matches_oracle_on_a_subset(data_dir):
    _need_gpu()
    import nsgp.gp as gpytorch
    from oracle import psgibbs, sparse
    x, y = _data(data_dir)
    g = torch.Generator().manual_seed(7)
    idx = torch.randperm(len(x), generator=g)
    # jitter the repeated grid cells so that train / test rows are distinct points
    xj = x + 0.02 * torch.randn(x.shape, generator=g)
    tr, te = idx[:900], idx[900:1050]
    xtr, ytr, xte = xj[tr], y[tr], xj[te]
    u = torch.linspace(-1.8, 1.8, 6)
    Z = torch.stack(torch.meshgrid(u, u, indexing='ij'), -1).reshape(-1, 2) + 0.05 * torch.randn(36, 2, generator=g)
    model, lik = _model(xtr, ytr, Z)
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')                       # no psd_safe jitter retry may be involved in a parity case
        val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    k = model.covar_module.base_kernel.base_kernel
    assert k.D.grad is not None and bool(torch.isfinite(k.D.grad).all())
    sp = torch.nn.functional.softplus
    H, Dm = k.H.detach().cpu().double(), k.D.detach().cpu().double()
    os_ = float(sp(model.covar_module.base_kernel.raw_outputscale.detach().cpu().double()))
    noise = float(sp(lik.noise_covar.raw_noise.detach().cpu().double()) + 1e-4)
    ls = torch.full((1, 2), math.log(2.0), dtype=F64)        # `lengthscale=` is swallowed: softplus(0)
    col = torch.eye(2, dtype=F64)
    Zd, xd, xsd, yd = Z.double(), xtr.double(), xte.double(), ytr.double()

    def K(a, b):
        return os_ * psgibbs.mv_gibbs_forward(a, b, Zd, H, Dm, ls, col, row_os=math.log(2.0))
    Kzz, Kxz, Ksz = K(Zd, Zd), K(xd, Zd), K(xsd, Zd)
    kd_x, kd_s = os_ * torch.ones(len(xd), dtype=F64), os_ * torch.ones(len(xsd), dtype=F64)   # PS kernel diagonal is 1
    ref = sparse.ipk_mll(Kzz, Kxz, kd_x, yd, noise)
    # + the registered matrix-normal prior on H (sparse_multivariate_gibbs_kernel.py:58-62: static row covariance of
    # Z_init under the row kernel, identity column covariance); ExactMarginalLogLikelihood adds log p(H) / N
    from oracle import kernels as OKk
    row = OKk.rbf_ard(Z.float(), Z.float(), torch.full((1, 2), math.log(2.0)), math.log(2.0))      # float32, as built
    prior = psgibbs.MatrixNormalPrior(torch.zeros(36, 2), row, torch.eye(2))
    ref = ref + prior.log_prob(H.float()) / len(xd)
    # float32 kernel build + float32 M x M Cholesky of a kernel matrix with near-duplicate inducing points
    print('[measured] configs[2] subset objective: rel err %.3g (bound %.1g)' % (abs(float(val) - float(ref)) / abs(float(ref)), OBJ_TOL))
    assert abs(float(val) - float(ref)) < OBJ_TOL * abs(float(ref)), (float(val), float(ref))
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(xte.cuda()))
    m_ref, c_ref = sparse.ipk_predict(Kzz, Kxz, kd_x, Ksz, kd_s, yd, noise)
    rel = float((pred.loc.cpu().double() - m_ref).norm() / m_ref.norm())
    print('configs[2] subset: posterior mean 2-norm rel err %.3g, max-norm rel %.3g' % (
        rel, float((pred.loc.cpu().double() - m_ref).abs().max() / m_ref.abs().max())))
    assert rel < MEAN_TOL, rel
    v, v_ref = torch.diagonal(pred.covariance_matrix).cpu().double(), torch.diagonal(c_ref)
    assert measured('configs[2] subset predictive variance', v, v_ref, rtol=VAR_TOL, atol=0.1 * VAR_TOL)


@pytest.mark.parametrize('dt', [torch.float32])
def test_sparse_multivariate_gibbs_gp_at_M512_with_the_jitter_pinned_on_both_sides(data_dir, monkeypatch, dt):
    """configs[2] AT ITS OWN SIZE (M = 512 inducing locations) against the oracle.  Kzz of 512 locations
    on 43 distinct grid cells is numerically singular, so the reference (psd_safe_cholesky) and this path both factor it
    only after a precision-dependent jitter retry -- here the retry ladder is taken out of the comparison: BOTH sides
    factor Kzz + J I with the same J = 1e-2 (the product's `chol_inv_safe` is patched to add exactly J, the oracle
    receives Kzz + J I).  900 training rows, 150 test rows; float32 (what the config runs -- the kernel class, like the
    reference's, creates its parameters H and D in float32 whatever the default dtype, so there is no float64 variant)."""
    _need_gpu()
    import nsgp.gp as gpytorch
    from nsgp import ops
    from nsgp.gp.utils import cholesky as chol_mod
    from oracle import psgibbs, sparse
    J = 1e-2
    x, y = _data(data_dir)
    g = torch.Generator().manual_seed(7)
    idx = torch.randperm(len(x), generator=g)
    xj = x + 0.02 * torch.randn(x.shape, generator=g)
    tr, te = idx[:900], idx[900:1050]
    xtr, ytr, xte = xj[tr], y[tr], xj[te]
    Z = _inducing(x)                                             # 512 k-means centres (+ 0.05 randn), as bench.py's B3 step
    assert Z.shape == (512, 2)

    def pinned(K, jitter=None, max_tries=3):
        W, info = ops.chol_inv(K + J * torch.eye(K.shape[-1], dtype=K.dtype, device=K.device))
        assert int(info.max().item()) == 0
        return W
    monkeypatch.setattr(chol_mod, 'chol_inv_safe', pinned)
    # (the kernel evaluates its row covariance on the inducing locations inside __init__ and keeps them as a plain tensor
    # attribute, like the reference: a float64 model is BUILT under a float64 default dtype, not converted afterwards)
    old_default = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        model, lik = _model(xtr.to(dt), ytr.to(dt), Z.to(dt))
    finally:
        torch.set_default_dtype(old_default)
    model.train(); lik.train()
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(lik, model)
    val = mll(model(model.train_inputs[0]), model.train_targets)
    val.backward()
    k = model.covar_module.base_kernel.base_kernel
    assert k.D.grad is not None and bool(torch.isfinite(k.D.grad).all())
    sp = torch.nn.functional.softplus
    H, Dm = k.H.detach().cpu().double(), k.D.detach().cpu().double()
    os_ = float(sp(model.covar_module.base_kernel.raw_outputscale.detach().cpu().double()))
    noise = float(sp(lik.noise_covar.raw_noise.detach().cpu().double()) + 1e-4)
    ls = torch.full((1, 2), math.log(2.0), dtype=F64)
    col = torch.eye(2, dtype=F64)
    Zd, xd, xsd, yd = Z.double(), xtr.double(), xte.double(), ytr.double()

    def K(a, b):
        return os_ * psgibbs.mv_gibbs_forward(a, b, Zd, H, Dm, ls, col, row_os=math.log(2.0))
    Kzz, Kxz, Ksz = K(Zd, Zd) + J * torch.eye(512, dtype=F64), K(xd, Zd), K(xsd, Zd)
    kd_x, kd_s = os_ * torch.ones(len(xd), dtype=F64), os_ * torch.ones(len(xsd), dtype=F64)
    ref = sparse.ipk_mll(Kzz, Kxz, kd_x, yd, noise)
    from oracle import kernels as OKk
    rdt = torch.float32 if dt == torch.float32 else F64
    row = OKk.rbf_ard(Z.to(rdt), Z.to(rdt), torch.full((1, 2), math.log(2.0), dtype=rdt), math.log(2.0))
    prior = psgibbs.MatrixNormalPrior(torch.zeros(512, 2, dtype=rdt), row, torch.eye(2, dtype=rdt))
    ref = ref + prior.log_prob(H.to(rdt)) / len(xd)
    obj_err = abs(float(val) - float(ref)) / abs(float(ref))
    model.eval(); lik.eval()
    with torch.no_grad():
        pred = lik(model(xte.to(dt).cuda()))
    m_ref, c_ref = sparse.ipk_predict(Kzz, Kxz, kd_x, Ksz, kd_s, yd, noise)
    mean_err = float((pred.loc.cpu().double() - m_ref).abs().max() / m_ref.abs().max())
    v, v_ref = torch.diagonal(pred.covariance_matrix).cpu().double(), torch.diagonal(c_ref)
    var_err = float((v - v_ref).abs().max() / v_ref.abs().max())
    kap = float(torch.linalg.cond(Kzz))
    print('[measured] configs[2] at M=512, J=%.0e, %s: kappa(Kzz + J I) %.3g, objective rel err %.3g, posterior mean '
          'max-norm rel err %.3g, variance max-norm rel err %.3g' % (J, str(dt)[6:], kap, obj_err, mean_err, var_err))
    tol = M512_TOL[dt]
    assert obj_err < tol[0] and mean_err < tol[1] and var_err < tol[2], (obj_err, mean_err, var_err)


# (objective, posterior mean, predictive variance) bounds at M = 512 with J = 1e-2: ~3x the errors measured on MI355X
M512_TOL = {torch.float32: (7e-3, 3e-3, 8e-4)}        # measured over three boxes: <= (2.2e-3, 9.1e-4, 2.4e-4)
