"""Spatio-temporal additive GPs restated on CPU (oracle; test infrastructure only; parity unpinned like every
gpytorch-backed piece, oracle/__init__.py).

Follows models/spatio_temporal_models.py:17-33 (SpatioTemporal_Stationary): zero mean,
k = os_t RBF(t; l_t) Periodic(t; l_p, p) + os_s RBF-ARD((lon, lat); l_s), exact GP or -- with inducing points --
gpytorch's InducingPointKernel over the summed kernel (SGPR low-rank covariance + Titsias trace term,
SURVEY A.5), and experiments/spatio_temporal_exp.py:139-163 (ExactMarginalLogLikelihood, eval-mode prediction).
"""
import math

import torch

from . import kernels
from .exact import mvn_log_prob


def st_kernel(x1, x2, p):
    """p: dict(os_t, ls_t, ls_p, period, os_s, ls_s:(2,)); x columns (t, lon, lat)."""
    t1, t2 = x1[..., 0:1], x2[..., 0:1]
    kt = p['os_t'] * kernels.rbf_ard(t1, t2, p['ls_t'].reshape(1, 1)) * kernels.periodic(t1, t2, p['ls_p'], p['period'])
    ks = p['os_s'] * kernels.rbf_ard(x1[..., 1:3], x2[..., 1:3], p['ls_s'].reshape(1, 2))
    return kt + ks


def st_exact_mll(x, y, p, noise):
    n = x.shape[-2]
    K = st_kernel(x, x, p) + noise * torch.eye(n, dtype=x.dtype)
    return mvn_log_prob(y, torch.zeros_like(y), K) / n


def st_exact_predict(x, y, p, noise, x_new, with_noise=True):
    n = x.shape[-2]
    K = st_kernel(x, x, p) + noise * torch.eye(n, dtype=x.dtype)
    Ks = st_kernel(x_new, x, p)
    L = torch.linalg.cholesky(K)
    alpha = torch.cholesky_solve(y.unsqueeze(-1), L)
    mean = (Ks @ alpha).squeeze(-1)
    V = torch.linalg.solve_triangular(L, Ks.transpose(-1, -2), upper=False)
    cov = st_kernel(x_new, x_new, p) - V.transpose(-1, -2) @ V
    if with_noise:
        cov = cov + noise * torch.eye(x_new.shape[-2], dtype=x.dtype)
    return mean, cov


def st_sgpr_mll(x, y, z, p, noise):
    """[log N(y | 0, Q + noise I) - 1/2 sum_i (k_ii - q_ii)/noise] / N with Q = K_xz Kzz^-1 K_zx over the SUMMED
    kernel (the InducingPointKernel wraps temporal + spatial, spatio_temporal_models.py:25-27)."""
    n = x.shape[-2]
    Kzz = st_kernel(z, z, p)
    U = torch.linalg.cholesky(Kzz).transpose(-1, -2)
    R = torch.linalg.solve_triangular(U, torch.eye(z.shape[-2], dtype=x.dtype), upper=True)
    root = st_kernel(x, z, p) @ R
    Q = root @ root.transpose(-1, -2)
    lp = mvn_log_prob(y, torch.zeros_like(y), Q + noise * torch.eye(n, dtype=x.dtype))
    k_diag = torch.diagonal(st_kernel(x, x, p))
    lp = lp - 0.5 * ((k_diag - torch.diagonal(Q)) / noise).sum()
    return lp / n


# ---------------------------------------------------------------------------------------------------------
# SparseSpatioTemporal_Nonstationary (models/spatio_temporal_models.py:35-126) over InducingGibbsKernelST
# (models/gibbs_kernels.py:268-363).  gpytorch mechanics [recalled, SURVEY A.2/A.5]: Kernel.__call__ selects
# `active_dims` columns; ScaleKernel.forward calls base_kernel.forward directly (the wrapped kernel's own
# active_dims are NOT applied a second time); InducingPointKernel = SGPR low-rank root + Titsias trace term in
# training + clamped diagonal correction in eval mode.
# ---------------------------------------------------------------------------------------------------------
def _temporal_kernel(t1, t2, p):
    """ScaleKernel(RBFKernel(active_dims=0) * PeriodicKernel(active_dims=0), outputscale > 7) on the time column
    (spatio_temporal_models.py:42)."""
    return p['os_t'] * kernels.rbf_ard(t1, t2, p['ls_t'].reshape(1, 1)) * kernels.periodic(t1, t2, p['ls_p'], p['period'])


def _inv_root(Kzz):
    """triangular_solve(I, chol_upper(Kzz)) = U^-1 (gibbs_kernels.py:298-300; gpytorch InducingPointKernel)."""
    U = torch.linalg.cholesky(Kzz).transpose(-1, -2)
    return torch.linalg.solve_triangular(U, torch.eye(U.shape[-1], dtype=U.dtype), upper=True)


def st_ns_roots(x, z, log_ell_z, p, prior):
    """Low-rank roots of the two SGPR components at the rows of x (columns t, lon, lat):
    temporal  root_t = K_t(x_t, z_t) R_t   over z[:, 0]                (InducingPointKernel, active_dims=(0))
    spatial   root_s = Gibbs(x_s, z_s; ell(x_s), ell_z) R_s  over z[:, (1, 2)], ell(x_s) = conditional mean given
              (z_s, ell_z)                                              (gibbs_kernels.py:310-322)."""
    from .sparse import sgpr_root
    xt, zt = x[..., 0:1], z[..., 0:1]
    root_t = _temporal_kernel(xt, zt, p) @ _inv_root(_temporal_kernel(zt, zt, p))
    root_s, _ = sgpr_root(x[..., 1:3], z[..., 1:3], torch.exp(log_ell_z), prior)
    return root_t, root_s


def st_ns_mll(x, y, z, log_ell_z, p, noise, prior):
    """`mll(model(x), y)` of SparseSpatioTemporal_Nonstationary in training mode under ExactMarginalLogLikelihood:
    [log N(y | 0, Q_t + os_s Q_s + noise I) - 1/2 sum_i (os_t - q_t,ii)/noise - 1/2 sum_i (1 - q_s,ii)/noise
     + sum_d prior.log_prob_d(z[:, (0, 1)], log_ell_z)] / N.
    The spatial trace term sees the UNSCALED Gibbs kernel (the ScaleKernel wraps the inducing kernel, :41); the
    registered prior's closure hands the full 3-column inducing points to the prior (:52-55), whose covariance
    selects ITS active_dims (0, 1) -- i.e. (time, lon): a reference quirk, kept."""
    n = x.shape[-2]
    root_t, root_s = st_ns_roots(x, z, log_ell_z, p, prior)
    Qt, Qs = root_t @ root_t.transpose(-1, -2), root_s @ root_s.transpose(-1, -2)
    cov = Qt + p['os_s'] * Qs + noise * torch.eye(n, dtype=x.dtype)
    lp = mvn_log_prob(y, torch.zeros_like(y), cov)
    lp = lp - 0.5 * ((p['os_t'] - torch.diagonal(Qt)) / noise).sum()
    lp = lp - 0.5 * ((1.0 - torch.diagonal(Qs)) / noise).sum()
    lp = lp + prior.log_prob(z[..., 0:2], log_ell_z).sum()
    return lp / n


def st_ns_predict(x_train, y_train, z, log_ell_z, p, noise, prior, x_new):
    """SparseSpatioTemporal_Nonstationary.predict (spatio_temporal_models.py:62-126), eval mode with
    sgpr_diagonal_correction on.  The joint covariance over [train; test] is a SUM of two lazy tensors, so the
    reference takes its dense branch (:104,110-111): L = C[n:, :], A^T = C[:n, :]/sigma, B = I + A A^T over all
    n + n* columns -- not a valid SGPR predictive (the docstring warns); the arithmetic is restated as written."""
    ntr = x_train.shape[-2]
    full = torch.cat([x_train, x_new], dim=-2)                                              # :79
    root_t, root_s = st_ns_roots(full, z, log_ell_z, p, prior)
    Qt, Qs = root_t @ root_t.transpose(-1, -2), root_s @ root_s.transpose(-1, -2)
    corr_t = (p['os_t'] - torch.diagonal(Qt)).clamp(0, math.inf)                            # InducingPointKernel, eval
    corr_s = (1.0 - torch.diagonal(Qs)).clamp(0, math.inf)                                  # gibbs_kernels.py:327-330
    C = Qt + torch.diag(corr_t) + p['os_s'] * (Qs + torch.diag(corr_s))                     # :60
    sig = math.sqrt(float(noise))
    L = C[ntr:, :]                                                                          # :104
    At = C[:ntr, :] / sig                                                                   # :110-111
    m = At.shape[-1]
    eye = torch.eye(m, dtype=C.dtype)
    B = eye + At.transpose(-1, -2) @ At                                                     # :113
    mean = L @ torch.linalg.solve(B, At.transpose(-1, -2) @ y_train) / sig                  # :115-116
    cov = C[ntr:, ntr:] - L @ ((eye - torch.inverse(B)) @ L.transpose(-1, -2))              # :118-122
    return mean, cov
