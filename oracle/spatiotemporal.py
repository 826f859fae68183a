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
