"""SGPR over the Gibbs kernel restated on CPU (oracle; test infrastructure only).

Follows models/gibbs_kernels.py:171-266 (InducingGibbsKernel) and
models/nonstationary_models.py:64-153 (DiagonalSparseGP), with gpytorch's
InducingPointKernelAddedLossTerm / RootLazyTensor._mul_constant semantics [recalled, SURVEY A.5].
"""
import math
import torch
from . import functional as fn
from . import kernels
from .exact import mvn_log_prob


def inducing_inv_root(Z, ell_z):
    """Kzz = Gibbs(Z,Z; ell) (:187-195), R = U^{-1} with Kzz = U^T U (:197-208). No jitter."""
    Kzz = kernels.gibbs(Z, Z, ell_z, ell_z)
    U = torch.linalg.cholesky(Kzz).transpose(-1, -2)
    eye = torch.eye(U.shape[-1], dtype=U.dtype)
    return torch.linalg.solve_triangular(U, eye, upper=True), Kzz


def sgpr_root(x, Z, ell_z, prior):
    """root = K_xz R  (unscaled), ell(x) = conditional mean given (Z, ell_z) (:210-225)."""
    ell_x = prior.conditional_mean_ell(x, Z, ell_z)                            # :212-214
    K_xz = kernels.gibbs(x, Z, ell_x, ell_z)                                   # :222-223
    R, _ = inducing_inv_root(Z, ell_z)
    return K_xz @ R, ell_x                                                     # :225


def sgpr_mll(x, y, Z, log_ell_z, outputscale, noise, prior):
    """`mll(model(x), y)` for DiagonalSparseGP in training mode.

    [log N(y | 0, os*Q + noise I) - 0.5*sum_i (1 - q_ii)/noise + sum_d prior.log_prob_d(Z, log ell_z)] / N
    with q the *unscaled* low-rank diagonal (the ScaleKernel wraps the inducing kernel,
    models/nonstationary_models.py:70-74; added loss at models/gibbs_kernels.py:252-261).
    """
    n = x.shape[-2]
    ell_z = torch.exp(log_ell_z)
    root, _ = sgpr_root(x, Z, ell_z, prior)
    Q = root @ root.transpose(-1, -2)
    cov = outputscale * Q + noise * torch.eye(n, dtype=x.dtype)
    lp = mvn_log_prob(y, torch.zeros_like(y), cov)
    k_diag = torch.ones(n, dtype=x.dtype)                # Gibbs kernel diagonal is exactly 1
    lp = lp - 0.5 * ((k_diag - torch.diagonal(Q)) / noise).sum()
    lp = lp + prior.log_prob(Z, log_ell_z).sum()
    return lp / n


def sgpr_predict(x_train, y_train, Z, log_ell_z, outputscale, noise, prior, x_new,
                 diag_correction=True):
    """DiagonalSparseGP.predict (models/nonstationary_models.py:91-153).

    diag_correction=True is eval mode with gpytorch.settings.sgpr_diagonal_correction on
    (models/gibbs_kernels.py:228-232).  Only the marginals are meaningful (:92-93).
    """
    ntr = x_train.shape[-2]
    ell_z = torch.exp(log_ell_z)
    full = torch.cat([x_train, x_new], dim=-2)                                 # :107
    root, _ = sgpr_root(full, Z, ell_z, prior)
    root = root * math.sqrt(float(outputscale))         # RootLazyTensor._mul_constant
    Lr = root[ntr:, :]                                                         # :128-131
    At = root[:ntr, :] / math.sqrt(float(noise))                               # :133-137
    M = At.shape[-1]
    B = torch.eye(M, dtype=x_train.dtype) + fn.t(At) @ At                      # :142
    mean = fn.mv(Lr, fn.mv(B, fn.mv(fn.t(At), y_train), invert=True)) / math.sqrt(float(noise))  # :144-145
    Kss = Lr @ Lr.transpose(-1, -2)                                            # os * Q_**
    if diag_correction:
        q = torch.diagonal(Kss) / float(outputscale)
        corr = (1.0 - q).clamp(0, math.inf)                                    # gibbs_kernels.py:230
        Kss = Kss + torch.diag(float(outputscale) * corr)
    cov = Kss - Lr @ ((torch.eye(M, dtype=x_train.dtype) - torch.inverse(B)) @ Lr.transpose(-1, -2))  # :147-150
    return mean, cov


# ---------------------------------------------------------------------------------------------------------
# gpytorch.kernels.InducingPointKernel over an arbitrary base kernel [recalled, SURVEY 3.5 / A.5]: SGPR low-rank
# covariance Q = K_xz Kzz^-1 K_zx, Titsias trace term in training, clamped diagonal correction in eval mode.
# Used with the sparse Paciorek-Schervish kernel (BASELINE configs[2]: models/sparse_multivariate_gibbs_kernel.py
# defines K only for x1 == x2 and for pairs where one side is the M inducing locations, which is exactly what an
# inducing-point GP needs).
# ---------------------------------------------------------------------------------------------------------
def ipk_root(Kzz, Kxz):
    """root = K_xz R, R = U^-1 with Kzz = U^T U (triangular_solve(I, chol_upper(Kzz)))."""
    U = torch.linalg.cholesky(Kzz).transpose(-1, -2)
    R = torch.linalg.solve_triangular(U, torch.eye(U.shape[-1], dtype=U.dtype), upper=True)
    return Kxz @ R


def ipk_mll(Kzz, Kxz, kdiag, y, noise, mean=None):
    """ExactMarginalLogLikelihood of an ExactGP whose covar_module is an InducingPointKernel (training mode):
    [log N(y | mu, Q + noise I) - 1/2 sum_i (k_ii - q_ii) / noise] / N."""
    n = y.shape[-1]
    root = ipk_root(Kzz, Kxz)
    Q = root @ root.transpose(-1, -2)
    mu = torch.zeros_like(y) if mean is None else mean
    lp = mvn_log_prob(y, mu, Q + noise * torch.eye(n, dtype=y.dtype))
    lp = lp - 0.5 * ((kdiag - torch.diagonal(Q)) / noise).sum()
    return lp / n


def ipk_predict(Kzz, Kxz, kdiag_x, Ksz, kdiag_s, y, noise, with_noise=True):
    """Eval-mode prediction (sgpr_diagonal_correction on): train covariance Q_xx + diag(clamp(k - q, 0)) + noise I,
    cross covariance Q_sx, test covariance Q_ss + diag(clamp(k - q, 0)) (+ noise I for likelihood(model(x)))."""
    n = y.shape[-1]
    R = ipk_root(Kzz, torch.eye(Kzz.shape[-1], dtype=Kzz.dtype))          # = U^-1
    rx, rs = Kxz @ R, Ksz @ R
    Qxx, Qsx, Qss = rx @ rx.T, rs @ rx.T, rs @ rs.T
    Kt = Qxx + torch.diag((kdiag_x - torch.diagonal(Qxx)).clamp(0, math.inf)) + noise * torch.eye(n, dtype=y.dtype)
    L = torch.linalg.cholesky(Kt)
    mean = Qsx @ torch.cholesky_solve(y.unsqueeze(-1), L).squeeze(-1)
    V = torch.linalg.solve_triangular(L, Qsx.T, upper=False)
    cov = Qss + torch.diag((kdiag_s - torch.diagonal(Qss)).clamp(0, math.inf)) - V.T @ V
    if with_noise:
        cov = cov + noise * torch.eye(Ksz.shape[-2], dtype=y.dtype)
    return mean, cov
