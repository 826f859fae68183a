"""Exact-GP paths restated on CPU (oracle; test infrastructure only).

* LogNormalPrior          -- models/gibbs_kernels.py:61-109 (LogNormalPriorProcess)
* gibbs_exact_mll         -- models/nonstationary_models.py:22-43 under gpytorch's
                             ExactMarginalLogLikelihood [recalled, SURVEY A.5]
* gibbs_exact_predict     -- models/nonstationary_models.py:45-62
* seard_mll / seard_predict -- models/dgps.py:113-122 (ExactGPModel) with ScaleKernel(RBF-ARD),
                             as driven by experiments/seard_spatial_benchmark.py:51-106
"""
import math
import torch
from . import functional as fn
from . import kernels


def mvn_log_prob(y, mean, cov):
    """log N(y | mean, cov) through a Cholesky factor (what gpytorch does for N <= 800)."""
    L = torch.linalg.cholesky(cov)
    d = (y - mean).unsqueeze(-1)
    a = torch.linalg.solve_triangular(L, d, upper=False).squeeze(-1)
    n = y.shape[-1]
    return -0.5 * (a * a).sum(-1) - torch.log(torch.diagonal(L, dim1=-1, dim2=-2)).sum(-1) \
        - 0.5 * n * math.log(2 * math.pi)


class LogNormalPrior:
    """D independent GPs on log-lengthscale (models/gibbs_kernels.py:61-109).

    mean_const:(D,)  lengthscale:(D,Din)  outputscale:(D,)  -- constrained (actual) values of
    ConstantMean(batch D) + ScaleKernel(RBF(ard=Din, batch D), batch D)  (:65-70).
    """

    def __init__(self, mean_const, lengthscale, outputscale):
        self.mean_const = mean_const
        self.lengthscale = lengthscale
        self.outputscale = outputscale

    def mean(self, x):
        return self.mean_const.unsqueeze(-1).expand(-1, x.shape[-2])          # (D,n)

    def cov(self, x1, x2):
        return kernels.rbf_ard(x1.unsqueeze(0), x2.unsqueeze(0),
                               self.lengthscale.unsqueeze(-2), self.outputscale)  # (D,n1,n2)

    def conditional_mean_ell(self, x, x_given, ell_given):
        """exp of the conditional mean only; jitter 1e-4 (models/gibbs_kernels.py:80-100)."""
        mean_g = self.mean(x_given)                                            # :83
        K_gg = self.cov(x_given, x_given)
        K_xg = self.cov(x, x_given).permute(1, 0, 2)                           # :85-86 (n,D,ng)
        prior_mean = self.mean(x).permute(1, 0)                                # :87   (n,D)
        jitter = 1e-4 * torch.eye(x_given.shape[-2], dtype=x.dtype)            # :88
        mu = prior_mean + fn.dot(K_xg, fn.mv(K_gg + jitter,
                                             torch.log(ell_given) - mean_g, invert=True))  # :89-93
        return torch.exp(mu).permute(1, 0)                                     # :100  (D,n)

    def log_prob(self, x, log_ell):
        """MVN log-density with +1e-4 I, divided by N (models/gibbs_kernels.py:102-109) -> (D,)."""
        n = x.shape[-2]
        sigma = self.cov(x, x) + 1e-4 * torch.eye(n, dtype=x.dtype)
        return mvn_log_prob(log_ell, self.mean(x), sigma) / n


def gibbs_exact_mll(x, y, log_ell, outputscale, noise, prior):
    """The scalar `mll(model(x), y)` of experiments/spatial_exp.py:200-201.

    [log N(y | 0, os*K_gibbs + noise I) + sum_d prior.log_prob_d] / N
    (models/nonstationary_models.py:35-43; ExactMarginalLogLikelihood recalled).
    """
    n = x.shape[-2]
    ell = torch.exp(log_ell)
    K = outputscale * kernels.gibbs(x, x, ell, ell)
    cov = K + noise * torch.eye(n, dtype=x.dtype)
    lp = mvn_log_prob(y, torch.zeros_like(y), cov)
    lp = lp + prior.log_prob(x, log_ell).sum()
    return lp / n


def gibbs_exact_predict(x_train, y_train, log_ell, outputscale, noise, prior, x_new):
    """Predictive mean / covariance of DiagonalExactGP.predict (models/nonstationary_models.py:45-62).

    Returns (mu, sigma + 1e-4 I) exactly as the reference forms them (explicit inverse at :57-58).
    """
    ell_tr = torch.exp(log_ell)
    n = x_train.shape[-2]
    K_xx = outputscale * kernels.gibbs(x_train, x_train, ell_tr, ell_tr)                     # :48
    ell2 = prior.conditional_mean_ell(x_new, x_train, ell_tr)                                # :49-50
    K_ss = outputscale * kernels.gibbs(x_new, x_new, ell2, ell2)                             # :51
    K_sx = outputscale * kernels.gibbs(x_new, x_train, ell2, ell_tr)                         # :52-53
    Kn = K_xx + noise * torch.eye(n, dtype=x_train.dtype)
    mu = fn.dot(K_sx, fn.mv(Kn, y_train, invert=True))                                       # :55-56
    sigma = K_ss - K_sx @ torch.inverse(Kn) @ fn.t(K_sx)                                     # :57-58
    return mu, sigma + 1e-4 * torch.eye(x_new.shape[-2], dtype=x_train.dtype), ell2          # :60-61


def seard_mll(x, y, lengthscale, outputscale, noise, mean_const):
    """ExactMarginalLogLikelihood of ExactGPModel(ScaleKernel(RBF-ARD)) (models/dgps.py:113-122)."""
    n = x.shape[-2]
    K = kernels.rbf_ard(x, x, lengthscale, outputscale) + noise * torch.eye(n, dtype=x.dtype)
    return mvn_log_prob(y, mean_const.expand(n), K) / n


def seard_predict(x, y, lengthscale, outputscale, noise, mean_const, x_new, with_noise=True):
    """gpytorch eval-mode prediction [recalled, SURVEY A.6]; `likelihood(model(x))` adds noise."""
    n = x.shape[-2]
    K = kernels.rbf_ard(x, x, lengthscale, outputscale) + noise * torch.eye(n, dtype=x.dtype)
    K_sx = kernels.rbf_ard(x_new, x, lengthscale, outputscale)
    K_ss = kernels.rbf_ard(x_new, x_new, lengthscale, outputscale)
    L = torch.linalg.cholesky(K)
    alpha = torch.cholesky_solve((y - mean_const).unsqueeze(-1), L).squeeze(-1)
    mean = mean_const + K_sx @ alpha
    V = torch.linalg.solve_triangular(L, K_sx.transpose(-1, -2), upper=False)
    cov = K_ss - V.transpose(-1, -2) @ V
    if with_noise:
        cov = cov + noise * torch.eye(x_new.shape[-2], dtype=x.dtype)
    return mean, cov
