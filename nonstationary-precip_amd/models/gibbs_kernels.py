"""Gibbs kernel, lengthscale prior processes and SGPR wrappers on the MI355X engine -- drop-in for
models/gibbs_kernels.py of the reference (same names, argument meaning and error behaviour).

  PositivePriorProcess                       reference models/gibbs_kernels.py:35-59
  LogNormalPriorProcess(input_dim=1, covariance_function=None, active_dims=None)      :61-109
  GibbsKernel(*a, lengthscale_prior=None, **kw).forward(x1, x2, ell1=None, ell2=None) :111-162
  GibbsSafeScaleKernel                                                                 :164-168
  InducingGibbsKernel(base_kernel, inducing_points, likelihood, active_dims=None)      :171-266
  InducingGibbsKernelST (same, slicing inducing points by active_dims)                 :268-363

What runs where: the K matrix is one gfx950 launch (nsgp_gibbs_build_fwd, with the ScaleKernel's
outputscale folded in) instead of 8 full-size temporaries; the prior's D batched RBF Gram matrices,
the Cholesky / triangular inverse of Kzz and every product go to the pairwise / potrf / MFMA kernels.
Quirks kept: conditional_sample returns the conditional MEAN; log_prob is divided
by N; ell1 / ell2 are cached on the module as a side effect.  Fixed: the CPU-only jitter eye (:88).
"""
import math
from typing import Optional, Tuple

import torch

import nsgp.gp as gpytorch
from nsgp import ops
from nsgp.gp.utils.cholesky import chol_inv_safe
from nsgp.gp import settings
from nsgp.gp.kernels import same_points
from nsgp.gp.lazy import (delazify, LowRankRootLazyTensor, LowRankRootAddedDiagLazyTensor, DiagLazyTensor,
                          MatmulLazyTensor)


class PositivePriorProcess(torch.nn.Module):
    """Base class of lengthscale prior processes: forward returns the distribution of the
    unconstrained value, sample / conditional_sample return positive values."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__()

    def forward(self, x):
        raise NotImplementedError

    def sample(self, x, **kwargs):
        raise NotImplementedError

    def conditional_sample(self, x, given, **kwargs):
        raise NotImplementedError


class LogNormalPriorProcess(PositivePriorProcess):
    """D independent GPs on log ell: ConstantMean(batch D) + ScaleKernel(RBF(ard=input_dim, batch D))."""

    def __init__(self, input_dim: int = 1, covariance_function=None, active_dims=None) -> None:
        super().__init__()
        bshape = torch.Size((input_dim,))
        self.mean_module = gpytorch.means.ConstantMean(batch_shape=bshape)
        if covariance_function is None:
            covariance_function = gpytorch.kernels.ScaleKernel(
                gpytorch.kernels.RBFKernel(ard_num_dims=input_dim, batch_shape=bshape, active_dims=active_dims),
                batch_shape=bshape, active_dims=active_dims)
        self.covar_module = covariance_function

    def forward(self, x):
        """Distribution of the log-value at x: batch (D,), event n."""
        return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def sample(self, x, **kwargs):
        return torch.exp(self.forward(x).rsample(**kwargs))

    def conditional_sample(self, x, given: Tuple[torch.Tensor, torch.Tensor], **kwargs):
        """exp of the conditional mean of log ell at x given (x_g, ell_g); jitter 1e-4; returns (D, n)."""
        x_g, ell_g = given
        n_g = x_g.shape[-2]
        mean_g = self.mean_module(x_g)                                              # (D, n_g)
        K_gg = self.covar_module(x_g).evaluate()                                    # (D, n_g, n_g)
        K_xg = self.covar_module(x, x_g).evaluate()                                 # (D, n, n_g)
        K_gg = K_gg + 1e-4 * torch.eye(n_g, dtype=K_gg.dtype, device=K_gg.device)
        W, _ = ops.chol_inv(K_gg)                                                   # K^-1 = W^T W, batched
        resid = (torch.log(ell_g) - mean_g).unsqueeze(-1)                           # (D, n_g, 1)
        alpha = ops.matmul(W, ops.matmul(W, resid, a_lower=True), True, False, a_lower=True)
        mu = self.mean_module(x) + ops.matmul(K_xg, alpha).squeeze(-1)              # (D, n)
        return torch.exp(mu)

    def log_prob(self, x_and_logell: Tuple[torch.Tensor, torch.Tensor]):
        """MVN log-density of log ell with +1e-4 I on the covariance, divided by N -> (D,)."""
        x, log_value = x_and_logell
        n = x.shape[-2]
        mu = self.mean_module(x)
        sigma = self.covar_module(x).evaluate()
        sigma = sigma + 1e-4 * torch.eye(n, dtype=sigma.dtype, device=sigma.device)
        return gpytorch.distributions.MultivariateNormal(mu, sigma).log_prob(log_value) / n


class GibbsKernel(gpytorch.kernels.Kernel):
    """Diagonal Gibbs kernel (Rasmussen & Williams eq. 4.32) with a prior process on ell(x)."""

    is_stationary = False
    fuses_outputscale = True           # GibbsSafeScaleKernel folds its outputscale into the same launch
    fuses_diag_add = True              # ... and likelihood(dist) folds noise * I into it too

    def __init__(self, *args, lengthscale_prior: PositivePriorProcess = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.lengthscale_prior = lengthscale_prior

    @property
    def batch_shape(self):
        """Own batch shape only: the prior's (D,)-batched sub-kernel must not leak a batch dim."""
        return self._batch_shape

    def forward(self, x1, x2, ell1: Optional[torch.Tensor] = None, ell2: Optional[torch.Tensor] = None,
                _outputscale=None, _diag_add=None, **kwargs):
        """ell1 / ell2 are the lengthscales (D, n) at x1 / x2.  Missing ell1 is sampled from the prior;
        if x1 and x2 differ and ell2 is missing it is the conditional mean given (x1, ell1)."""
        if ell1 is None:
            ell1 = self.lengthscale_prior.sample(x1)
            self.ell1 = ell1
        if same_points(x1, x2):
            ell2 = ell1
        elif ell2 is None:
            ell2 = self.lengthscale_prior.conditional_sample(x2, given=(x1, ell1))
            self.ell2 = ell2
        return ops.gibbs_kernel(x1, x2, ell1, ell2, _outputscale, _diag_add)


class GibbsSafeScaleKernel(gpytorch.kernels.ScaleKernel):
    @property
    def batch_shape(self):
        return self._batch_shape


class InducingGibbsKernel(gpytorch.kernels.InducingPointKernel):
    """SGPR wrapper: call with ell = lengthscales at the inducing points; the lengthscales at the
    train/test points are the conditional mean given those."""

    def __init__(self, base_kernel: GibbsKernel, inducing_points: torch.Tensor, likelihood,
                 active_dims: Optional[Tuple[int, ...]] = None):
        super().__init__(base_kernel, inducing_points, likelihood, active_dims)

    @property
    def batch_shape(self):
        return self._batch_shape

    def _z(self):
        return self.inducing_points

    def _inducing_mat(self, ell=None):
        if not self.training and hasattr(self, '_cached_kernel_mat'):
            return self._cached_kernel_mat
        z = self._z()
        res = delazify(self.base_kernel(z, z, ell1=ell))
        if not self.training:
            self._cached_kernel_mat = res
        return res

    def _inducing_inv_root(self, ell=None):
        """R with R R^T = Kzz^-1: the reference's triangular_solve(I, chol_upper(Kzz)) = U^-1 = (L^-1)^T."""
        if not self.training and hasattr(self, '_cached_kernel_inv_root'):
            return self._cached_kernel_inv_root
        W = chol_inv_safe(self._inducing_mat(ell))               # W = L^-1 (lower); jitter only on failure
        res = W.transpose(-1, -2)
        if not self.training:
            self._cached_kernel_inv_root = res
        return res

    def _get_covariance(self, x1, x2, ell):
        prior, z = self.base_kernel.lengthscale_prior, self._z()
        equal = same_points(x1, x2)
        if equal:
            ell1 = prior.conditional_sample(x1, given=(z, ell))
            ell2 = ell1
        else:
            ell_cond = prior.conditional_sample(torch.cat((x1, x2), dim=-2), given=(z, ell))
            ell1 = ell_cond[..., :x1.shape[-2]]
            ell2 = ell_cond[..., x1.shape[-2]:]
        k_ux1 = delazify(self.base_kernel(x1, z, ell1=ell1, ell2=ell))
        R = self._inducing_inv_root(ell)
        root1 = ops.matmul(k_ux1, R, b_lower=False)
        if equal:
            covar = LowRankRootLazyTensor(root1)
            if not self.training and settings.sgpr_diagonal_correction.on():
                k_diag = self.base_kernel(x1, x2, diag=True, ell1=ell1, ell2=ell2)
                correction = (k_diag - covar.diag()).clamp(0, math.inf)
                covar = LowRankRootAddedDiagLazyTensor(covar, DiagLazyTensor(correction))
        else:
            k_ux2 = delazify(self.base_kernel(x2, z, ell1=ell2, ell2=ell))
            covar = MatmulLazyTensor(root1, ops.matmul(k_ux2, R).transpose(-1, -2))
        return covar, ell1, ell2

    def _covar_diag(self, inputs, ell):
        if inputs.ndimension() == 1:
            inputs = inputs.unsqueeze(1)
        # the Gibbs kernel's diagonal is exactly 1 (unscaled base kernel)
        return DiagLazyTensor(torch.ones(inputs.shape[:-1], dtype=inputs.dtype, device=inputs.device))

    def forward(self, x1, x2, diag=False, ell=None, **kwargs):
        covar, ell1, ell2 = self._get_covariance(x1, x2, ell=ell)
        if self.training:
            if not same_points(x1, x2):
                raise RuntimeError('x1 should equal x2 in training mode')
            zero_mean = torch.zeros_like(x1.select(-1, 0))
            term = gpytorch.mlls.InducingPointKernelAddedLossTerm(
                gpytorch.distributions.MultivariateNormal(zero_mean, self._covar_diag(x1, ell1)),
                gpytorch.distributions.MultivariateNormal(zero_mean, covar), self.likelihood)
            self.update_added_loss_term('inducing_point_loss_term', term)
        return covar.diag() if diag else covar


class InducingGibbsKernelST(InducingGibbsKernel):
    """Spatio-temporal variant: the inducing points carry all input columns, the Gibbs kernel sees
    inducing_points[:, active_dims]."""

    def _z(self):
        return self.inducing_points[:, self.active_dims]
