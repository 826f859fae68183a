"""MAP exact GP and SGPR GP over the Gibbs kernel on the MI355X engine -- drop-in for
models/nonstationary_models.py of the reference.

  DiagonalExactGP(train_x, train_y, likelihood, prior, num_dim=1)      reference :22-62
  DiagonalSparseGP(train_x, train_y, likelihood, prior, z, num_dim=1)  reference :64-153
Both behave like an ExactGP in training (forward on the training inputs, trained with
ExactMarginalLogLikelihood); predictions are made with .predict(x_new), which returns an MVN with
.loc / .covariance_matrix / .log_prob like the reference.  The reference's explicit torch.inverse
calls (:57, :149) are Cholesky solves here (tolerance-equal); every matrix build,
factorisation and product runs on the gfx950 kernels.
"""
import math

import torch

import nsgp.gp as gpytorch
from nsgp import ops
from models.gibbs_kernels import GibbsKernel, GibbsSafeScaleKernel, InducingGibbsKernel


class DiagonalExactGP(gpytorch.models.ExactGP):
    """MAP inference of a diagonal-Gibbs-kernel GP: log ell at the training points is a parameter
    (D, N), initialised at the prior mean and regularised by the prior process."""

    def __init__(self, train_x, train_y, likelihood, prior, num_dim=1):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gpytorch.means.ZeroMean()
        self.covar_module = GibbsSafeScaleKernel(GibbsKernel(lengthscale_prior=prior, ard_num_dims=num_dim))
        self.register_parameter('log_ell_train_x', torch.nn.Parameter(prior.mean_module(train_x).detach().clone()))
        self.register_prior('ell_train_prior', prior, lambda module: (module.train_inputs[0], module.log_ell_train_x))

    def forward(self, x):
        return gpytorch.distributions.MultivariateNormal(
            self.mean_module(x), self.covar_module(x, ell1=torch.exp(self.log_ell_train_x)))

    def predict(self, x_new):
        """Predictive at x_new given the MAP lengthscales at the training points; covariance + 1e-4 I."""
        x_tr, y_tr = self.train_inputs[0], self.train_targets
        ell_tr = torch.exp(self.log_ell_train_x)
        prior = self.covar_module.base_kernel.lengthscale_prior
        n, ns = x_tr.shape[-2], x_new.shape[-2]
        ell_new = prior.conditional_sample(x_new, given=(x_tr, ell_tr))
        K_xx = self.covar_module(x_tr, ell1=ell_tr).add_diag(self.likelihood.noise).evaluate()
        K_ss = self.covar_module(x_new, ell1=ell_new).evaluate()
        K_sx = self.covar_module(x_new, x_tr, ell1=ell_new, ell2=ell_tr).evaluate()
        W, _ = ops.chol_inv(K_xx)                                   # (K_xx + noise I)^-1 = W^T W
        V = ops.matmul(W, K_sx, False, True, a_lower=True)          # W K_xs        (n, ns)
        a = ops.matmul(W, y_tr.unsqueeze(-1), a_lower=True)         # W y
        mu = ops.matmul(V, a, True, False).squeeze(-1)
        sigma = K_ss - ops.matmul(V, V, True, False)
        sigma = sigma + 1e-4 * torch.eye(ns, dtype=sigma.dtype, device=sigma.device)
        return gpytorch.distributions.MultivariateNormal(mu, sigma)


class DiagonalSparseGP(gpytorch.models.ExactGP):
    """MAP inference of the sparse (SGPR) Gibbs-kernel GP: log ell lives at the inducing points."""

    def __init__(self, train_x, train_y, likelihood, prior, z, num_dim=1):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gpytorch.means.ZeroMean()
        self.covar_module = GibbsSafeScaleKernel(
            InducingGibbsKernel(GibbsKernel(lengthscale_prior=prior, ard_num_dims=num_dim), z, likelihood))
        self.register_parameter('log_ell_z', torch.nn.Parameter(prior.mean_module(z).detach().clone()))
        self.register_prior('ell_z_prior', prior,
                            lambda module: (module.covar_module.base_kernel.inducing_points, module.log_ell_z))

    def forward(self, x, ell=None):
        return gpytorch.distributions.MultivariateNormal(
            self.mean_module(x), self.covar_module(x, ell=torch.exp(self.log_ell_z)))

    def predict(self, x_new):
        """SGPR predictive at x_new (only the marginals are meaningful, as in the reference).

        With root = sqrt(os) K_.z Kzz^-1/2 over [train; test]:  A^T = root_train / sigma,
        B = I + A A^T,  mean = L B^-1 A y / sigma,  cov = K_** - L (I - B^-1) L^T.
        """
        x_tr, y_tr = self.train_inputs[0], self.train_targets
        if x_new.ndimension() == 1:
            x_new = x_new.unsqueeze(-1)
        ntr = x_tr.shape[-2]
        full_output = self.forward(torch.cat([x_tr, x_new], dim=-2))
        full_covar = full_output.lazy_covariance_matrix.evaluate_kernel()
        low_rank = full_covar if isinstance(full_covar, gpytorch.lazy.LowRankRootLazyTensor) \
            else full_covar._lazy_tensor
        root = low_rank.root.evaluate()
        sigma = torch.sqrt(self.likelihood.noise)
        L = root[ntr:, :].contiguous()
        At = (root[:ntr, :] / sigma).contiguous()
        M = At.shape[-1]
        eye = torch.eye(M, dtype=At.dtype, device=At.device)
        B = eye + ops.matmul(At, At, True, False)
        Wb, _ = ops.chol_inv(B)                                      # B^-1 = Wb^T Wb
        v = ops.matmul(At, y_tr.unsqueeze(-1), True, False)          # A y
        Binv_v = ops.matmul(Wb, ops.matmul(Wb, v, a_lower=True), True, False, a_lower=True)
        mean = ops.matmul(L, Binv_v).squeeze(-1) / sigma + full_output.loc[ntr:]
        LW = ops.matmul(L, Wb, False, True, b_lower=True)            # L Wb^T
        K_ss = full_covar.evaluate()[ntr:, ntr:]
        cov = K_ss - ops.matmul(L, L, False, True) + ops.matmul(LW, LW, False, True)
        return full_output.__class__(mean.contiguous(), cov)
