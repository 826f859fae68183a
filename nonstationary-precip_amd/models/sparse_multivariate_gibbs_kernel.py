"""Sparse Paciorek-Schervish kernel: the latent H lives at M inducing locations Z -- drop-in for
models/sparse_multivariate_gibbs_kernel.py of the reference (:20-154), whose broken import
`kernels.latent_priors` (:11) is fixed to models.latent_priors.

  SparseMultivariateGibbsKernel(Z, input_dim, Z_init)  with .H (M,2), .D (2,2),
      .expectation_conditional_matrix_variate_dist(x_star), .forward(x1, x2, diag=False)
Differences from the dense variant, as in the reference: row kernel ScaleKernel(RBF) (the
`lengthscale=[1.3, 1.1]` keyword is swallowed by gpytorch's Kernel.__init__),
column covariance I, the PRIOR uses the static row covariance of Z_init while the conditional mean
re-inverts the row covariance of the CURRENT Z at every call (:69).
"""
import torch

import nsgp.gp as gpytorch
from nsgp import ops
from nsgp.gp.kernels import RBFKernel, ScaleKernel, same_points
from models.latent_priors import MatrixVariateNormalPrior
from models.multivariate_gibbs_kernel import _sigma

jitter = 1e-5


class SparseMultivariateGibbsKernel(gpytorch.kernels.Kernel):
    is_stationary = False

    def __init__(self, Z, input_dim, Z_init, **kwargs):
        super().__init__(**kwargs)
        self.inducing_locations = Z
        self.d = input_dim
        self.m = self.inducing_locations.shape[0]
        if input_dim == 1:
            raise ValueError('Use gibbs 1d kernel for dim 1')
        dev = Z.device
        self.row_covar_kernel = ScaleKernel(RBFKernel(ard_num_dims=self.d, lengthscale=torch.Tensor([1.3, 1.1]))).to(dev)
        self.row_covar_kernel.requires_grad_(False)
        self.loc = torch.zeros(self.m, self.d, device=dev)
        self.row_covar = self.row_covar_kernel(self.inducing_locations).evaluate()
        self.static_row_covar = self.row_covar_kernel(Z_init).evaluate()
        Z_init.requires_grad_(False)
        self.col_covar = torch.tensor([[1., 0.], [0., 1.]], device=dev)
        self.H_matrix_prior = MatrixVariateNormalPrior(self.loc, row_covariance_matrix=self.static_row_covar,
                                                       column_covariance_matrix=self.col_covar)
        H_init = self.H_matrix_prior.sample_n(1)
        self.register_parameter(name='H', parameter=torch.nn.Parameter(H_init.to(torch.float32)))
        self.register_prior('prior_H', self.H_matrix_prior, 'H')
        D_init = torch.diag(torch.randn(2))
        self.register_parameter(name='D', parameter=torch.nn.Parameter(D_init.to(torch.float32).to(dev)))

    def expectation_conditional_matrix_variate_dist(self, x_star):
        eye = torch.eye(self.m, dtype=self.row_covar.dtype, device=self.row_covar.device)
        W, _ = ops.chol_inv((self.row_covar.detach() + eye * jitter).double().contiguous())
        rhs = ops.matmul(W, self.H.detach().double().contiguous(), a_lower=True)
        sol = ops.matmul(W, rhs, True, False, a_lower=True).to(self.H.dtype)      # (row + jI)^-1 H
        cross = self.row_covar_kernel(x_star, self.inducing_locations).evaluate()
        return ops.gemm(cross.contiguous(), sol.contiguous())

    def _latent(self, x1, x2):
        H = self.H.detach()
        if same_points(x1, x2):
            Hx = H if len(x1) == H.shape[0] else self.expectation_conditional_matrix_variate_dist(x1).detach()
            return Hx, Hx
        if x1.shape[0] == H.shape[0]:
            return H, self.expectation_conditional_matrix_variate_dist(x2).detach()
        if x2.shape[0] == H.shape[0]:
            return self.expectation_conditional_matrix_variate_dist(x1).detach(), H
        raise ValueError('neither input has as many rows as the latent H')

    def forward(self, x1, x2, diag=False, **params):
        H1, H2 = self._latent(x1, x2)
        s1 = _sigma(H1, self.D)
        s2 = s1 if H2 is H1 else _sigma(H2, self.D)
        self.sigma_matrix_i, self.sigma_matrix_j = s1, s2
        K = ops.ps2d_kernel(x1, x2, s1, s2, jitter)
        return torch.diagonal(K, dim1=-1, dim2=-2) if diag else K
