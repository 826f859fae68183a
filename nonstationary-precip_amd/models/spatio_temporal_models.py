"""Spatio-temporal additive GPs behind the reference's class surface (models/spatio_temporal_models.py:17-126,
SURVEY 8f.1): a temporal component  ScaleKernel(RBF(t) * Periodic(t), outputscale > 7)  on column 0 plus a
spatial component on columns (1, 2) -- stationary RBF-ARD (exact, or SGPR when inducing points are given),
or the sparse non-stationary Gibbs kernel with MAP lengthscales at the inducing points.

The temporal factor is ONE launch of the fused RBF x Periodic gfx950 kernel (nsgp.ops.rbf_periodic_kernel);
Cholesky / triangular work runs on nsgp.ops like the rest of the path."""
import torch

import nsgp.gp as gpytorch
from nsgp import ops
from nsgp.gp.constraints import GreaterThan
from nsgp.gp.kernels import InducingPointKernel, PeriodicKernel, RBFKernel, ScaleKernel
from models.gibbs_kernels import GibbsKernel, GibbsSafeScaleKernel, InducingGibbsKernelST


def _temporal_kernel():
    return ScaleKernel(RBFKernel(active_dims=(0)) * PeriodicKernel(active_dims=(0)),
                       outputscale_constraint=GreaterThan(7), active_dims=0)


class SpatioTemporal_Stationary(gpytorch.models.ExactGP):
    """k = os_t RBF(t) Periodic(t) + os_s RBF-ARD(lon, lat); SGPR over the sum when `z` is given."""

    def __init__(self, train_x, train_y, likelihood, z=None):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gpytorch.means.ZeroMean()
        self.temporal_covar_module = _temporal_kernel()
        self.spatial_covar_module = ScaleKernel(RBFKernel(active_dims=(1, 2)), active_dims=(1, 2))
        if z is not None:
            self.covar_module = InducingPointKernel(base_kernel=self.temporal_covar_module + self.spatial_covar_module,
                                                    inducing_points=z, likelihood=likelihood)
        else:
            self.covar_module = self.temporal_covar_module + self.spatial_covar_module

    def forward(self, x):
        return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))


class SparseSpatioTemporal_Nonstationary(gpytorch.models.ExactGP):
    """MAP inference of the sparse Gibbs-kernel GP over (lon, lat) plus an SGPR temporal component that shares
    the spatial kernel's inducing points (frozen for the temporal part, spatio_temporal_models.py:42-43)."""

    def __init__(self, train_x, train_y, likelihood, prior, z, num_dim=1):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gpytorch.means.ZeroMean()
        self.spatial_covar_module = GibbsSafeScaleKernel(
            InducingGibbsKernelST(GibbsKernel(lengthscale_prior=prior, active_dims=(0, 1)), inducing_points=z,
                                  likelihood=likelihood, active_dims=(1, 2)), active_dims=(1, 2))
        self.temporal_covar_module = InducingPointKernel(
            _temporal_kernel(), inducing_points=self.spatial_covar_module.base_kernel.inducing_points,
            likelihood=likelihood, active_dims=(0))
        self.temporal_covar_module.inducing_points.requires_grad = False
        # gpytorch wraps the very same storage (no copy): the temporal part follows the spatial inducing points
        self.temporal_covar_module.inducing_points.data = self.spatial_covar_module.base_kernel.inducing_points.data
        self.covar_module = self.spatial_covar_module + self.temporal_covar_module
        gk = self.spatial_covar_module.base_kernel.base_kernel
        self.register_parameter('log_ell_z', torch.nn.Parameter(gk.lengthscale_prior.forward(z).mean.clone()))
        self.register_prior('ell_z_prior', gk.lengthscale_prior,
                            lambda module: (module.spatial_covar_module.base_kernel.inducing_points,
                                            module.log_ell_z))

    def forward(self, x, ell=None):
        covar = self.temporal_covar_module(x) + self.spatial_covar_module(x, ell=torch.exp(self.log_ell_z))
        return gpytorch.distributions.MultivariateNormal(self.mean_module(x), covar)

    def predict(self, x_new):
        """SGPR-style predictive at x_new from the joint [train; test] covariance (marginals only are
        meaningful, as the reference's docstring warns): with the dense joint covariance C,
        A^T = C[:n, :] / sigma, L = C[n:, :], B = I + A A^T, mean = L B^-1 A y / sigma,
        cov = C[n:, n:] - L (I - B^-1) L^T   (spatio_temporal_models.py:101-126, dense branch)."""
        x_tr, y_tr = self.train_inputs[0], self.train_targets
        if x_new.ndimension() == 1:
            x_new = x_new.unsqueeze(-1)
        ntr = x_tr.shape[-2]
        full_output = self.forward(torch.cat([x_tr, x_new], dim=-2))
        C = gpytorch.lazy.delazify(full_output.lazy_covariance_matrix)
        sigma = torch.sqrt(self.likelihood.noise)
        L = C[..., ntr:, :].contiguous()
        At = (C[..., :ntr, :] / sigma).contiguous()
        m = At.shape[-1]
        eye = torch.eye(m, dtype=At.dtype, device=At.device)
        B = eye + ops.matmul(At, At, True, False)
        Wb, _ = ops.chol_inv(B)                                       # B^-1 = Wb^T Wb
        v = ops.matmul(At, y_tr.unsqueeze(-1), True, False)
        Binv_v = ops.matmul(Wb, ops.matmul(Wb, v, a_lower=True), True, False, a_lower=True)
        mean = ops.matmul(L, Binv_v).squeeze(-1) / sigma + full_output.loc[ntr:]
        LW = ops.matmul(L, Wb, False, True, b_lower=True)             # L Wb^T
        cov = C[..., ntr:, ntr:] - (ops.matmul(L, L, False, True) - ops.matmul(LW, LW, False, True))
        return full_output.__class__(mean, cov)
