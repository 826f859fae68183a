"""Latent priors on the MI355X engine -- drop-in for models/latent_priors.py of the reference.

  MatrixVariateNormalPrior(loc, row_covariance_matrix, column_covariance_matrix)   reference :27-64
  LearnedSoftPlus                                                                  reference :16-25
The matrix-normal prior is a MultivariateNormalPrior over vec(X) with Kronecker covariance
kron(row + 1e-5 I, col) (float64).  Quirk kept: the covariance uses row-major
vec order while log_prob flattens x.T (column-stacking) and kron_cov_inv is built in
column-stacking order -- numerically identical to the reference by construction.
"""
import torch

import nsgp.gp as gpytorch

jitter = 1e-5


class LearnedSoftPlus(torch.nn.Module):
    def __init__(self, init_beta=1.0, threshold=20):
        super().__init__()
        self.log_beta = torch.nn.Parameter(torch.tensor(float(init_beta)).log())
        self.threshold = 20

    def forward(self, x):
        beta = self.log_beta.exp()
        bx = beta * x
        return torch.where(bx < 20, torch.log1p(bx.exp()) / beta, x)


class MatrixVariateNormalPrior(gpytorch.priors.MultivariateNormalPrior):
    """Matrix normal prior for an N x D real matrix: rows ~ row_covariance (N x N), columns ~
    column_covariance (D x D)."""

    def __init__(self, loc, row_covariance_matrix, column_covariance_matrix):
        n, d = row_covariance_matrix.shape[0], column_covariance_matrix.shape[0]
        eye = torch.eye(n, dtype=row_covariance_matrix.dtype, device=row_covariance_matrix.device)
        row_j = row_covariance_matrix + eye * jitter
        vec_loc = loc.flatten()
        kron_cov = torch.kron(row_j, column_covariance_matrix)
        super().__init__(loc=vec_loc.double(), covariance_matrix=kron_cov.double())
        self.row_covariance_matrix = row_covariance_matrix
        self.col_covariance_matrix = column_covariance_matrix
        self.vec_loc = vec_loc
        self.kron_cov = kron_cov
        self._row_j = row_j
        self._kron_cov_inv = None
        self.n, self.d = n, d

    @property
    def kron_cov_inv(self):
        """kron(col^-1, (row + jitter I)^-1): the (N x N) inverse runs on the GPU Cholesky in float64."""
        if self._kron_cov_inv is None:
            from nsgp import ops
            W, _ = ops.chol_inv(self._row_j.double().contiguous())
            row_inv = ops.gemm(W, W, ta=True, flags=ops.GEMM_A_UPPER | ops.GEMM_B_LOWER).to(self._row_j.dtype)
            self._row_inv = row_inv
            col_inv = torch.linalg.inv(self.col_covariance_matrix.cpu()).contiguous().to(row_inv.device)
            self._kron_cov_inv = torch.kron(col_inv, row_inv.contiguous())
        return self._kron_cov_inv

    @property
    def row_inv(self):
        _ = self.kron_cov_inv
        return self._row_inv

    def sample_n(self, num_samples):
        vec_sample = super().sample_n(num_samples).T
        return vec_sample.reshape(self.n, self.d)

    def log_prob(self, x):
        return super().log_prob(x.T.flatten())
