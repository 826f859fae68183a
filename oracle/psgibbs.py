"""Multivariate (Paciorek-Schervish) Gibbs kernels restated on CPU (oracle; test infrastructure only).

Follows models/latent_priors.py:27-64 (MatrixVariateNormalPrior),
models/multivariate_gibbs_kernel.py:20-150 and models/sparse_multivariate_gibbs_kernel.py:20-154.
The gpytorch row kernel (RBF-ARD, optionally under a ScaleKernel) is passed in as its
constrained lengthscale / outputscale [gpytorch semantics recalled, SURVEY A.2].
"""
import math
import torch
from . import kernels

JITTER = 1e-5       # models/latent_priors.py:14, models/multivariate_gibbs_kernel.py:17


class MatrixNormalPrior:
    """models/latent_priors.py:27-64 -- note the two vec orders are inconsistent and preserved:
    covariance kron(row+jI, col) is row-major vec (:45), log_prob flattens x.T (column-stacking, :64),
    kron_cov_inv = kron(col^-1, (row+jI)^-1) is column-stacking (:46)."""

    def __init__(self, loc, row_cov, col_cov):
        n, d = row_cov.shape[0], col_cov.shape[0]
        rowj = row_cov + torch.eye(n, dtype=row_cov.dtype) * JITTER
        self.n, self.d = n, d
        self.vec_loc = loc.flatten().double()                                   # :44,48
        self.kron_cov = torch.kron(rowj, col_cov).double()                      # :45
        self.kron_cov_inv = torch.kron(col_cov.inverse(), rowj.inverse())       # :46

    def log_prob(self, x):
        from .exact import mvn_log_prob
        return mvn_log_prob(x.T.flatten().double(), self.vec_loc, self.kron_cov)  # :63-64

    def sample_from_eps(self, eps):
        """sample_n(1) given a standard-normal draw eps:(n*d,) (:59-61): loc + L eps, reshape (n,d)."""
        L = torch.linalg.cholesky(self.kron_cov)
        return (self.vec_loc + L @ eps.double()).reshape(self.n, self.d)


def conditional_H(x_star, x, H, row_ls, col_cov, row_os=1.0, kron_cov_inv=None):
    """expectation_conditional_matrix_variate_dist (multivariate_gibbs_kernel.py:65-75;
    sparse variant :67-82 recomputes the Kronecker inverse from the current row covariance)."""
    n, d = H.shape
    if kron_cov_inv is None:
        row = kernels.rbf_ard(x, x, row_ls, row_os)
        kron_cov_inv = torch.kron(col_cov.inverse(),
                                  (row + torch.eye(n, dtype=row.dtype) * JITTER).inverse())
    row_cross = kernels.rbf_ard(x_star, x, row_ls, row_os)                      # :68 (N*,N)
    cross = torch.kron(col_cov, row_cross)                                      # :71
    vec = cross @ kron_cov_inv @ H.T.flatten()                                  # :73
    return vec.reshape(d, x_star.shape[0]).T                                    # :75


def mv_gibbs_forward(x1, x2, x_train, H, Dmat, row_ls, col_cov, row_os=1.0, kron_cov_inv=None):
    """MultivariateGibbsKernel.forward (multivariate_gibbs_kernel.py:77-150) /
    SparseMultivariateGibbsKernel.forward (sparse_multivariate_gibbs_kernel.py:84-154).

    x_train are the locations H lives at (train inputs, or inducing Z for the sparse variant).
    Branches: x1 == x2 and len == len(H) -> H itself; x1 == x2 otherwise -> conditional mean;
    x1 != x2 -> whichever side has len(H) rows uses H, the other the conditional mean.
    H enters detached (no gradient through K, :85,98).
    """
    Hd = H.detach()

    def cond(xs):
        return conditional_H(xs, x_train, Hd, row_ls, col_cov, row_os, kron_cov_inv).detach()

    if torch.equal(x1, x2):
        Hx = Hd if x1.shape[0] == H.shape[0] else cond(x1)
        s1 = s2 = kernels.ps_sigma(Hx, Dmat)
    else:
        if x1.shape[0] == H.shape[0]:
            H1, H2 = Hd, cond(x2)
        elif x2.shape[0] == H.shape[0]:
            H1, H2 = cond(x1), Hd
        else:
            raise ValueError('neither input matches the latent H (reference leaves this undefined)')
        s1, s2 = kernels.ps_sigma(H1, Dmat), kernels.ps_sigma(H2, Dmat)
    return kernels.ps2d(x1, x2, s1, s2, JITTER)
