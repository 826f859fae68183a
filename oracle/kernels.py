"""Kernel-matrix builds, restated on CPU (oracle; test infrastructure only).

* gibbs            -- models/gibbs_kernels.py:154-162 (op sequence kept, incl. the 8 temporaries)
* gibbs_scalar     -- the same formula entry by entry (R&W eq. 4.32), used to pin `gibbs`
* rbf_ard          -- gpytorch RBFKernel(ard)+ScaleKernel as built at models/dgps.py:44-46 and
                      models/gibbs_kernels.py:67-69  [gpytorch semantics recalled; SURVEY A.2]
* periodic         -- gpytorch PeriodicKernel as used by models/spatio_temporal_models.py:23
                      [recalled, gpytorch<1.9: exp(-2 sum sin^2(pi d/p) / ell)]
* ps2d             -- models/multivariate_gibbs_kernel.py:98-150 (Paciorek-Schervish, D=2)
"""
import math
import torch
from . import functional as fn


def gibbs(x1, x2, ell1, ell2):
    """Diagonal Gibbs kernel, exactly the reference op sequence.

    x1:(n1,D) x2:(n2,D) row-major; ell1:(D,n1) ell2:(D,n2) dim-major -> (n1,n2).
    models/gibbs_kernels.py:154-162.
    """
    sq_sum = ell1.unsqueeze(-1) ** 2 + ell2.unsqueeze(-2) ** 2            # :154
    out = torch.sqrt(2 * fn.op(ell1, ell2) / sq_sum)                        # :155
    out = torch.prod(out, dim=-3)                                           # :156
    diff = x1.unsqueeze(-2) - x2.unsqueeze(-3)                              # :157
    nd = sq_sum.dim()
    out = out * torch.exp(-torch.sum(
        diff ** 2 / sq_sum.permute(*range(nd - 3), -2, -1, -3), dim=-1))    # :158-161
    return out


def gibbs_scalar(x1, x2, ell1, ell2, i, j):
    """One entry of the Gibbs kernel from the textbook formula (R&W 4.32); pins `gibbs`."""
    D = x1.shape[-1]
    pre, ex = 1.0, 0.0
    for d in range(D):
        a, b = float(ell1[d, i]), float(ell2[d, j])
        s = a * a + b * b
        pre *= math.sqrt(2 * a * b / s)
        ex += (float(x1[i, d]) - float(x2[j, d])) ** 2 / s
    return pre * math.exp(-ex)


def rbf_ard(x1, x2, lengthscale, outputscale=None):
    """outputscale * exp(-0.5 * sum_d ((x1_d - x2_d)/ell_d)^2)   [gpytorch RBFKernel, recalled].

    x1:(...,n1,D) x2:(...,n2,D); lengthscale broadcastable to (...,1,D); outputscale (...,) or None.
    """
    a = x1 / lengthscale
    b = x2 / lengthscale
    d2 = (a.unsqueeze(-2) - b.unsqueeze(-3)).pow(2).sum(-1)
    k = torch.exp(-0.5 * d2)
    if outputscale is not None:
        os_ = torch.as_tensor(outputscale, dtype=k.dtype)
        k = k * os_.reshape(*os_.shape, 1, 1) if os_.dim() else k * os_
    return k


def periodic(x1, x2, lengthscale, period):
    """gpytorch PeriodicKernel, pre-ARD form (< 1.6, recalled; the reference pins no version):
    exp(-2 sin^2(pi |x/p - x'/p|_2) / ell) -- Euclidean distance of the period-scaled inputs, division by the
    lengthscale (not its square).  The reference only uses it on one column (active_dims=0:
    models/spatio_temporal_models.py:22,42, experiments/temporal_exp.py:39), where the later per-dimension
    form coincides with this one up to the ell vs ell^2 convention.  Parity unpinned."""
    diff = (x1 / period).unsqueeze(-2) - (x2 / period).unsqueeze(-3)      # period: scalar or (..., 1, 1)
    r2 = diff.pow(2).sum(-1)
    pos = r2 > 0                                       # sqrt has an infinite slope at 0; sin^2 has zero slope there
    r = torch.where(pos, torch.where(pos, r2, torch.ones_like(r2)).sqrt(), torch.zeros_like(r2))
    return torch.exp(-2.0 * torch.sin(math.pi * r).pow(2) / lengthscale)


def softplus(x):
    return torch.nn.functional.softplus(x)


def ps_sigma(H, Dmat):
    """Sigma_i = softplus((h_i h_i^T)**2) + D**2 (elementwise squares), (N,2,2).

    models/multivariate_gibbs_kernel.py:98 (python loop over rows through numpy, vectorised here).
    """
    outer = H.unsqueeze(-1) * H.unsqueeze(-2)
    return softplus(outer ** 2) + Dmat ** 2


def ps2d(x1, x2, sig1, sig2, jitter=1e-5):
    """Paciorek-Schervish kernel for D=2 with per-point 2x2 matrices sig1:(n1,2,2), sig2:(n2,2,2).

    |S_i|^{1/4} |S_j|^{1/4} |(S_i+S_j)/2|^{-1/2} exp(-d^T ((S_i+S_j)/2 + jitter I)^{-1} d)
    models/multivariate_gibbs_kernel.py:104-150 (note: no 1/2 in the exponent, jitter only in the
    inverse, not in the determinant).
    """
    n1, n2 = x1.shape[0], x2.shape[0]
    Si = sig1.unsqueeze(1).expand(n1, n2, 2, 2)
    Sj = sig2.unsqueeze(0).expand(n1, n2, 2, 2)
    det_i = torch.det(sig1).pow(0.25)
    det_j = torch.det(sig2).pow(0.25)
    det_product = det_i.unsqueeze(1) * det_j.unsqueeze(0)                     # :105-106 / :135-137
    avg = (Si + Sj) / 2                                                       # :139
    prefactor = det_product * torch.det(avg).pow(-0.5)                        # :140-141
    inv = torch.inverse(avg + jitter * torch.eye(2, dtype=avg.dtype))         # :143
    diff = x1.unsqueeze(-2) - x2.unsqueeze(-3)                                # :144
    q = (diff.unsqueeze(-2) @ inv @ diff.unsqueeze(-1)).reshape(n1, n2)       # :145-146
    return prefactor * torch.exp(-q)                                          # :148
