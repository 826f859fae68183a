"""Terse batched linear-algebra helpers with the reference's names (utils/functional.py:14-64 are
the ones its hot path uses: dot, t, mv, op).  Elementwise helpers are plain torch (device agnostic);
`mv(..., invert=True)` -- the reference's torch.linalg.solve at :33, only ever applied to symmetric
positive definite K + noise I -- runs on the MI355X Cholesky + MFMA solves and needs CUDA tensors."""
import math
from collections import namedtuple
from typing import Optional

import torch


def dot(v1, v2):
    """Batch dot product over the last dim."""
    return (v1 * v2).sum(-1)


def t(x):
    """Transpose of the two trailing dims."""
    return torch.transpose(x, -1, -2)


def tr(x):
    """Trace over the two trailing dims."""
    return torch.diagonal(x, dim1=-1, dim2=-2).sum(-1)


def mv(matrix, vector, invert=False):
    """matrix @ vector, or matrix^-1 vector when invert=True.  The reference's torch.linalg.solve (utils/functional.py:33)
    takes any invertible matrix; its hot path only ever passes K + noise I, so this is a Cholesky solve on the GPU and
    a matrix that is not symmetric positive definite raises NotPSDError (after psd_safe_cholesky's jitter retries)
    instead of being LU-factored."""
    from nsgp import ops
    if matrix.is_cuda:
        rhs = vector.unsqueeze(-1)
        if not invert:
            return ops.matmul(matrix, rhs).squeeze(-1)
        from nsgp.gp.utils.cholesky import chol_inv_safe
        W = chol_inv_safe(matrix.contiguous())     # reads `info`: a matrix that is not SPD raises NotPSDError
        return ops.matmul(W, ops.matmul(W, rhs, a_lower=True), True, False, a_lower=True).squeeze(-1)
    if invert:
        raise ops.BackendError('fn.mv(invert=True) runs on the MI355X Cholesky: move the operands to the GPU')
    return (matrix * vector.unsqueeze(-2)).sum(-1)


def quad(v, matrix, v2=None, invert=False):
    v2 = v if v2 is None else v2
    return dot(v, mv(matrix, v2, invert=invert))


def expquad(v, matrix, invert=False, out_scale=1.0, exp_scale=0.5):
    return out_scale * torch.exp(-exp_scale * quad(v, matrix, invert=invert))


def sym(x):
    """Force symmetry."""
    return 0.5 * (x + t(x))


def op(v1, v2: Optional[torch.Tensor] = None):
    """Outer product over the last dim (broadcast multiply; no BLAS call)."""
    if v2 is None:
        v2 = v1
    return v1.unsqueeze(-1) * v2.unsqueeze(-2)


def vec(x):
    """Column-stacking vectorisation."""
    return t(x).contiguous().view(*x.shape[:-2], x.shape[-2] * x.shape[-1])


def vech(x):
    """Half vectorisation (lower triangle)."""
    D = x.shape[-2]
    if x.shape[-1] != D:
        raise ValueError('Matrix must be square for half vectorisation, but got shape {}'.format(x.shape))
    return x[..., torch.tril(torch.ones(D, D, device=x.device)) == 1]


def kron(x, y):
    """Batch Kronecker product."""
    res = x.unsqueeze(-1).unsqueeze(-3) * y.unsqueeze(-2).unsqueeze(-4)
    return res.reshape(*res.shape[:-4], x.shape[-2] * y.shape[-2], x.shape[-1] * y.shape[-1])


def diff(x, boundary_value=None, dim=-2):
    """Forward differences along `dim`, length kept by appending boundary_value (or repeating the last)."""
    x = x.transpose(dim, -1)
    d = x[..., 1:] - x[..., :-1]
    if boundary_value is None:
        boundary_value = d[..., -1]
    return torch.cat((d, boundary_value.unsqueeze(-1)), dim=-1).transpose(dim, -1)


def normalise(x, **kwargs):
    """Zero mean, unit norm (times sqrt(n)) over `dim` (default last)."""
    dim = kwargs.get('dim', -1)
    x = x - torch.mean(x, dim=dim, keepdim=True)
    return math.sqrt(x.shape[dim]) * torch.nn.functional.normalize(x, **kwargs)


def robust_logdet(x, init_scale=1e-30, max_scale=1e-6):
    """logdet with growing diagonal regularisation while the result is NaN."""
    out = torch.logdet(x)
    reg = init_scale
    eye = torch.eye(x.shape[-1], device=x.device, dtype=x.dtype)
    while reg <= max_scale and torch.any(torch.isnan(out)):
        out = torch.logdet(x + reg * eye)
        reg *= 10
    return out
