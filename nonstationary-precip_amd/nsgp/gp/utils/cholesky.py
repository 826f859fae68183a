"""gpytorch.utils.cholesky.psd_safe_cholesky on the MI355X potrf (models/gibbs_kernels.py:201,298).

Semantics [SURVEY A.6]: factor; on failure retry with jitter 1e-6 * 10^i (float32) / 1e-8 * 10^i
(float64), i < 3, warning each time; raise if still not positive definite."""
import warnings

import torch

from ... import ops
from .. import settings


class NumericalWarning(RuntimeWarning):
    pass


class NanError(RuntimeError):
    pass


class NotPSDError(RuntimeError):
    pass


def _capturing(t):
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


def psd_safe_cholesky(A, upper=False, out=None, jitter=None, max_tries=3):
    L, info = ops.potrf(A)
    if _capturing(A):                                   # hipGraph capture: no host read of `info`, no retry ladder (the
        return L.transpose(-1, -2) if upper else L      # eager warm-up steps before the capture went through the checks)
    if int(info.max().item()) != 0:                     # the reference syncs here too (torch raises)
        if torch.isnan(A).any():
            raise NanError(f'cholesky_cpu: {int(torch.isnan(A).sum())} of {A.numel()} elements are NaN')
        base = jitter if jitter is not None else settings.cholesky_jitter.value(A.dtype)
        n = A.shape[-1]
        eye = torch.eye(n, dtype=A.dtype, device=A.device)
        ok = False
        prev = 0.0
        Ap = A.clone()
        for i in range(max_tries):
            jit = base * (10 ** i)
            Ap = Ap + (jit - prev) * eye
            prev = jit
            L, info = ops.potrf(Ap)
            if int(info.max().item()) == 0:
                warnings.warn(f'A not p.d., added jitter of {jit:.1e} to the diagonal', NumericalWarning)
                ok = True
                break
        if not ok:
            raise NotPSDError(f'Matrix not positive definite after repeatedly adding jitter up to {jit:.1e}.')
    if upper:
        L = L.transpose(-1, -2)
    return L


def chol_inv_safe(K, jitter=None, max_tries=3):
    """W = chol(K)^-1 (lower, differentiable: ops.CholInvFn) with psd_safe_cholesky's retry policy: if the
    factorisation fails, add jitter 1e-6 * 10^i (float32) / 1e-8 * 10^i (float64) to the diagonal, warn, retry.
    The SGPR kernels of the reference rely on this (psd_safe_cholesky at models/gibbs_kernels.py:201,298 and in
    gpytorch's InducingPointKernel): e.g. the temporal component of SparseSpatioTemporal_Nonstationary evaluates
    its kernel on the TIME column of inducing points that share time stamps, a singular Kzz.
    One host sync per call (reading `info`), like the reference's exception-driven retry."""
    W, info = ops.chol_inv(K)
    if _capturing(K) or int(info.max().item()) == 0:     # (capture: see psd_safe_cholesky)
        return W
    if torch.isnan(K).any():
        raise NanError(f'cholesky: {int(torch.isnan(K).sum())} of {K.numel()} elements are NaN')
    base = jitter if jitter is not None else settings.cholesky_jitter.value(K.dtype)
    eye = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
    for i in range(max_tries):
        jit = base * (10 ** i)
        W, info = ops.chol_inv(K + jit * eye)
        if int(info.max().item()) == 0:
            warnings.warn(f'A not p.d., added jitter of {jit:.1e} to the diagonal', NumericalWarning)
            return W
    raise NotPSDError(f'Matrix not positive definite after repeatedly adding jitter up to {jit:.1e}.')
