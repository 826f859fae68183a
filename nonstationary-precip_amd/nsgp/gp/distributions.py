"""MultivariateNormal / MultitaskMultivariateNormal with the surface the reference uses
(.loc/.mean, .covariance_matrix, .lazy_covariance_matrix, .variance, .log_prob, .rsample,
batch/event shapes, `dist.__class__(mean, covar)` at models/nonstationary_models.py:153)
[gpytorch.distributions semantics recalled, SURVEY A.5/A.6].

log_prob runs on the GPU Cholesky (always-Cholesky): dense covariances through
potrf + trtri with the closed-form adjoint  Kbar = g/2 (alpha alpha^T - K^-1); low-rank-plus-diagonal
covariances (SGPR training) through the M x M Woodbury system."""
import math

import torch

from .. import ops
from . import settings
from .lazy import (LazyTensor, NonLazyTensor, lazify, AddedDiagLazyTensor, RootLazyTensor, DiagLazyTensor,
                   LazyEvaluatedKernelTensor)

LOG2PI = math.log(2 * math.pi)


class MvnLogProbFn(torch.autograd.Function):
    """log N(d | 0, K) for dense (batched) K:(...,n,n), d:(...,n) -> (...)."""

    @staticmethod
    def forward(ctx, K, d):
        batch_shape = K.shape[:-2]
        n = K.shape[-1]
        K3 = K.reshape(-1, n, n)
        d3 = d.expand(*batch_shape, n).reshape(-1, n, 1)
        L, info = ops.potrf(K3)
        W = ops.trtri(L)
        a = ops.gemm(W, d3, flags=ops.GEMM_A_LOWER)                       # W d
        quad = (a * a).sum((-1, -2))
        logdet = -2.0 * torch.log(torch.diagonal(W, dim1=-1, dim2=-2)).sum(-1)
        ctx.save_for_backward(W, a)
        ctx.shapes = (K.shape, d.shape, batch_shape)
        ctx.mark_non_differentiable(info)
        return (-0.5 * (quad + logdet + n * LOG2PI)).reshape(batch_shape), info

    @staticmethod
    def backward(ctx, g, _):
        W, a = ctx.saved_tensors
        Kshape, dshape, batch_shape = ctx.shapes
        alpha = ops.gemm(W, a, ta=True, flags=ops.GEMM_A_UPPER)           # K^-1 d
        g3 = g.reshape(-1, 1, 1)
        gK = gd = None
        if ctx.needs_input_grad[0]:
            Kinv = ops.gemm(W, W, ta=True, flags=ops.GEMM_A_UPPER | ops.GEMM_B_LOWER)
            gK = (0.5 * g3) * (alpha * alpha.transpose(-1, -2) - Kinv)      # outer product, elementwise
            gK = gK.reshape(Kshape)
        if ctx.needs_input_grad[1]:
            gd = (-g3 * alpha).squeeze(-1).reshape(*batch_shape, -1)
            if tuple(dshape) != tuple(gd.shape):
                gd = gd.sum_to_size(dshape)
        return gK, gd


def _checked_log_prob(lp, info, K, diff):
    """psd_safe_cholesky's policy for the dense MVN log-density (gpytorch reaches it from MultivariateNormal.log_prob
    -> lazy covariance Cholesky): a failed factorisation is retried with jitter 1e-6 * 10^i (float32) /
    1e-8 * 10^i (float64), i < 3, under a NumericalWarning; then NotPSDError (NanError for NaN input) -- never a silent
    NaN objective.  One host sync (reading `info`); the exact-GP loops sync on float(loss) every iteration anyway."""
    if int(info.max().item()) == 0:
        return lp
    import warnings
    from .utils.cholesky import NanError, NotPSDError, NumericalWarning
    if torch.isnan(K).any():
        raise NanError(f'cholesky: {int(torch.isnan(K).sum())} of {K.numel()} elements are NaN')
    base = settings.cholesky_jitter.value(K.dtype)
    eye = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
    for i in range(3):
        jit = base * (10 ** i)
        lp, info = MvnLogProbFn.apply(K + jit * eye, diff)
        if int(info.max().item()) == 0:
            warnings.warn(f'A not p.d., added jitter of {jit:.1e} to the diagonal', NumericalWarning)
            return lp
    raise NotPSDError(f'Matrix not positive definite after repeatedly adding jitter up to {jit:.1e}.')


def _lowrank_diag_log_prob(root, diag, d):
    """log N(d | 0, R R^T + diag(D)) via the k x k Woodbury system (R:(n,k))."""
    n, k = root.shape[-2], root.shape[-1]
    dinv = 1.0 / diag
    Rs = root * dinv.sqrt().unsqueeze(-1)                                # D^-1/2 R
    B = ops.matmul(Rs, Rs, True, False) + torch.eye(k, dtype=root.dtype, device=root.device)
    W, _ = ops.chol_inv(B)                                               # B^-1 = W^T W
    v = ops.matmul(Rs, (d * dinv.sqrt()).unsqueeze(-1), True, False)    # R^T D^-1 d
    u = ops.matmul(W, v, a_lower=True)
    quad = (d * d * dinv).sum(-1) - (u * u).sum((-1, -2))
    logdet = -2.0 * torch.log(torch.diagonal(W, dim1=-1, dim2=-2)).sum(-1) + torch.log(diag).sum(-1)
    return -0.5 * (quad + logdet + n * LOG2PI)


class MultivariateNormal:
    def __init__(self, mean, covariance_matrix, validate_args=False):
        self.loc = mean
        self._covar = covariance_matrix
        self._islazy = isinstance(covariance_matrix, LazyTensor)

    # -- basic accessors
    @property
    def mean(self):
        return self.loc

    @property
    def lazy_covariance_matrix(self):
        return self._covar if self._islazy else lazify(self._covar)

    @property
    def covariance_matrix(self):
        return self._covar.evaluate() if self._islazy else self._covar

    @property
    def variance(self):
        v = self._covar.diag() if self._islazy else torch.diagonal(self._covar, dim1=-1, dim2=-2)
        return v.expand(self.loc.shape) if v.shape != self.loc.shape else v

    @property
    def stddev(self):
        return self.variance.sqrt()

    @property
    def event_shape(self):
        return self.loc.shape[-1:]

    @property
    def batch_shape(self):
        return torch.broadcast_shapes(self.loc.shape[:-1], tuple(self._covar.shape[:-2]))

    def confidence_region(self):
        s2 = self.stddev * 2
        return self.mean - s2, self.mean + s2

    def expand(self, batch_size):
        batch_size = torch.Size(batch_size)
        mean = self.loc.expand(*batch_size, self.loc.shape[-1])
        cov = self.covariance_matrix
        return self.__class__(mean, cov.expand(*batch_size, *cov.shape[-2:]))

    def __getitem__(self, idx):
        return self.__class__(self.loc[idx], self.covariance_matrix[idx])

    # -- densities / sampling
    def log_prob(self, value):
        diff = value - self.loc
        cov = self._covar
        if isinstance(cov, LazyEvaluatedKernelTensor):
            cov = cov.evaluate_kernel()
        if isinstance(cov, AddedDiagLazyTensor) and isinstance(cov._lazy_tensor, RootLazyTensor) \
                and cov._lazy_tensor.root.shape[-1] < cov.shape[-1] and diff.dim() == 1:
            return _lowrank_diag_log_prob(cov._lazy_tensor.root.evaluate(), cov._diag_tensor.diag(), diff)
        K = cov.evaluate() if isinstance(cov, LazyTensor) else cov
        lp, info = MvnLogProbFn.apply(K, diff)
        if settings.check_mvn_cholesky.on() and not torch.cuda.is_current_stream_capturing():
            lp = _checked_log_prob(lp, info, K, diff)
        return lp

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        from .utils.cholesky import psd_safe_cholesky
        K = self.covariance_matrix
        n = K.shape[-1]
        bshape = self.batch_shape
        L = psd_safe_cholesky(K).expand(*bshape, n, n).reshape(-1, n, n).contiguous()       # (B,n,n)
        sample_shape = torch.Size(sample_shape)
        ns = sample_shape.numel() if len(sample_shape) else 1
        if base_samples is None:
            base_samples = torch.randn(*sample_shape, *bshape, n, dtype=self.loc.dtype, device=self.loc.device)
        z = base_samples.reshape(ns, -1, n).permute(1, 2, 0).contiguous()                   # (B,n,ns)
        out = ops.matmul(L, z, a_lower=True)                                                # L z on MFMA
        out = out.permute(2, 0, 1).reshape(*sample_shape, *bshape, n)
        return out + self.loc

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    def sample_n(self, n):
        return self.sample(torch.Size((n,)))

    def __add__(self, other):
        if isinstance(other, MultivariateNormal):
            return self.__class__(self.loc + other.loc, self.lazy_covariance_matrix + other.lazy_covariance_matrix)
        return self.__class__(self.loc + other, self._covar)


class MultitaskMultivariateNormal(MultivariateNormal):
    """Independent-task form produced by DeepGPLayer (block-diagonal over tasks, non-interleaved):
    mean (..., n, t); only the marginal variances are ever consumed downstream, so the
    node keeps the (t, ns, n) device layout the fused sampling kernel reads."""

    def __init__(self, mean, covariance_matrix=None, interleaved=False, _var=None, _tsn=None):
        self.loc = mean
        self._covar = covariance_matrix
        self._islazy = isinstance(covariance_matrix, LazyTensor)
        self._var = _var
        self._tsn = _tsn               # (mean_tsn, var_tsn, S): internal layout for ops.dgp_sample

    @property
    def variance(self):
        if self._var is not None:
            return self._var
        n, t = self.loc.shape[-2:]
        v = torch.diagonal(self.covariance_matrix, dim1=-1, dim2=-2)
        return v.reshape(*v.shape[:-1], t, n).transpose(-1, -2)

    @property
    def event_shape(self):
        return self.loc.shape[-2:]

    @property
    def batch_shape(self):
        return self.loc.shape[:-2]

    def expand(self, batch_size):
        batch_size = torch.Size(batch_size)
        new = MultitaskMultivariateNormal(self.loc.expand(*batch_size, *self.loc.shape[-2:]), None,
                                          _var=self.variance.expand(*batch_size, *self.loc.shape[-2:]),
                                          _tsn=self._tsn)
        return new

    def rsample(self, sample_shape=torch.Size(), base_samples=None):
        shape = torch.Size(sample_shape) + self.loc.shape
        if base_samples is None:
            base_samples = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        return self.loc + self.variance.sqrt() * base_samples

    def log_prob(self, value):
        raise NotImplementedError('MultitaskMultivariateNormal.log_prob is not on the reference hot path')
