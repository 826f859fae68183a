"""Kernels [gpytorch.kernels semantics recalled, SURVEY A.2]: Kernel base (ard_num_dims, batch_shape,
active_dims, softplus-constrained raw_lengthscale of shape (*batch, 1, D); unknown keyword arguments
such as `lengthscale=` are swallowed exactly like gpytorch's Kernel.__init__(**kwargs) does),
RBFKernel, ScaleKernel, InducingPointKernel, plus sum / product composition.

Every matrix build runs on the gfx950 pairwise kernels (nsgp.ops.rbf_kernel / gibbs_kernel), with
ScaleKernel's outputscale folded into the same launch."""
import math

import torch

from .. import ops
from . import settings
from .constraints import Positive
from .lazy import (LazyEvaluatedKernelTensor, LazyTensor, delazify, lazify, LowRankRootLazyTensor,
                   LowRankRootAddedDiagLazyTensor, DiagLazyTensor, MatmulLazyTensor)
from .module import Module


def same_points(x1, x2):
    """`torch.equal(x1, x2)` of the reference (models/gibbs_kernels.py:148) without a device sync when
    the two arguments are literally the same storage (the training path)."""
    if x1 is x2:
        return True
    if x1.shape != x2.shape:
        return False
    if x1.data_ptr() == x2.data_ptr() and x1.stride() == x2.stride():
        return True
    return bool(torch.equal(x1, x2))


class Kernel(Module):
    has_lengthscale = False
    is_stationary = False

    def __init__(self, ard_num_dims=None, batch_shape=torch.Size([]), active_dims=None, lengthscale_prior=None,
                 lengthscale_constraint=None, eps=1e-6, **kwargs):
        super().__init__()
        self._batch_shape = torch.Size(batch_shape)
        if active_dims is not None and not torch.is_tensor(active_dims):
            active_dims = torch.tensor([active_dims] if isinstance(active_dims, int) else list(active_dims),
                                       dtype=torch.long)
        self.register_buffer('active_dims', active_dims)
        self.ard_num_dims = ard_num_dims
        self.eps = eps
        if self.has_lengthscale:
            nd = 1 if ard_num_dims is None else ard_num_dims
            self.register_parameter('raw_lengthscale', torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, nd)))
            self.register_constraint('raw_lengthscale', lengthscale_constraint or Positive())
            if lengthscale_prior is not None:
                self.register_prior('lengthscale_prior', lengthscale_prior, lambda m: m.lengthscale)

    @property
    def batch_shape(self):
        kernels = list(self.sub_kernels())
        if len(kernels):
            return torch.broadcast_shapes(self._batch_shape, *[k.batch_shape for k in kernels])
        return self._batch_shape

    @batch_shape.setter
    def batch_shape(self, val):
        self._batch_shape = val

    @property
    def lengthscale(self):
        return self._get_constrained('raw_lengthscale') if self.has_lengthscale else None

    @lengthscale.setter
    def lengthscale(self, value):
        self._set_constrained('raw_lengthscale', value)

    def sub_kernels(self):
        for _, m in self.named_children():
            if isinstance(m, Kernel):
                yield m

    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, **params):
        raise NotImplementedError

    def __call__(self, x1, x2=None, diag=False, last_dim_is_batch=False, **params):
        x1_, x2_ = x1, x2
        if self.active_dims is not None:
            x1_ = x1_.index_select(-1, self.active_dims)
            if x2_ is not None:
                x2_ = x2_.index_select(-1, self.active_dims)
        if x1_.dim() == 1:
            x1_ = x1_.unsqueeze(1)
        if x2_ is not None:
            if x2_.dim() == 1:
                x2_ = x2_.unsqueeze(1)
            if x1_.size(-1) != x2_.size(-1):
                raise RuntimeError('x1_ and x2_ must have the same number of dimensions!')
        if x2_ is None:
            x2_ = x1_
        if diag:
            res = self.forward(x1_, x2_, diag=True, **params)
            if not isinstance(res, LazyEvaluatedKernelTensor):
                res = delazify(res)
                if res.dim() == x1_.dim() and res.shape[-2:] == torch.Size((x1_.size(-2), x2_.size(-2))):
                    res = torch.diagonal(res, dim1=-1, dim2=-2)     # the kernel ate the diag option
            return res
        return LazyEvaluatedKernelTensor(x1_, x2_, kernel=self, **params)

    def __add__(self, other):
        return AdditiveKernel(*(list(self.kernels) if isinstance(self, AdditiveKernel) else [self]),
                              *(list(other.kernels) if isinstance(other, AdditiveKernel) else [other]))

    def __mul__(self, other):
        return ProductKernel(*(list(self.kernels) if isinstance(self, ProductKernel) else [self]),
                             *(list(other.kernels) if isinstance(other, ProductKernel) else [other]))


def _batched_inputs(x, B):
    """Bring x to what the batched pairwise kernels accept: (n,D) shared or (B,n,D)."""
    if x.dim() == 2:
        return x
    if x.dim() == 3 and x.shape[0] == B:
        return x
    if x.dim() == 3 and x.shape[0] == 1:
        return x[0]
    raise ops.BackendError(f'kernel inputs of shape {tuple(x.shape)} do not match kernel batch {B}')


class RBFKernel(Kernel):
    """exp(-1/2 |(x1 - x2)/lengthscale|^2)  (models/dgps.py:44-46, models/gibbs_kernels.py:67-69)."""
    has_lengthscale = True
    is_stationary = True

    def _flat(self, x1, x2):
        ls = self.lengthscale                                  # (*batch, 1, D)
        bshape = torch.broadcast_shapes(ls.shape[:-2], x1.shape[:-2], x2.shape[:-2])
        B = bshape.numel() if len(bshape) else 1
        D = x1.shape[-1]
        ls2 = ls.expand(*bshape, 1, ls.shape[-1]).reshape(B, ls.shape[-1])
        if ls2.shape[-1] != D:
            ls2 = ls2.expand(B, D)
        return bshape, B, ls2.contiguous()

    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, _outputscale=None, **params):
        bshape, B, ls = self._flat(x1, x2)
        if _outputscale is None:
            os_ = torch.ones(B, dtype=x1.dtype, device=x1.device)
        else:
            os_ = _outputscale.expand(bshape).reshape(B) if _outputscale.dim() else _outputscale.expand(B)
        if diag:
            if same_points(x1, x2):
                return os_.reshape(*bshape, 1).expand(*bshape, x1.shape[-2]) if len(bshape) else \
                    os_.expand(x1.shape[-2])
            d = ((x1 - x2) / self.lengthscale).pow(2).sum(-1)
            return torch.exp(-0.5 * d) * (os_.reshape(*bshape, 1) if len(bshape) else os_)
        xa = x1.reshape(-1, *x1.shape[-2:]) if x1.dim() > 3 else x1
        xb = x2.reshape(-1, *x2.shape[-2:]) if x2.dim() > 3 else x2
        K = ops.rbf_kernel(_batched_inputs(xa, B), _batched_inputs(xb, B), ls, os_.contiguous())
        return K.reshape(*bshape, K.shape[-2], K.shape[-1])


class ScaleKernel(Kernel):
    """outputscale * base_kernel  (raw_outputscale of shape batch_shape, softplus, init 0)."""

    def __init__(self, base_kernel, outputscale_prior=None, outputscale_constraint=None, **kwargs):
        if base_kernel.active_dims is not None:
            kwargs['active_dims'] = base_kernel.active_dims
        super().__init__(**kwargs)
        self.base_kernel = base_kernel
        bshape = self._batch_shape
        self.register_parameter('raw_outputscale', torch.nn.Parameter(torch.zeros(*bshape) if len(bshape)
                                                                      else torch.tensor(0.)))
        self.register_constraint('raw_outputscale', outputscale_constraint or Positive())
        if outputscale_prior is not None:
            self.register_prior('outputscale_prior', outputscale_prior, lambda m: m.outputscale)

    @property
    def fuses_diag_add(self):
        return getattr(self.base_kernel, 'fuses_outputscale', False) and \
            getattr(self.base_kernel, 'fuses_diag_add', False)

    @property
    def outputscale(self):
        return self._get_constrained('raw_outputscale')

    @outputscale.setter
    def outputscale(self, value):
        self._set_constrained('raw_outputscale', value)

    def forward(self, x1, x2, last_dim_is_batch=False, diag=False, **params):
        os_ = self.outputscale
        if getattr(self.base_kernel, 'fuses_outputscale', False) or isinstance(self.base_kernel, RBFKernel):
            return self.base_kernel.forward(x1, x2, diag=diag, _outputscale=os_, **params)
        orig = self.base_kernel.forward(x1, x2, diag=diag, **params)
        if diag:
            return delazify(orig) * (os_.unsqueeze(-1) if os_.dim() else os_)
        osv = os_.view(*os_.shape, 1, 1) if os_.dim() else os_
        if isinstance(orig, LazyTensor):
            return orig.mul(osv)
        return orig * osv


class AdditiveKernel(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    def forward(self, x1, x2, diag=False, **params):
        res = None
        for k in self.kernels:
            nxt = k(x1, x2, diag=diag, **params)
            nxt = nxt if diag else nxt.evaluate_kernel()
            res = nxt if res is None else res + nxt
        return res


def _same_dims(a, b):
    if a is None or b is None:
        return a is None and b is None
    return a.shape == b.shape and bool(torch.equal(a.cpu(), b.cpu()))


class ProductKernel(Kernel):
    def __init__(self, *kernels):
        super().__init__()
        self.kernels = torch.nn.ModuleList(kernels)

    def _rbf_periodic(self):
        """(rbf, periodic) when this is RBFKernel * PeriodicKernel on the same active dims: one fused launch."""
        if len(self.kernels) != 2:
            return None
        a, b = self.kernels
        if isinstance(a, PeriodicKernel) and isinstance(b, RBFKernel):
            a, b = b, a
        if isinstance(a, RBFKernel) and isinstance(b, PeriodicKernel) and _same_dims(a.active_dims, b.active_dims):
            return a, b
        return None

    @property
    def fuses_outputscale(self):
        return self._rbf_periodic() is not None

    def forward(self, x1, x2, diag=False, _outputscale=None, **params):
        pair = self._rbf_periodic()
        if pair is not None and not diag:
            rbf, per = pair
            if rbf.active_dims is not None:
                x1, x2 = x1.index_select(-1, rbf.active_dims), x2.index_select(-1, rbf.active_dims)
            return per.forward(x1, x2, _outputscale=_outputscale, _rbf=rbf)
        res = None
        for k in self.kernels:
            nxt = delazify(k(x1, x2, diag=diag, **params))
            res = nxt if res is None else res * nxt
        if _outputscale is not None:
            os_ = _outputscale
            res = res * (os_.unsqueeze(-1) if (diag and os_.dim()) else
                         (os_.view(*os_.shape, 1, 1) if os_.dim() else os_))
        return res


class PeriodicKernel(Kernel):
    """exp(-2 sin^2(pi |x1 - x2| / period_length) / lengthscale)  [gpytorch < 1.9 as recalled, SURVEY A.2/A.7:
    Euclidean distance of x / period, division by the lengthscale (not its square)]; used as
    RBFKernel * PeriodicKernel on the time column (models/spatio_temporal_models.py:22,42,
    experiments/temporal_exp.py:39).  Built by the fused RBF x Periodic gfx950 kernel (nsgp.ops.rbf_periodic_kernel)."""
    has_lengthscale = True
    is_stationary = True

    def __init__(self, period_length_prior=None, period_length_constraint=None, **kwargs):
        super().__init__(**kwargs)
        self.register_parameter('raw_period_length', torch.nn.Parameter(torch.zeros(*self._batch_shape, 1, 1)))
        self.register_constraint('raw_period_length', period_length_constraint or Positive())
        if period_length_prior is not None:
            self.register_prior('period_length_prior', period_length_prior, lambda m: m.period_length)

    @property
    def period_length(self):
        return self._get_constrained('raw_period_length')

    @period_length.setter
    def period_length(self, value):
        self._set_constrained('raw_period_length', value)

    def forward(self, x1, x2, diag=False, last_dim_is_batch=False, _outputscale=None, _rbf=None, **params):
        ls, per = self.lengthscale, self.period_length                         # (*batch, 1, 1)
        shapes = [ls.shape[:-2], per.shape[:-2], x1.shape[:-2], x2.shape[:-2]]
        if _rbf is not None:
            shapes.append(_rbf.lengthscale.shape[:-2])
        bshape = torch.broadcast_shapes(*shapes)
        B = bshape.numel() if len(bshape) else 1
        D = x1.shape[-1]
        flat = lambda t: t.expand(*bshape, 1, 1).reshape(B).contiguous()
        ls_b, per_b = flat(ls), flat(per)
        ls_rbf = None
        if _rbf is not None:
            lr = _rbf.lengthscale
            ls_rbf = lr.expand(*bshape, 1, lr.shape[-1]).reshape(B, lr.shape[-1])
            if ls_rbf.shape[-1] != D:
                ls_rbf = ls_rbf.expand(B, D)
            ls_rbf = ls_rbf.contiguous()
        os_ = None
        if _outputscale is not None:
            os_ = (_outputscale.expand(bshape).reshape(B) if _outputscale.dim() else _outputscale.expand(B)).contiguous()
        if diag:
            d = x1 - x2
            r = d.pow(2).sum(-1).sqrt()
            res = torch.exp(-2.0 * torch.sin(math.pi * r / per.squeeze(-1)).pow(2) / ls.squeeze(-1))
            if _rbf is not None:
                res = res * torch.exp(-0.5 * (d / _rbf.lengthscale).pow(2).sum(-1))
            if _outputscale is not None:
                res = res * (_outputscale.unsqueeze(-1) if _outputscale.dim() else _outputscale)
            return res
        xa = x1.reshape(-1, *x1.shape[-2:]) if x1.dim() > 3 else x1
        xb = x2.reshape(-1, *x2.shape[-2:]) if x2.dim() > 3 else x2
        K = ops.rbf_periodic_kernel(_batched_inputs(xa, B), _batched_inputs(xb, B), ls_rbf, ls_b, per_b, os_)
        return K.reshape(*bshape, K.shape[-2], K.shape[-1])


class MaternKernel(Kernel):
    """Declared for import compatibility (experiments/seard_spatial_benchmark.py:15); never exercised."""
    has_lengthscale = True

    def __init__(self, nu=2.5, **kwargs):
        super().__init__(**kwargs)
        self.nu = nu

    def forward(self, x1, x2, diag=False, **params):
        raise NotImplementedError('MaternKernel is imported but not exercised by the reference hot path')


class InducingPointKernel(Kernel):
    """gpytorch.kernels.InducingPointKernel: `inducing_points` is a learnable Parameter; generic SGPR low-rank
    covariance for stationary base kernels here, the Gibbs-kernel variants (lengthscales conditioned on the
    inducing points) in the subclasses (models/gibbs_kernels.py:171-363)."""

    def __init__(self, base_kernel, inducing_points, likelihood, active_dims=None):
        super().__init__(active_dims=active_dims)
        self.base_kernel = base_kernel
        self.likelihood = likelihood
        if inducing_points.dim() == 1:
            inducing_points = inducing_points.unsqueeze(-1)
        self.register_parameter('inducing_points', torch.nn.Parameter(inducing_points.clone()))
        self.register_added_loss_term('inducing_point_loss_term')

    def _clear_cache(self):
        for k in ('_cached_kernel_mat', '_cached_kernel_inv_root'):
            if hasattr(self, k):
                delattr(self, k)

    def train(self, mode=True):
        self._clear_cache()
        return super().train(mode)

    # ---- generic SGPR arithmetic (stationary base kernels; gpytorch InducingPointKernel) -------
    def _inducing_mat(self):
        if not self.training and hasattr(self, '_cached_kernel_mat'):
            return self._cached_kernel_mat
        res = delazify(self.base_kernel(self.inducing_points, self.inducing_points))
        if not self.training:
            self._cached_kernel_mat = res
        return res

    def _inducing_inv_root(self):
        """R with R R^T = Kzz^-1 (= triangular_solve(I, chol_upper(Kzz)) = (L^-1)^T)."""
        if not self.training and hasattr(self, '_cached_kernel_inv_root'):
            return self._cached_kernel_inv_root
        from .utils.cholesky import chol_inv_safe
        res = chol_inv_safe(self._inducing_mat()).transpose(-1, -2)
        if not self.training:
            self._cached_kernel_inv_root = res
        return res

    def _get_covariance(self, x1, x2):
        k_ux1 = delazify(self.base_kernel(x1, self.inducing_points))
        R = self._inducing_inv_root()
        root1 = ops.matmul(k_ux1, R, b_lower=False)
        if same_points(x1, x2):
            covar = LowRankRootLazyTensor(root1)
            if not self.training and settings.sgpr_diagonal_correction.on():
                correction = (self.base_kernel(x1, x2, diag=True) - covar.diag()).clamp(0, math.inf)
                covar = LowRankRootAddedDiagLazyTensor(covar, DiagLazyTensor(correction))
            return covar
        k_ux2 = delazify(self.base_kernel(x2, self.inducing_points))
        return MatmulLazyTensor(root1, ops.matmul(k_ux2, R, b_lower=False).transpose(-1, -2))

    def _covar_diag(self, inputs):
        if inputs.ndimension() == 1:
            inputs = inputs.unsqueeze(1)
        return DiagLazyTensor(self.base_kernel(inputs, diag=True))

    def forward(self, x1, x2, diag=False, **kwargs):
        covar = self._get_covariance(x1, x2)
        if self.training:
            if not same_points(x1, x2):
                raise RuntimeError('x1 should equal x2 in training mode')
            from .distributions import MultivariateNormal
            from .mlls import InducingPointKernelAddedLossTerm
            zero_mean = torch.zeros_like(x1.select(-1, 0))
            term = InducingPointKernelAddedLossTerm(MultivariateNormal(zero_mean, self._covar_diag(x1)),
                                                    MultivariateNormal(zero_mean, covar), self.likelihood)
            self.update_added_loss_term('inducing_point_loss_term', term)
        return covar.diag() if diag else covar
