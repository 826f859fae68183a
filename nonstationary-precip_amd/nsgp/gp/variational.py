"""Whitened variational GP approximation [gpytorch.variational recalled, SURVEY A.3]:
CholeskyVariationalDistribution (variational_mean (*batch,M) init 0 (+N(0,1e-3^2) on first use),
chol_variational_covar (*batch,M,M) init I) and VariationalStrategy (learnable inducing_points),
with state-dict keys `variational_strategy.inducing_points`,
`variational_strategy._variational_distribution.{variational_mean,chol_variational_covar}`.

The marginals are ONE fused autograd node on the MI355X (nsgp.svgp.SVGPLayerFn): Kzz / Cholesky /
inverse once per call in float64, then MFMA GEMMs -- no per-sample recomputation."""
import torch

from .. import ops
from ..svgp import svgp_marginal, whiten, VAR_JITTER
from . import settings
from .distributions import MultivariateNormal
from .kernels import RBFKernel, ScaleKernel
from .lazy import CholLazyTensor
from .means import ConstantMean, LinearMean, ZeroMean
from .module import Module


class _VariationalDistribution(Module):
    def __init__(self, num_inducing_points, batch_shape=torch.Size(), mean_init_std=1e-3):
        super().__init__()
        self.num_inducing_points = num_inducing_points
        self.batch_shape = torch.Size(batch_shape)
        self.mean_init_std = mean_init_std


class CholeskyVariationalDistribution(_VariationalDistribution):
    def __init__(self, num_inducing_points, batch_shape=torch.Size(), mean_init_std=1e-3, **kwargs):
        super().__init__(num_inducing_points, batch_shape, mean_init_std)
        M = num_inducing_points
        self.register_parameter('variational_mean', torch.nn.Parameter(torch.zeros(*self.batch_shape, M)))
        self.register_parameter('chol_variational_covar',
                                torch.nn.Parameter(torch.eye(M).repeat(*self.batch_shape, 1, 1)))

    def forward(self):
        L = torch.tril(self.chol_variational_covar)
        return MultivariateNormal(self.variational_mean, CholLazyTensor(L))

    def initialize_variational_distribution(self, prior_dist=None):
        """Whitened prior is N(0, I): mean <- 0 + N(0, mean_init_std^2), chol <- I."""
        with torch.no_grad():
            self.variational_mean.zero_()
            self.variational_mean.add_(torch.randn_like(self.variational_mean), alpha=self.mean_init_std)
            M = self.num_inducing_points
            eye = torch.eye(M, dtype=self.chol_variational_covar.dtype, device=self.chol_variational_covar.device)
            self.chol_variational_covar.copy_(eye.expand_as(self.chol_variational_covar))


class _VariationalStrategy(Module):
    def __init__(self, model, inducing_points, variational_distribution, learn_inducing_locations=True):
        super().__init__()
        object.__setattr__(self, 'model', model)        # not a submodule (the model owns the strategy)
        inducing_points = inducing_points.clone()
        if inducing_points.dim() == 1:
            inducing_points = inducing_points.unsqueeze(-1)
        if learn_inducing_locations:
            self.register_parameter('inducing_points', torch.nn.Parameter(inducing_points))
        else:
            self.register_buffer('inducing_points', inducing_points)
        self._variational_distribution = variational_distribution
        self.register_buffer('variational_params_initialized', torch.tensor(0))

    @property
    def variational_distribution(self):
        return self._variational_distribution()

    def _maybe_init(self):
        if self.training and not getattr(self, '_init_done', False):
            if not bool(self.variational_params_initialized.item()):        # one host sync, first call only
                self._variational_distribution.initialize_variational_distribution()
                self.variational_params_initialized.fill_(1)
            self._init_done = True


class VariationalStrategy(_VariationalStrategy):
    """Whitened strategy: prior on the whitened inducing values is N(0, I)."""

    def _flat_params(self):
        Z = self.inducing_points
        m = self._variational_distribution.variational_mean
        Lq = self._variational_distribution.chol_variational_covar
        if Z.dim() == 2:
            Z, m, Lq = Z.unsqueeze(0), m.unsqueeze(0), Lq.unsqueeze(0)
        b, M, D = Z.shape
        kern = self.model.covar_module
        if not (isinstance(kern, ScaleKernel) and isinstance(kern.base_kernel, RBFKernel)):
            raise NotImplementedError('VariationalStrategy on the MI355X path supports ScaleKernel(RBFKernel) '
                                      '(the only kernel models/dgps.py builds)')
        ls = kern.base_kernel.lengthscale.reshape(-1, kern.base_kernel.lengthscale.shape[-1])
        if ls.shape[-1] != D:
            ls = ls.expand(ls.shape[0], D)
        ls = ls.expand(b, D)
        os_ = kern.outputscale.reshape(-1).expand(b)
        return Z, ls, os_, m, Lq

    def whiten_group(self):
        """(Z, ls, os) of this layer's GPs for the batched Kzz -> Cholesky -> inverse chain."""
        Z, ls, os_, _, _ = self._flat_params()
        return Z, ls.contiguous(), os_.contiguous()

    def marginals(self, x_flat, feeds_next=False):
        """x_flat:(n,D) shared by all output GPs -> mean (b,n) incl. the prior mean, var (b,n).
        feeds_next: this layer's output is the next layer's input (settings.hidden_kzx_f64: Kzx built in float64)."""
        self._maybe_init()
        Z, ls, os_, m, Lq = self._flat_params()
        jitter = settings.variational_cholesky_jitter.value(x_flat.dtype)
        W64 = getattr(self, '_W64_shared', None)         # set by DeepGP.__call__ (one chain for all layers)
        routed = getattr(self, '_kernel_params_shared', None)
        if W64 is not None and routed is not None:
            Z, ls, os_ = routed                          # same values; gradients return through the whitening node
        b = Z.shape[0]
        fused, mean_w, mean_c = self._affine_prior_mean(b, x_flat.shape[-1])
        mean, var, _info = svgp_marginal(x_flat, Z, ls.contiguous(), os_.contiguous(), m, Lq, jitter=jitter,
                                         chol_bwd_f64=settings.chol_bwd_f64.on(), W64=W64, mean_w=mean_w,
                                         mean_c=mean_c, W64f=getattr(self, '_W64f_shared', None) if W64 is not None else None,
                                         kzx_f64=bool(feeds_next) and settings.hidden_kzx_f64.on())
        if fused:
            return mean, var
        xin = x_flat if b == 1 and self.inducing_points.dim() == 2 else x_flat.unsqueeze(0).expand(b, *x_flat.shape)
        prior_mean = self.model.mean_module(xin).reshape(b, -1)
        return mean + prior_mean, var

    def _affine_prior_mean(self, b, D):
        """(fused, weights, constant) of the layer's mean module when it is one of the stock affine means (exact types:
        a subclass may override forward), in the layouts SVGPLayerFn takes; (False, None, None) otherwise."""
        mm = getattr(self.model, 'mean_module', None)
        if type(mm) is ZeroMean:
            return True, None, None
        if type(mm) is ConstantMean and mm.constant.numel() in (1, b):
            return True, None, mm.constant.reshape(-1)
        if type(mm) is LinearMean and mm.weights.shape[-2] == D and mm.weights.numel() in (D, b * D) \
                and (mm.bias is None or mm.bias.numel() * D == mm.weights.numel()):
            return True, mm.weights.reshape(-1, D), None if mm.bias is None else mm.bias.reshape(-1)
        return False, None, None

    def full_covariance(self, x):
        """Dense q(f) covariance at x:(S,n,D) for a single-output GP (used only by predict / nlpd):
        Kxx + 1e-4 I + C^T C - A^T A per sample, batched GEMMs; no autograd."""
        with torch.no_grad():
            Z, ls, os_, m, Lq = self._flat_params()
            if Z.shape[0] != 1:
                raise NotImplementedError('full_covariance: single-output layer expected')
            jitter = settings.variational_cholesky_jitter.value(x.dtype)
            Kzz = ops.rbf_build(Z.double(), Z.double(), ls.double(), os_.double(), diag_add=jitter)
            L, _ = ops.potrf(Kzz, overwrite=True)
            W64 = ops.trtri(L)[0]
            S, n, D = x.shape
            lsS, osS = ls.expand(S, D).contiguous(), os_.expand(S).contiguous()
            Kzx = ops.rbf_build(Z[0], x, lsS, osS)                         # (S,M,n)
            if x.dtype == torch.float32 and settings.whiten_matmul_f64.on():
                # A = L^-1 Kzx accumulated in float64 like the reference's solve (settings.whiten_matmul_f64); predict
                # is not a hot path: a plain float64 GEMM on a float64 copy of Kzx
                A = ops.cast(ops.gemm(W64, ops.cast(Kzx, torch.float64), flags=ops.GEMM_A_LOWER), torch.float32)
            else:
                A = ops.gemm(ops.cast(W64, x.dtype), Kzx, flags=ops.GEMM_A_LOWER)
            C = ops.gemm(Lq[0], A, ta=True, flags=ops.GEMM_A_UPPER)
            Kxx = ops.rbf_build(x, x, lsS, osS, diag_add=VAR_JITTER)
            ops.gemm(C, C, ta=True, beta=1.0, out=Kxx)
            ops.gemm(A, A, ta=True, alpha=-1.0, beta=1.0, out=Kxx)
            return Kxx

    def kl_divergence(self):
        """KL(N(m, Lq Lq^T) || N(0, I)) per batch element, shape batch_shape."""
        m = self._variational_distribution.variational_mean
        Lq = self._variational_distribution.chol_variational_covar
        return ops.KlWhitenedFn.apply(m, Lq)

    def __call__(self, x, prior=False, **kwargs):
        if prior:
            return self.model.forward(x)
        if x.dim() == 2:
            mean, var = self.marginals(x)
            return _DiagMVN(mean[0], var[0], self, x.unsqueeze(0))
        S, n, D = x.shape
        mean, var = self.marginals(x.reshape(S * n, D))
        return _DiagMVN(mean.reshape(S, n), var.reshape(S, n), self, x)


class _DiagMVN(MultivariateNormal):
    """q(f) with marginal variances materialised and the dense covariance computed on demand."""
    _diag_only = True

    def __init__(self, mean, var, strategy=None, x=None, noise=None, _cov=None):
        self.loc = mean
        self._var = var
        self._strategy, self._x, self._noise = strategy, x, noise
        self._islazy = False
        self._cov_cache = _cov

    @property
    def variance(self):
        return self._var

    @property
    def _covar(self):
        return self.covariance_matrix

    @property
    def covariance_matrix(self):
        if self._cov_cache is None:
            K = self._strategy.full_covariance(self._x)
            if self._noise is not None:
                n = K.shape[-1]
                K = K + self._noise.detach() * torch.eye(n, dtype=K.dtype, device=K.device)
            self._cov_cache = K if self.loc.dim() == 2 else K[0]
        return self._cov_cache

    @property
    def lazy_covariance_matrix(self):
        from .lazy import NonLazyTensor
        return NonLazyTensor(self.covariance_matrix)

    @property
    def batch_shape(self):
        return self.loc.shape[:-1]

    def _with_noise(self, noise):
        return _DiagMVN(self.loc, self._var + noise, self._strategy, self._x,
                        noise if self._noise is None else self._noise + noise)

    def __class_getitem__(cls, item):
        return cls
