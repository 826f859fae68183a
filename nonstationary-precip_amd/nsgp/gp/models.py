"""GP model bases [gpytorch.models recalled, SURVEY A.4/A.6]: ExactGP (train mode returns the prior
over the training inputs, eval mode the exact posterior by Cholesky), ApproximateGP, and
deep_gps.{DeepGPLayer, DeepGP} as subclassed by models/dgps.py:15-111."""
import torch

from .. import ops
from . import settings
from .distributions import MultivariateNormal, MultitaskMultivariateNormal
from .lazy import delazify
from .likelihoods import GaussianLikelihood
from .module import Module, cut_cached_transforms, prefill_softplus_transforms, transform_cache


class GP(Module):
    pass


class ExactGP(GP):
    def __init__(self, train_inputs, train_targets, likelihood):
        if train_inputs is not None and torch.is_tensor(train_inputs):
            train_inputs = (train_inputs,)
        super().__init__()
        self.train_inputs = None if train_inputs is None else tuple(
            t.unsqueeze(-1) if t.ndimension() == 1 else t for t in train_inputs)
        self.train_targets = train_targets
        self.likelihood = likelihood
        self.prediction_strategy = None

    def _apply(self, fn):
        # .cuda()/.double()/.to() must move the stored data too (experiments/spatial_exp.py:173)
        if self.train_inputs is not None:
            self.train_inputs = tuple(fn(t) for t in self.train_inputs)
            self.train_targets = fn(self.train_targets)
        return super()._apply(fn)

    def train(self, mode=True):
        if mode:
            self.prediction_strategy = None
        return super().train(mode)

    def set_train_data(self, inputs=None, targets=None, strict=True):
        if inputs is not None:
            if torch.is_tensor(inputs):
                inputs = (inputs,)
            self.train_inputs = tuple(t.unsqueeze(-1) if t.ndimension() == 1 else t for t in inputs)
        if targets is not None:
            self.train_targets = targets
        self.prediction_strategy = None

    def __call__(self, *args, **kwargs):
        inputs = [a.unsqueeze(-1) if a.ndimension() == 1 else a for a in args]
        train_inputs = list(self.train_inputs) if self.train_inputs is not None else []
        if self.training:
            if self.train_inputs is None:
                raise RuntimeError('train_inputs, train_targets cannot be None in training mode. '
                                   'Call .eval() for prior predictions, or call .set_train_data() to add training data.')
            if settings.debug.on():
                from .kernels import same_points
                if not all(same_points(ti, i) for ti, i in zip(train_inputs, inputs)):
                    raise RuntimeError('You must train on the training inputs!')
            return super().__call__(*inputs, **kwargs)
        if self.train_inputs is None or self.train_targets is None:
            return super().__call__(*inputs, **kwargs)
        # exact posterior (always-Cholesky; gpytorch switches to CG above 800 points)
        x_tr, x_te = train_inputs[0], inputs[0]
        ntr = x_tr.shape[-2]
        full = torch.cat([x_tr, x_te], dim=-2)
        full_out = super().__call__(full, **kwargs)
        full_mean, K = full_out.loc, delazify(full_out.lazy_covariance_matrix)
        noise = self.likelihood.noise
        Ktt = K[..., :ntr, :ntr] + noise * torch.eye(ntr, dtype=K.dtype, device=K.device)
        W, _ = ops.chol_inv(Ktt.contiguous())
        resid = (self.train_targets - full_mean[..., :ntr]).unsqueeze(-1)
        a = ops.matmul(W, resid, a_lower=True)
        Kst = K[..., ntr:, :ntr].contiguous()
        V = ops.matmul(W, Kst, False, True, a_lower=True)                   # W K_ts
        mean = ops.matmul(V, a, True, False).squeeze(-1) + full_mean[..., ntr:]
        cov = K[..., ntr:, ntr:] - ops.matmul(V, V, True, False)
        return full_out.__class__(mean, cov)


class ApproximateGP(GP):
    def __init__(self, variational_strategy):
        super().__init__()
        self.variational_strategy = variational_strategy

    def forward(self, x):
        raise NotImplementedError

    def __call__(self, inputs, prior=False, **kwargs):
        if inputs.dim() == 1:
            inputs = inputs.unsqueeze(-1)
        return self.variational_strategy(inputs, prior=prior, **kwargs)


class DeepGPLayer(ApproximateGP):
    """gpytorch.models.deep_gps.DeepGPLayer.__call__ semantics: a MultitaskMVN input is
    sampled through its marginals only, the sample is shared by the layer's output GPs, deterministic
    inputs are expanded to num_likelihood_samples."""

    def __init__(self, variational_strategy, input_dims, output_dims):
        super().__init__(variational_strategy)
        self.input_dims = input_dims
        self.output_dims = output_dims

    def forward(self, x):
        raise NotImplementedError

    def __call__(self, inputs, are_samples=False, **kwargs):
        deterministic_inputs = not are_samples
        if isinstance(inputs, MultitaskMultivariateNormal):
            S = inputs.loc.shape[0]
            mean_tsn, var_tsn = inputs._tsn
            b, ns, n = mean_tsn.shape
            prov = settings.eps_provider.value()
            if prov is None:
                eps = torch.randn(S, n, b, dtype=mean_tsn.dtype, device=mean_tsn.device)
            else:
                eps = prov((S, n, b), mean_tsn.dtype, mean_tsn.device)
            inputs = ops.DgpSampleFn.apply(mean_tsn, var_tsn, eps)           # (S, n, b)
            deterministic_inputs = False
            plan = settings.backward_stages.value()
            if plan is not None:                         # staged backward: this layer and what follows form a stage
                inputs, = plan.cut([inputs])
        if settings.debug.on():
            if not torch.is_tensor(inputs):
                raise ValueError('`inputs` should either be a MultitaskMultivariateNormal or a Tensor, got '
                                 f'{inputs.__class__.__name__}')
            if inputs.size(-1) != self.input_dims:
                raise RuntimeError(f'Input shape did not match self.input_dims. Got total feature dims '
                                   f'[{inputs.size(-1)}], expected [{self.input_dims}]')
        S = settings.num_likelihood_samples.value()
        vs = self.variational_strategy
        if inputs.dim() == 2:
            n, ns = inputs.shape[0], 1
            flat = inputs
        else:
            ns, n = inputs.shape[0], inputs.shape[1]
            flat = inputs.reshape(ns * n, inputs.shape[-1])
        mean, var = vs.marginals(flat, feeds_next=self.output_dims is not None)    # (b, ns*n)
        b = mean.shape[0]
        if self.output_dims is not None:
            mean_tsn, var_tsn = mean.reshape(b, ns, n), var.reshape(b, ns, n)
            Sout = S if deterministic_inputs else ns
            m_view = mean_tsn.permute(1, 2, 0).expand(Sout, n, b)
            v_view = var_tsn.permute(1, 2, 0).expand(Sout, n, b)
            return MultitaskMultivariateNormal(m_view, None, _var=v_view, _tsn=(mean_tsn, var_tsn))
        from .variational import _DiagMVN
        if ns == 1 and inputs.dim() == 2:
            out = _DiagMVN(mean[0], var[0], vs, inputs.unsqueeze(0))
            if deterministic_inputs:
                out = _DiagMVN(mean.expand(S, n), var.expand(S, n), vs, inputs.unsqueeze(0).expand(S, n, -1))
            return out
        return _DiagMVN(mean.reshape(ns, n), var.reshape(ns, n), vs, inputs)


class _DeepGPVariationalStrategy(object):
    def __init__(self, model):
        self.model = model

    @property
    def sub_variational_strategies(self):
        if not hasattr(self, '_sub_variational_strategies_memo'):
            self._sub_variational_strategies_memo = [
                m.variational_strategy for m in self.model.modules() if isinstance(m, ApproximateGP)]
        return self._sub_variational_strategies_memo

    def kl_divergence(self):
        # tied layers are one module -> counted once
        total = None
        for s in self.sub_variational_strategies:
            kl = s.kl_divergence()
            kl = kl.reshape(()) if kl.numel() == 1 else kl.sum()
            total = kl if total is None else total + kl
        return total if total is not None else 0.0


class DeepGP(GP):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, 'variational_strategy', _DeepGPVariationalStrategy(self))

    def forward(self, x):
        raise NotImplementedError

    def __call__(self, *args, **kwargs):
        """Kzz, its Cholesky factor and inverse depend on parameters only: do them for EVERY layer of
        the model up front in one batched potrf/trtri chain (one serial dependency chain per step
        instead of one per layer), then run the user's forward."""
        from ..svgp import whiten
        strategies = [s for s in self.variational_strategy.sub_variational_strategies
                      if hasattr(s, 'whiten_group')]
        shared = False
        plan = settings.backward_stages.value()
        if plan is not None:
            plan.begin()
        with transform_cache():
            if args and torch.is_tensor(args[0]) and args[0].is_cuda:
                keys = prefill_softplus_transforms(self)  # all raw hyper-parameters in one softplus launch
                if plan is not None:
                    cut_cached_transforms(keys, plan)
            if len(strategies) > 1 and len({s.inducing_points.shape[-2] for s in strategies}) == 1 \
                    and args and torch.is_tensor(args[0]) and args[0].is_cuda:
                for s in strategies:
                    s._maybe_init()
                groups = [s.whiten_group() for s in strategies]
                Ws, _info, passed, W64s = whiten(groups, settings.variational_cholesky_jitter.value(args[0].dtype),
                                                 settings.chol_bwd_f64.on(), passthrough=True, out_dtype=args[0].dtype,
                                                 with_f64=True)
                if plan is not None:                     # staged backward: the whitening chain's adjoint is its own stage
                    flat = plan.cut(list(Ws) + [t for zlo in passed for t in zlo])
                    Ws, passed = flat[:len(Ws)], [tuple(flat[len(Ws) + 3 * i:len(Ws) + 3 * i + 3])
                                                  for i in range(len(passed))]
                for s, W, zlo, Wd in zip(strategies, Ws, passed, W64s):
                    s._W64_shared = W
                    s._W64f_shared = Wd                      # float64 companion (forward projection accumulates in it)
                    s._kernel_params_shared = zlo            # (Z, ls, os) routed through the whitening node
                shared = True
            try:
                return super().__call__(*args, **kwargs)
            finally:
                if shared:
                    for s in strategies:
                        s._W64_shared = None
                        s._W64f_shared = None
                        s._kernel_params_shared = None


class DeepLikelihood(GaussianLikelihood):
    pass
