"""Module base with constraints, priors and added-loss terms -- the gpytorch.Module mechanics the
reference models rely on (register_parameter / register_prior with a closure at
models/nonstationary_models.py:31-38, update_added_loss_term at models/gibbs_kernels.py:261,
`model.covar_module.outputscale = 0.644` style setters at experiments/spatial_exp.py:176-186)
[gpytorch semantics recalled, SURVEY A.1/A.5]."""
from collections import OrderedDict

import torch
from torch import nn

from .constraints import Interval


# Scoped cache of constrained parameter values: inside `transform_cache()` every softplus(raw) is computed once
# per (parameter, version) instead of once per property access (a DSVI step reads each lengthscale /
# outputscale three times).  The scope ends with the forward pass, so no autograd graph outlives its backward.
_transform_cache = []
_lower_bound_vectors = {}


def transform_cache_active():
    """True inside a `transform_cache()` scope.  The cache keys on parameter versions, which the raw-pointer Adam
    kernel (nsgp_adam_step_f32) never bumps: a scope must not span an optimiser step (FusedAdam.step checks)."""
    return bool(_transform_cache)


class transform_cache:
    def __enter__(self):
        _transform_cache.append(_transform_cache[-1] if _transform_cache else {})   # nested scopes share
        return self

    def __exit__(self, *exc):
        _transform_cache.pop()
        return False


class _SplitPackedFn(torch.autograd.Function):
    """packed:(n,) -> views of it in the given shapes; the backward is ONE cat of the incoming gradients."""

    @staticmethod
    def forward(ctx, packed, *shapes):
        ctx.shapes = shapes
        outs, off = [], 0
        for shp in shapes:
            n = 1
            for d in shp:
                n *= d
            outs.append(packed[off:off + n].view(shp))
            off += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        parts = []
        for g, shp in zip(grads, ctx.shapes):
            if g is None:
                n = 1
                for d in shp:
                    n *= d
                parts.append(None if n == 0 else n)
            else:
                parts.append(g.reshape(-1))
        ref = next(p for p in parts if torch.is_tensor(p))
        parts = [p if torch.is_tensor(p) else ref.new_zeros(p) for p in parts if p is not None]
        return (torch.cat(parts), *([None] * len(ctx.shapes)))


def prefill_softplus_transforms(root):
    """Inside a `transform_cache()` scope: compute softplus(raw) for EVERY softplus-constrained CUDA parameter under
    `root` with one cat + one softplus (and one add where a lower bound is non-zero) instead of a launch per parameter,
    and seed the cache with views of the result; the backward is one cat + one softplus_backward.  (A DSVI step of
    models/dgps.py has five such parameters: two lengthscales, two output scales, the noise.)"""
    import math
    if not _transform_cache:
        return []
    cache = _transform_cache[-1]
    todo, seen = [], set()
    for mod in root.modules():
        cons = getattr(mod, '_constraints', None)
        if not cons:
            continue
        for cname, c in cons.items():
            p = mod._parameters.get(cname[:-len('_constraint')])
            if p is None or id(p) in seen or not p.is_cuda or type(c).transform is not Interval.transform \
                    or not math.isinf(float(c.upper_bound)) or p.numel() == 0:
                continue
            key = (id(p), p._version, torch.is_grad_enabled())
            if key in cache:
                continue
            seen.add(id(p))
            todo.append((key, p, float(c.lower_bound)))
    if len(todo) < 2 or len({(t[1].dtype, t[1].device) for t in todo}) != 1:
        return []
    sp = torch.nn.functional.softplus(torch.cat([p.reshape(-1) for _, p, _ in todo]))
    if any(lb != 0.0 for _, _, lb in todo):
        k = (tuple((id(p), p.numel(), lb) for _, p, lb in todo), sp.dtype, sp.device)
        vec = _lower_bound_vectors.get(k)
        if vec is None:                                   # built once (before any graph capture), then resident
            vec = _lower_bound_vectors[k] = torch.cat(
                [torch.full((p.numel(),), lb, dtype=sp.dtype) for _, p, lb in todo]).to(sp.device)
        sp = sp + vec
    for (key, _, _), val in zip(todo, _SplitPackedFn.apply(sp, *[tuple(p.shape) for _, p, _ in todo])):
        cache[key] = val
    return [key for key, _, _ in todo]


def cut_cached_transforms(keys, plan):
    """Staged backward (nsgp/stages.py): the cached softplus values under `keys` become leaves of `plan`."""
    if not keys or not _transform_cache:
        return
    cache = _transform_cache[-1]
    for key, leaf in zip(keys, plan.cut([cache[k] for k in keys])):
        cache[key] = leaf


class Module(nn.Module):
    def __init__(self):
        super().__init__()
        self._added_loss_terms = OrderedDict()
        self._priors = OrderedDict()
        self._constraints = OrderedDict()

    # ---- parameters / constraints ---------------------------------------------------------
    def register_parameter(self, name, parameter):
        if '_parameters' not in self.__dict__:
            raise AttributeError('Cannot assign parameter before Module.__init__() call')
        super().register_parameter(name, parameter)

    def register_constraint(self, param_name, constraint, replace=True):
        if param_name not in self._parameters:
            raise RuntimeError(f'Attempting to register constraint for nonexistent parameter {param_name}')
        name = param_name + '_constraint'
        self.add_module(name, constraint)
        self._constraints[name] = constraint

    def constraint_for_parameter_name(self, param_name):
        return self._constraints.get(param_name + '_constraint')

    def initialize(self, **kwargs):
        """Set parameters (raw names) or constrained properties by keyword, e.g. initialize(noise=0.011)."""
        for name, val in kwargs.items():
            if isinstance(val, (int, float)):
                val = float(val)
            if '.' in name:
                mod, rest = name.split('.', 1)
                getattr(self, mod).initialize(**{rest: val})
            elif name in self._parameters:
                p = self._parameters[name]
                with torch.no_grad():
                    if torch.is_tensor(val):
                        p.copy_(val.to(p).expand_as(p) if val.numel() != p.numel() else val.to(p).view_as(p))
                    else:
                        p.fill_(val)
            elif hasattr(type(self), name) and isinstance(getattr(type(self), name), property):
                setattr(self, name, val)
            else:
                raise AttributeError(f'Unknown parameter {name} for {type(self).__name__}')
        return self

    def _set_constrained(self, raw_name, value):
        """Apply the inverse transform and write the raw parameter in place (keeps Parameter identity)."""
        p = self._parameters[raw_name]
        if not torch.is_tensor(value):
            value = torch.as_tensor(value)
        value = value.to(dtype=p.dtype, device=p.device)
        c = self.constraint_for_parameter_name(raw_name)
        raw = c.inverse_transform(value) if c is not None else value
        with torch.no_grad():
            p.copy_(raw.expand_as(p) if raw.numel() != p.numel() else raw.reshape(p.shape))

    def _get_constrained(self, raw_name):
        p = self._parameters[raw_name]
        c = self.constraint_for_parameter_name(raw_name)
        if c is None:
            return p
        cache = _transform_cache[-1] if _transform_cache else None
        if cache is None:
            return c.transform(p)
        key = (id(p), p._version, torch.is_grad_enabled())
        val = cache.get(key)
        if val is None:
            val = cache[key] = c.transform(p)
        return val

    # ---- priors -----------------------------------------------------------------------------
    def register_prior(self, name, prior, param_or_closure, setting_closure=None):
        if isinstance(param_or_closure, str):
            pname = param_or_closure
            if pname not in self._parameters and not hasattr(self, pname):
                raise AttributeError(f'Unknown parameter {pname} for {type(self).__name__}')

            def closure(module, _p=pname):
                return getattr(module, _p)
        else:
            closure = param_or_closure
        self.add_module(name, prior)
        self._priors[name] = (prior, closure, setting_closure)

    def named_priors(self, memo=None, prefix=''):
        if memo is None:
            memo = set()
        if hasattr(self, '_priors'):
            for name, (prior, closure, inv) in self._priors.items():
                if prior is not None and prior not in memo:
                    memo.add(prior)
                    yield prefix + ('.' if prefix else '') + name, self, prior, closure, inv
        for mname, module in self.named_children():
            sub = prefix + ('.' if prefix else '') + mname
            if hasattr(module, 'named_priors'):
                yield from module.named_priors(memo, sub)

    # ---- added loss terms -------------------------------------------------------------------
    def register_added_loss_term(self, name):
        self._added_loss_terms[name] = None

    def update_added_loss_term(self, name, added_loss_term):
        self._added_loss_terms[name] = added_loss_term

    def added_loss_terms(self):
        seen = set()
        for m in self.modules():
            if id(m) in seen:
                continue
            seen.add(id(m))
            for term in getattr(m, '_added_loss_terms', {}).values():
                if term is not None:
                    yield term

    def hyperparameters(self):
        for _, p in self.named_parameters():
            yield p
