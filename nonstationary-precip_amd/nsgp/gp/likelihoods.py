"""GaussianLikelihood [gpytorch.likelihoods recalled, SURVEY A.1/A.5]: noise = softplus(raw_noise) + 1e-4
(raw init 0 -> 0.6932), state-dict key `noise_covar.raw_noise`; `likelihood(dist)` adds noise * I,
expected_log_prob / log_marginal as used by VariationalELBO and models/dgps.py:109."""
import math

import torch

from .. import ops
from .constraints import GreaterThan
from .distributions import MultivariateNormal, MultitaskMultivariateNormal
from .lazy import LazyTensor, DiagLazyTensor, lazify
from .module import Module


class Likelihood(Module):
    pass


class HomoskedasticNoise(Module):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size()):
        super().__init__()
        self.register_parameter('raw_noise', torch.nn.Parameter(torch.zeros(*batch_shape, 1)))
        self.register_constraint('raw_noise', noise_constraint or GreaterThan(1e-4))
        if noise_prior is not None:
            self.register_prior('noise_prior', noise_prior, lambda m: m.noise)

    @property
    def noise(self):
        return self._get_constrained('raw_noise')

    @noise.setter
    def noise(self, value):
        self._set_constrained('raw_noise', value)


class GaussianLikelihood(Likelihood):
    def __init__(self, noise_prior=None, noise_constraint=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.noise_covar = HomoskedasticNoise(noise_prior, noise_constraint, batch_shape)

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, value):
        self.noise_covar.noise = value

    @property
    def raw_noise(self):
        return self.noise_covar.raw_noise

    def _shaped_noise_covar(self, shape, *params, **kwargs):
        n = shape[-1]
        return DiagLazyTensor(self.noise.expand(*shape[:-1], n) if len(shape) > 1 else self.noise.expand(n))

    def forward(self, function_samples, *params, **kwargs):
        return torch.distributions.Normal(function_samples, self.noise.sqrt())

    def marginal(self, function_dist, *params, **kwargs):
        mean = function_dist.mean
        noise = self.noise
        if isinstance(function_dist, MultitaskMultivariateNormal):
            return MultitaskMultivariateNormal(mean, None, _var=function_dist.variance + noise,
                                               _tsn=function_dist._tsn)
        if getattr(function_dist, '_diag_only', False):
            return function_dist._with_noise(noise)
        covar = function_dist.lazy_covariance_matrix
        return function_dist.__class__(mean, covar.add_diag(noise.reshape(-1)[:1] if noise.numel() == 1 else noise))

    def __call__(self, input, *params, **kwargs):
        if torch.is_tensor(input):
            return self.forward(input, *params, **kwargs)
        return self.marginal(input, *params, **kwargs)

    def expected_log_prob(self, target, input, *params, **kwargs):
        """-1/2 [((y - mu)^2 + v)/noise + log noise + log 2 pi]  per point (shape of input.mean)."""
        mean, variance = input.mean, input.variance
        noise = self.noise
        res = ((target - mean) ** 2 + variance) / noise + noise.log() + math.log(2 * math.pi)
        return res.mul(-0.5)

    def log_marginal(self, observations, function_dist, *params, **kwargs):
        """Normal(mu, sqrt(clamp_min(v + noise, 1e-8))).log_prob(y)  (models/dgps.py:109)."""
        mean, variance = function_dist.mean, function_dist.variance
        v = (variance + self.noise).clamp_min(1e-8)
        return -0.5 * ((observations - mean) ** 2 / v + v.log() + math.log(2 * math.pi))


class _OneDimensionalLikelihood(Likelihood):
    pass
