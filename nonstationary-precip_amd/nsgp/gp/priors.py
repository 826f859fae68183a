"""Priors [gpytorch.priors recalled]: a Prior is an nn.Module with log_prob; MultivariateNormalPrior is
the base of models/latent_priors.py:27 (MatrixVariateNormalPrior)."""
import torch

from .module import Module
from .distributions import MultivariateNormal


class Prior(Module):
    def log_prob(self, x):
        raise NotImplementedError


class MultivariateNormalPrior(Prior):
    def __init__(self, loc, covariance_matrix=None, precision_matrix=None, scale_tril=None, validate_args=False,
                 transform=None):
        super().__init__()
        if covariance_matrix is None:
            raise NotImplementedError('MultivariateNormalPrior: pass covariance_matrix')
        self.register_buffer('loc', loc)
        self.register_buffer('covariance_matrix', covariance_matrix)
        self._transform = transform

    def _dist(self):
        return MultivariateNormal(self.loc, self.covariance_matrix)

    def log_prob(self, x):
        if self._transform is not None:
            x = self._transform(x)
        return self._dist().log_prob(x.to(self.loc.dtype))

    def rsample(self, sample_shape=torch.Size()):
        return self._dist().rsample(sample_shape)

    def sample_n(self, n):
        return self._dist().sample(torch.Size((n,)))

    def sample(self, sample_shape=torch.Size()):
        return self._dist().sample(sample_shape)


class NormalPrior(Prior):
    def __init__(self, loc, scale, validate_args=False, transform=None):
        super().__init__()
        self.register_buffer('loc', torch.as_tensor(float(loc)) if not torch.is_tensor(loc) else loc)
        self.register_buffer('scale', torch.as_tensor(float(scale)) if not torch.is_tensor(scale) else scale)

    def log_prob(self, x):
        return torch.distributions.Normal(self.loc, self.scale).log_prob(x)
