"""DSVI deep GP on the MI355X engine -- drop-in for models/dgps.py of the reference.

Same surface (reference file:line):
  num_output_dims                                   models/dgps.py:13
  DeepGPHiddenLayer(input_dims, output_dims, num_inducing=250, mean_type='constant')   :15-70
  DeepGP(num_layers, train_x_shape)  with .layers / .last_layer / .likelihood / .forward / .predict(loader)   :72-111
  ExactGPModel(train_x, train_y, likelihood, kernel)                                    :113-122
Reference quirks kept: the hidden layers are ONE tied layer object repeated
num_layers times, and predict() returns only the last batch's distribution.  Additions (keyword-only,
defaults reproduce the reference): DeepGP(..., num_inducing=250, tie_layers=True).
"""
import torch

import nsgp.gp as gpytorch
from nsgp.gp import settings
from nsgp.gp.distributions import MultivariateNormal, MultitaskMultivariateNormal
from nsgp.gp.kernels import RBFKernel, ScaleKernel
from nsgp.gp.likelihoods import GaussianLikelihood
from nsgp.gp.means import ConstantMean, LinearMean
from nsgp.gp.models import DeepGPLayer, DeepGP as _DeepGPBase, ExactGP
from nsgp.gp.variational import CholeskyVariationalDistribution, VariationalStrategy

num_output_dims = 2


class DeepGPHiddenLayer(DeepGPLayer):
    """One whitened SVGP layer: Z ~ randn, q(u) = N(m, Lq Lq^T), ScaleKernel(RBF-ARD)."""

    def __init__(self, input_dims, output_dims, num_inducing=250, mean_type='constant'):
        batch_shape = torch.Size([]) if output_dims is None else torch.Size([output_dims])
        Z = torch.randn(*batch_shape, num_inducing, input_dims)
        q_u = CholeskyVariationalDistribution(num_inducing_points=num_inducing, batch_shape=batch_shape)
        strategy = VariationalStrategy(self, Z, q_u, learn_inducing_locations=True)
        super().__init__(strategy, input_dims, output_dims)
        self.mean_module = ConstantMean(batch_shape=batch_shape) if mean_type == 'constant' \
            else LinearMean(input_dims)
        self.covar_module = ScaleKernel(RBFKernel(batch_shape=batch_shape, ard_num_dims=input_dims),
                                        batch_shape=batch_shape, ard_num_dims=None)

    def forward(self, x):
        """Prior at x (what gpytorch's VariationalStrategy evaluates at [Z; x])."""
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))

    def __call__(self, x, *other_inputs, **kwargs):
        """Concatenation skip connections, as in the reference (unused by DeepGP.forward)."""
        if len(other_inputs):
            if isinstance(x, MultitaskMultivariateNormal):
                x = x.rsample()
            S = settings.num_likelihood_samples.value()
            x = torch.cat([x] + [inp.unsqueeze(0).expand(S, *inp.shape) for inp in other_inputs], dim=-1)
        return super().__call__(x, are_samples=bool(len(other_inputs)))


class DeepGP(_DeepGPBase):
    def __init__(self, num_layers, train_x_shape, *, num_inducing=250, tie_layers=True):
        hidden = DeepGPHiddenLayer(input_dims=train_x_shape[-1], output_dims=num_output_dims,
                                   num_inducing=num_inducing, mean_type='linear')
        last = DeepGPHiddenLayer(input_dims=hidden.output_dims, output_dims=None,
                                 num_inducing=num_inducing, mean_type='constant')
        super().__init__()
        if tie_layers:
            stack = [hidden for _ in range(num_layers)]
        else:
            stack = [hidden] + [DeepGPHiddenLayer(hidden.output_dims, num_output_dims, num_inducing, 'linear')
                                for _ in range(num_layers - 1)]
        self.layers = torch.nn.ModuleList(stack)
        self.last_layer = last
        self.likelihood = GaussianLikelihood()

    def forward(self, inputs):
        rep = inputs
        for layer in self.layers:
            rep = layer(rep)
        return self.last_layer(rep)

    def predict(self, test_loader):
        """Returns (predictive of the LAST batch, means (S,N), variances (S,N), per-point log marginals (S,N)):
        the tuple of the reference's DeepGP.predict (:100-111).  Each batch is propagated twice, once for the
        noisy predictive and once for the log marginal of its targets, exactly like the reference."""
        parts = {'mean': [], 'variance': [], 'll': []}
        last_predictive = None
        with torch.no_grad():
            for xb, yb in test_loader:
                last_predictive = self.likelihood(self(xb))
                parts['mean'].append(last_predictive.mean)
                parts['variance'].append(last_predictive.variance)
                parts['ll'].append(self.likelihood.log_marginal(yb, self(xb)))
        joined = {k: torch.cat(v, dim=-1) for k, v in parts.items()}
        return last_predictive, joined['mean'], joined['variance'], joined['ll']


class ExactGPModel(ExactGP):
    """Constant-mean exact GP over a caller-supplied kernel (SE-ARD baselines)."""

    def __init__(self, train_x, train_y, likelihood, kernel):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = ConstantMean()
        self.covar_module = kernel

    def forward(self, x):
        return MultivariateNormal(self.mean_module(x), self.covar_module(x))
