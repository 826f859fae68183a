"""Parameter constraints (softplus transforms), restating gpytorch.constraints [SURVEY A.1].

`Positive`: value = softplus(raw); `GreaterThan(lb)`: value = softplus(raw) + lb; the setters apply the
inverse transform.  Used by GaussianLikelihood.noise (GreaterThan(1e-4)), ScaleKernel.outputscale,
RBFKernel.lengthscale, and spatio_temporal_models.py's outputscale_constraint=GreaterThan(7).
"""
import math

import torch


def inv_softplus(x):
    # log(exp(x) - 1), stable for large x
    return x + torch.log(-torch.expm1(-x))


class Interval(torch.nn.Module):
    def __init__(self, lower_bound, upper_bound, transform=None, inv_transform=None, initial_value=None):
        super().__init__()
        self.lower_bound = torch.as_tensor(float(lower_bound))
        self.upper_bound = torch.as_tensor(float(upper_bound))
        self.initial_value = initial_value

    @property
    def enforced(self):
        return True

    def transform(self, raw):
        lb, ub = float(self.lower_bound), float(self.upper_bound)
        if math.isinf(ub):
            sp = torch.nn.functional.softplus(raw)
            return sp if lb == 0.0 else sp + lb               # Positive: no extra elementwise launch
        return torch.sigmoid(raw) * (ub - lb) + lb

    def inverse_transform(self, value):
        lb, ub = float(self.lower_bound), float(self.upper_bound)
        if math.isinf(ub):
            return inv_softplus(value - lb)
        p = (value - lb) / (ub - lb)
        return torch.log(p) - torch.log1p(-p)

    def check(self, value):
        return bool(torch.all(value <= float(self.upper_bound)) and torch.all(value >= float(self.lower_bound)))

    def __repr__(self):
        return f'{type(self).__name__}({float(self.lower_bound):.3E}, {float(self.upper_bound):.3E})'


class GreaterThan(Interval):
    def __init__(self, lower_bound, transform=None, inv_transform=None, initial_value=None):
        super().__init__(lower_bound, math.inf, initial_value=initial_value)


class Positive(GreaterThan):
    def __init__(self, transform=None, inv_transform=None, initial_value=None):
        super().__init__(0.0, initial_value=initial_value)


class LessThan(Interval):
    def __init__(self, upper_bound, transform=None, inv_transform=None, initial_value=None):
        super().__init__(-math.inf, upper_bound, initial_value=initial_value)

    def transform(self, raw):
        return float(self.upper_bound) - torch.nn.functional.softplus(-raw)

    def inverse_transform(self, value):
        return -inv_softplus(float(self.upper_bound) - value)
