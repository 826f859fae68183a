"""Mean functions [gpytorch.means semantics recalled, SURVEY A.1]: ZeroMean, ConstantMean (`constant`
parameter of shape (*batch, 1), init 0; experiments/spatial_exp.py:162-164 overwrites it with a new
Parameter), LinearMean (`weights` (input_size, 1) and `bias` (1,) ~ randn; models/dgps.py:43)."""
import torch

from .module import Module


class Mean(Module):
    def __call__(self, x):
        if x.dim() == 1:
            x = x.unsqueeze(1)
        return self.forward(x)


class ZeroMean(Mean):
    def __init__(self, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape

    def forward(self, x):
        shp = torch.broadcast_shapes(self.batch_shape, x.shape[:-2])
        return torch.zeros(*shp, x.shape[-2], dtype=x.dtype, device=x.device)


class ConstantMean(Mean):
    def __init__(self, prior=None, batch_shape=torch.Size(), **kwargs):
        super().__init__()
        self.batch_shape = batch_shape
        self.register_parameter('constant', torch.nn.Parameter(torch.zeros(*batch_shape, 1)))
        if prior is not None:
            self.register_prior('mean_prior', prior, 'constant')

    def forward(self, x):
        c = self.constant
        if c.shape[:-1] == x.shape[:-2] or c.dim() == 1:
            return c.expand(*x.shape[:-1])
        shp = torch.broadcast_shapes(c.shape[:-1], x.shape[:-2])
        return c.expand(*shp, 1).expand(*shp, x.shape[-2])


class LinearMean(Mean):
    def __init__(self, input_size, batch_shape=torch.Size(), bias=True):
        super().__init__()
        self.register_parameter('weights', torch.nn.Parameter(torch.randn(*batch_shape, input_size, 1)))
        if bias:
            self.register_parameter('bias', torch.nn.Parameter(torch.randn(*batch_shape, 1)))
        else:
            self.bias = None

    def forward(self, x):
        res = (x * self.weights.squeeze(-1).unsqueeze(-2)).sum(-1)     # (n x D)(D x 1) as an elementwise reduce
        if self.bias is not None:
            res = res + self.bias
        return res
