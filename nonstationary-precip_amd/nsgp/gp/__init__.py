"""nsgp.gp -- a from-scratch GP module namespace providing the gpytorch symbol subset that the
in-scope reference scripts and models import, with every O(N M), O(M^3) and
O(M^2 N) operation running on the gfx950 kernels of nsgp.ops.  `install_as_gpytorch()` registers it
under the name `gpytorch` when the real library is absent, so `import gpytorch` in
experiments/*.py resolves to this namespace."""
import sys
import types

from . import constraints, settings, lazy, distributions, means, kernels, likelihoods, priors, mlls, variational, \
    models, utils  # noqa: F401
from .module import Module  # noqa: F401

__version__ = '0.0+nsgp-mi355x'

# gpytorch-style sub-namespace `models.deep_gps`
deep_gps = types.ModuleType(__name__ + '.models.deep_gps')
deep_gps.DeepGPLayer = models.DeepGPLayer
deep_gps.DeepGP = models.DeepGP
deep_gps.DeepLikelihood = models.DeepLikelihood
models.deep_gps = deep_gps
sys.modules[__name__ + '.models.deep_gps'] = deep_gps


def install_as_gpytorch(force=False):
    """Make `import gpytorch` resolve to this namespace (only if the real gpytorch is not importable)."""
    if 'gpytorch' in sys.modules and not force:
        return sys.modules['gpytorch']
    if not force:
        import importlib.util
        if importlib.util.find_spec('gpytorch') is not None:
            import gpytorch            # the genuine library wins
            return gpytorch
    me = sys.modules[__name__]
    sys.modules['gpytorch'] = me
    for sub in ('constraints', 'settings', 'lazy', 'distributions', 'means', 'kernels', 'likelihoods', 'priors',
                'mlls', 'variational', 'models', 'utils'):
        sys.modules['gpytorch.' + sub] = getattr(me, sub)
    sys.modules['gpytorch.models.deep_gps'] = deep_gps
    sys.modules['gpytorch.utils.cholesky'] = utils.cholesky
    sys.modules['gpytorch.utils.broadcasting'] = utils.broadcasting
    return me
