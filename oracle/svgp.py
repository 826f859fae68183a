"""DSVI deep GP restated on CPU (oracle; test infrastructure only).

Follows models/dgps.py:15-111 (DeepGPHiddenLayer, DeepGP incl. the tied hidden layers at :88)
and the gpytorch pieces it executes [recalled -- gpytorch is absent here; SURVEY A.3-A.5]:
whitened VariationalStrategy.forward, CholeskyVariationalDistribution, DeepGPLayer.__call__,
GaussianLikelihood.expected_log_prob, VariationalELBO, DeepApproximateMLL.

A layer is a dict of *constrained* tensors:
  Z:(b,M,D)|(M,D)  lengthscale:(b,1,D)|(1,D)  outputscale:(b,)|()   m:(b,M)|(M,)  Lq:(b,M,M)|(M,M)
  mean: ('constant', const:(b,1)|(1,))  or  ('linear', weights:(D,1), bias:(1,))
`mirror=True` reproduces gpytorch's op sequence including its redundancy (Z expanded to the
input's batch shape so Kzz / Cholesky are recomputed per sample; Cholesky + solve in float64,
everything else in the working dtype) and is what bench.py times as the CPU baseline.
"""
import math
import torch
from . import kernels


def _mean(x, mean):
    """ConstantMean(batch b): const (b,1) | (1,) expanded over x's leading dims; LinearMean: x@w+b."""
    if mean[0] == 'constant':
        return mean[1].expand(*x.shape[:-1])
    _, w, b = mean
    return (x @ w).squeeze(-1) + b


def svgp_marginal(x, p, jitter=1e-4, mirror=False, full_cov=False):
    """q(f) at x for one (possibly batched) whitened SVGP layer.

    x:(...,n,D) -- already expanded to (...,b,n,D) for a b-output layer.  Returns mean (...,n),
    and var (...,n) or the full covariance (...,n,n).
    VariationalStrategy.forward [recalled]: Kzz + jitter I -> L = chol(Kzz.double());
    A = L^{-1} Kzx (double, cast back); mean = A^T m + mu(x); cov = Kxx + 1e-4 I + A^T (S - I) A.
    """
    Z, ls, os_ = p['Z'], p['lengthscale'], p['outputscale']
    dt = x.dtype
    if mirror and x.dim() > Z.dim():
        Z = Z.expand(*x.shape[:-2], *Z.shape[-2:])          # the S-fold redundancy
    M = Z.shape[-2]
    Kzz = kernels.rbf_ard(Z, Z, ls, os_) + jitter * torch.eye(M, dtype=dt)
    Kzx = kernels.rbf_ard(Z, x, ls, os_)
    L = torch.linalg.cholesky(Kzz.double())
    A = torch.linalg.solve_triangular(L, Kzx.double(), upper=False).to(dt)     # (...,M,n)
    m, Lq = p['m'], torch.tril(p['Lq'])
    mean = (A.transpose(-1, -2) @ m.unsqueeze(-1)).squeeze(-1) + _mean(x, p['mean'])
    S_minus_I = Lq @ Lq.transpose(-1, -2) - torch.eye(M, dtype=dt)
    SA = S_minus_I @ A
    if full_cov:
        n = x.shape[-2]
        Kxx = kernels.rbf_ard(x, x, ls, os_)
        return mean, Kxx + 1e-4 * torch.eye(n, dtype=dt) + A.transpose(-1, -2) @ SA
    os_b = torch.as_tensor(os_, dtype=dt)
    kxx = os_b.reshape(*os_b.shape, 1) if os_b.dim() else os_b
    var = kxx + 1e-4 + (A * SA).sum(-2)
    return mean, var


def kl_whitened(p):
    """KL(N(m, Lq Lq^T) || N(0, I)) summed over the layer's batch dims [SURVEY A.3]."""
    m, Lq = p['m'], torch.tril(p['Lq'])
    M = m.shape[-1]
    tr = (Lq * Lq).sum((-1, -2))
    logdet = 2.0 * torch.log(torch.diagonal(Lq, dim1=-1, dim2=-2).abs()).sum(-1)
    return (0.5 * (tr + (m * m).sum(-1) - M - logdet)).sum()


def dgp_forward(x, hidden, last, num_hidden_calls, eps_list, S, jitter=1e-4, mirror=False,
                full_cov_last=False):
    """DeepGP.forward (models/dgps.py:92-98) with DeepGPLayer.__call__ semantics [SURVEY A.4].

    x:(B,D).  `hidden` is the single tied hidden-layer dict applied `num_hidden_calls` times
    (models/dgps.py:88), `last` the scalar-output layer.  eps_list[k]:(S,B,b) is the standard
    normal draw used to sample the k-th MultitaskMVN (one per layer transition).
    Returns mean (S,B) and var (S,B) (or full cov (S,B,B)) of the last layer.
    """
    b = hidden['Z'].shape[0]
    h = x
    k = 0
    deterministic = True
    for _ in range(num_hidden_calls):
        xin = h.unsqueeze(-3).expand(*h.shape[:-2], b, *h.shape[-2:])           # (..., b, n, D)
        mean, var = svgp_marginal(xin, hidden, jitter, mirror)
        mean, var = mean.transpose(-1, -2), var.transpose(-1, -2)              # (..., n, b)
        if deterministic:
            mean = mean.expand(S, *mean.shape)
            var = var.expand(S, *var.shape)
            deterministic = False
        h = mean + var.sqrt() * eps_list[k]                                     # Normal(...).rsample()
        k += 1
    if deterministic:            # num_hidden_calls == 0: single-layer SVGP, expanded to S samples
        out = svgp_marginal(h, last, jitter, mirror, full_cov=full_cov_last)
        return tuple(o.expand(S, *o.shape) for o in out)
    return svgp_marginal(h, last, jitter, mirror, full_cov=full_cov_last)


def gauss_ell(y, mean, var, noise):
    """GaussianLikelihood.expected_log_prob [SURVEY A.5]: per-point, shape of mean."""
    noise = torch.as_tensor(noise, dtype=mean.dtype)
    return -0.5 * (((y - mean) ** 2 + var) / noise + torch.log(noise) + math.log(2 * math.pi))


def dsvi_elbo(x, y, hidden, last, num_hidden_calls, eps_list, S, noise, num_data,
              jitter=1e-4, mirror=False):
    """DeepApproximateMLL(VariationalELBO(likelihood, model, num_data))(model(x), y).

    = mean_s[ sum_i ELL_{s,i} / B ] - KL / num_data, tied layers counted once [SURVEY A.5];
    experiments/deepgp_spatial_bench.py:61,84-88.
    """
    mean, var = dgp_forward(x, hidden, last, num_hidden_calls, eps_list, S, jitter, mirror)
    B = x.shape[-2]
    ell = gauss_ell(y, mean, var, noise).sum(-1) / B                            # (S,)
    kl = kl_whitened(last)
    if num_hidden_calls > 0:
        kl = kl + kl_whitened(hidden)
    return (ell - kl / num_data).mean(0)


def dgp_predict(x, y, hidden, last, num_hidden_calls, eps_list, S, noise, jitter=1e-4):
    """DeepGP.predict for one batch (models/dgps.py:100-111): means, variances (+noise), lls (S,n).

    lls = Normal(mu, sqrt(clamp_min(v + noise, 1e-8))).log_prob(y)  [log_marginal, SURVEY A.5].
    """
    mean, var = dgp_forward(x, hidden, last, num_hidden_calls, eps_list, S, jitter)
    v = (var + noise).clamp_min(1e-8)
    lls = -0.5 * ((y - mean) ** 2 / v + torch.log(v) + math.log(2 * math.pi))
    return mean, var + noise, lls


def adam_step(params, grads, state, lr=0.01, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam's update restated (experiments/deepgp_spatial_bench.py:74-76, lr 0.01)."""
    state['t'] = state.get('t', 0) + 1
    t = state['t']
    out = []
    for i, (p, g) in enumerate(zip(params, grads)):
        m = state.setdefault(('m', i), torch.zeros_like(p))
        v = state.setdefault(('v', i), torch.zeros_like(p))
        m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
        v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
        bc1 = 1 - betas[0] ** t
        bc2 = 1 - betas[1] ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        out.append(p - (lr / bc1) * m / denom)
    return out
