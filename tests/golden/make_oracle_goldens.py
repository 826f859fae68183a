#!/usr/bin/env python3
"""Golden fixtures produced by the CPU ORACLE (not by the reference: its models cannot be imported here, SURVEY 8c,
so these pin the oracle against drift and give the GPU tests committed vectors to meet -- parity stays "unpinned"
with respect to gpytorch itself).

    python tests/golden/make_oracle_goldens.py          # rewrites tests/golden/oracle_cfg2_gibbs.npz, oracle_dgp.npz

cfg2: data/uib_spatial.csv, float64, z-scored, seed-173 shuffle, 316 train / 78 test
(experiments/spatial_exp.py:112-150); fixed hyper-parameters of spatial_exp.py:76-80 (prior outputscale 1,
lengthscale 1.3, mean log 0.3; noise 0.011, outputscale 0.644); a fixed random log-lengthscale field.  Stored:
sampled entries + checksums of K, the MLL value, dMLL/dlog ell, predict() mean / variance at the 78 test points.
DGP: DeepGP(num_layers=1) state (Z, hypers, q(u)) drawn from a seeded generator, a 315-point batch of the same CSV,
fixed eps -> ELBO, per-parameter gradients, layer-1 mean/variance."""
import math
import os
import sys

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import exact, kernels, svgp  # noqa: E402

F64 = torch.float64


def uib_spatial(seed=173):
    df = pd.read_csv(os.path.join(HERE, 'data', 'uib_spatial.csv'), dtype=np.float64)
    arr = torch.tensor(np.array(df)).double()
    x, y = arr[:, 0:2], arr[:, -1]
    stdx, meanx = torch.std_mean(x, dim=-2)
    stdy, meany = torch.std_mean(y)
    xn, yn = (x - meanx) / stdx, (y - meany) / stdy
    rng = np.random.default_rng(seed)
    idx = np.arange(y.shape[0])
    rng.shuffle(idx)
    ntr = math.ceil(0.8 * y.shape[0])
    return xn[idx[:ntr]], yn[idx[:ntr]], xn[idx[ntr:]], yn[idx[ntr:]]


def cfg2():
    xtr, ytr, xte, yte = uib_spatial()
    prior = exact.LogNormalPrior(torch.full((2,), math.log(0.3), dtype=F64), torch.full((2, 2), 1.3, dtype=F64),
                                 torch.ones(2, dtype=F64))
    g = torch.Generator().manual_seed(2024)
    log_ell = (0.2 * torch.randn(2, len(xtr), generator=g, dtype=F64) + math.log(0.3)).requires_grad_()
    os_, noise = 0.644, 0.011
    mll = exact.gibbs_exact_mll(xtr, ytr, log_ell, os_, noise, prior)
    (grad,) = torch.autograd.grad(mll, log_ell)
    with torch.no_grad():
        K = os_ * kernels.gibbs(xtr, xtr, torch.exp(log_ell), torch.exp(log_ell))
        mu, sigma, ell2 = exact.gibbs_exact_predict(xtr, ytr, log_ell, os_, noise, prior, xte)
    ii = torch.tensor([0, 1, 17, 100, 200, 315])
    return dict(log_ell=log_ell.detach().numpy(), K_sample=K[ii][:, ii].numpy(), K_sum=float(K.sum()),
                K_fro=float((K ** 2).sum().sqrt()), K_trace=float(torch.trace(K)), mll=float(mll.detach()),
                grad_log_ell=grad.numpy(), pred_mean=mu.numpy(), pred_var=torch.diagonal(sigma).numpy(),
                ell_test=ell2.numpy(), outputscale=os_, noise=noise)


def dgp():
    xtr, ytr, _, _ = uib_spatial()
    B, D, M, S, N = 315, 2, 40, 3, 316
    g = torch.Generator().manual_seed(77)
    x, y = xtr[:B], ytr[:B]

    def layer(b, Din, linear):
        shp = (b,) if b else ()
        p = dict(Z=torch.randn(*shp, M, Din, generator=g, dtype=F64),
                 lengthscale=torch.rand(*shp, 1, Din, generator=g, dtype=F64) + 0.6,
                 outputscale=torch.rand(shp, generator=g, dtype=F64) + 0.5,
                 m=0.3 * torch.randn(*shp, M, generator=g, dtype=F64),
                 Lq=torch.tril(0.1 * torch.randn(*shp, M, M, generator=g, dtype=F64)) + torch.eye(M, dtype=F64))
        if linear:
            p['mean'] = ('linear', torch.randn(Din, 1, generator=g, dtype=F64), torch.randn(1, generator=g, dtype=F64))
        else:
            p['mean'] = ('constant', torch.zeros(1, dtype=F64))
        return p
    hidden, last = layer(2, D, True), layer(0, 2, False)
    eps = torch.randn(S, B, 2, generator=g, dtype=F64)
    noise = torch.tensor(0.25, dtype=F64)
    names = ['Z', 'lengthscale', 'outputscale', 'm', 'Lq']
    leaves = []
    for p in (hidden, last):
        for k in names:
            p[k] = p[k].clone().requires_grad_()
            leaves.append(p[k])
    elbo = svgp.dsvi_elbo(x, y, hidden, last, 1, [eps], S, noise, N)
    grads = torch.autograd.grad(elbo, leaves)
    out = dict(x=x.numpy(), y=y.numpy(), eps=eps.numpy(), noise=float(noise), num_data=N, elbo=float(elbo.detach()),
               h_w=hidden['mean'][1].numpy(), h_b=hidden['mean'][2].numpy())
    for tag, p, gs in (('h', hidden, grads[:5]), ('l', last, grads[5:])):
        for k, gk in zip(names, gs):
            out[f'{tag}_{k}'] = p[k].detach().numpy()
            out[f'{tag}_grad_{k}'] = (torch.tril(gk) if k == 'Lq' else gk).numpy()
    with torch.no_grad():
        m1, v1 = svgp.svgp_marginal(x.unsqueeze(0).expand(2, B, D), {k: hidden[k].detach() for k in names} |
                                    {'mean': hidden['mean']})
    out['h_mean'], out['h_var'] = m1.numpy(), v1.numpy()
    return out


if __name__ == '__main__':
    np.savez(os.path.join(HERE, 'oracle_cfg2_gibbs.npz'), **cfg2())
    np.savez(os.path.join(HERE, 'oracle_dgp.npz'), **dgp())
    print('wrote oracle_cfg2_gibbs.npz, oracle_dgp.npz')
