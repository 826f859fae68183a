#!/usr/bin/env python3
"""Spatio-temporal additive GP on data/uib_spatio_temporal.csv -- the flow of the reference's
experiments/spatio_temporal_exp.py:36-182: the 215-point subset (year 2000, months 1-5; months 1-4 train,
month 5 test), z-scored inputs/targets, SpatioTemporal_Stationary (exact, `--M` k-means inducing points for
SGPR) or SparseSpatioTemporal_Nonstationary (`--model Non-Stationary`), Adam(lr=0.015) on
-ExactMarginalLogLikelihood for 500 iterations, then likelihood(model(x_test)) [stationary] or
likelihood(model.predict(x_test)) [non-stationary], rmse * stdy and the Gaussian NLPD of utils/metrics.py.

    python examples/spatio_temporal.py --iters 300
"""
import argparse
import math

import _path  # noqa: F401
import numpy as np
import pandas as pd
import torch

from models.gibbs_kernels import LogNormalPriorProcess          # registers nsgp.gp as `gpytorch` if needed
from models.spatio_temporal_models import SparseSpatioTemporal_Nonstationary, SpatioTemporal_Stationary
import gpytorch                                                  # noqa: E402
from utils.config import DATASET_DIR                             # noqa: E402
from utils.metrics import negative_log_predictive_density, rmse  # noqa: E402


def load_train_test(csv):
    data = pd.read_csv(csv)
    data = data[data['time'] < 2001].copy()
    data['month'] = data['time'].rank(method='dense').astype('int')
    sub = data[data['month'] < 6]
    x = torch.Tensor(np.array(sub))[:, 1:4]
    y = torch.Tensor(np.array(sub)[:, -2])
    stdx, meanx = torch.std_mean(x, dim=-2)
    stdy, meany = torch.std_mean(y)
    xn, yn = (x - meanx) / stdx, (y - meany) / stdy
    k = int((sub['month'] < 5).sum())
    return xn[:k], yn[:k], xn[k:], yn[k:], stdy


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--csv', default=str(DATASET_DIR / 'uib_spatio_temporal.csv'))
    ap.add_argument('--model', choices=('Stationary', 'Non-Stationary'), default='Stationary')
    ap.add_argument('--iters', type=int, default=500)
    ap.add_argument('--M', type=int, default=0, help='inducing points (0: exact GP; Non-Stationary needs M > 0)')
    ap.add_argument('--lr', type=float, default=0.015)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('examples/spatio_temporal.py needs the MI355X: nsgp has no CPU path')
    x_train, y_train, x_test, y_test, stdy = load_train_test(args.csv)
    z = None
    if args.M > 0 or args.model == 'Non-Stationary':
        from sklearn.cluster import KMeans
        m = args.M if args.M > 0 else 50
        z = torch.tensor(KMeans(m, n_init=1, random_state=173).fit(x_train.numpy()).cluster_centers_, dtype=x_train.dtype)
    likelihood = gpytorch.likelihoods.GaussianLikelihood()
    if args.model == 'Stationary':
        model = SpatioTemporal_Stationary(x_train, y_train, likelihood, z)
    else:
        prior = LogNormalPriorProcess(input_dim=2, active_dims=(0, 1))
        prior.covar_module.outputscale = torch.ones_like(prior.covar_module.outputscale)
        prior.covar_module.base_kernel.lengthscale = 1.3 * torch.ones_like(prior.covar_module.base_kernel.lengthscale)
        prior.mean_module.constant = torch.nn.Parameter(math.log(0.3) * torch.ones_like(prior.mean_module.constant))
        for p in prior.parameters():
            p.requires_grad = False
        model = SparseSpatioTemporal_Nonstationary(x_train, y_train, likelihood, prior, z, num_dim=2)
    model = model.cuda()
    x_test, y_test = x_test.cuda(), y_test.cuda()
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=args.lr)
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)
    xd, yd = model.train_inputs[0], model.train_targets
    for i in range(args.iters):
        optimizer.zero_grad()
        with gpytorch.settings.max_cg_iterations(4000):
            loss = -mll(model(xd), yd)
        loss.backward()
        if i % 50 == 0:
            print('Iter %d/%d - Loss: %.3f  noise: %.3f' % (i + 1, args.iters, loss.item(), model.likelihood.noise.item()),
                  flush=True)
        optimizer.step()
    model.eval()
    likelihood.eval()
    with torch.no_grad():
        pred = likelihood(model.predict(x_test)) if args.model == 'Non-Stationary' else likelihood(model(x_test))
        y_mean = pred.loc
        y_std = pred.covariance_matrix.diag().clamp_min(1e-12).sqrt()
    print('RMSE test =  %.4f' % float(rmse(y_mean, y_test, stdy.cuda())))
    print('NLPD test = %.4f' % float(negative_log_predictive_density(y_test, y_mean, y_std)))


if __name__ == '__main__':
    main()
