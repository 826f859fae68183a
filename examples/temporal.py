#!/usr/bin/env python3
"""Stationary temporal GP on data/khyber_time_series.csv (BASELINE configs[0]) -- the flow of the reference's
experiments/temporal_exp.py:28-118: z-scored time, Box-Cox targets, first 80 % train (no shuffle: extrapolation),
ExactGP(ConstantMean, ScaleKernel(RBFKernel() * PeriodicKernel(), outputscale > 7)), noise initialised to 0.1,
Adam(lr=0.01) on -ExactMarginalLogLikelihood, then likelihood(model(x_test)) and rmse * stdy / nlpd
(utils/metrics.py:36-45).  The RBF x Periodic Gram matrix is one launch of the fused gfx950 build kernel.

    python examples/temporal.py --iters 500
"""
import argparse
import math

import _path  # noqa: F401
import numpy as np
import pandas as pd
import scipy.stats
import torch

import models  # noqa: F401  (registers nsgp.gp as `gpytorch` if the real one is absent)
import gpytorch                                                  # noqa: E402
from gpytorch.constraints import GreaterThan                      # noqa: E402
from gpytorch.kernels import PeriodicKernel, RBFKernel, ScaleKernel   # noqa: E402
from utils.config import DATASET_DIR                             # noqa: E402
from utils.metrics import nlpd, rmse                             # noqa: E402


class KhyberTemporalStat(gpytorch.models.ExactGP):
    def __init__(self, train_x, train_y, likelihood):
        super().__init__(train_x, train_y, likelihood)
        self.mean_module = gpytorch.means.ConstantMean()
        self.covar_module = ScaleKernel(RBFKernel() * PeriodicKernel(), outputscale_constraint=GreaterThan(7))

    def forward(self, x):
        return gpytorch.distributions.MultivariateNormal(self.mean_module(x), self.covar_module(x))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--csv', default=str(DATASET_DIR / 'khyber_time_series.csv'))
    ap.add_argument('--iters', type=int, default=2000)
    ap.add_argument('--lr', type=float, default=0.01)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('examples/temporal.py needs the MI355X: nsgp has no CPU path')
    data = pd.read_csv(args.csv)
    x = torch.Tensor(np.array(data))[:, 0]
    y = torch.Tensor(np.array(data)[:, -1])
    y_tr, _bc = scipy.stats.boxcox(y)
    stdx, meanx = torch.std_mean(x)
    x_norm = (x - meanx) / stdx
    stdy, _meany = torch.std_mean(y)
    y_norm = torch.as_tensor(y_tr, dtype=torch.float32)
    k = math.ceil(0.8 * y.shape[0])
    x_train, y_train = x_norm[:k].cuda(), y_norm[:k].cuda()
    x_test, y_test = x_norm[k:].cuda(), y_norm[k:].cuda()
    likelihood = gpytorch.likelihoods.GaussianLikelihood()
    model = KhyberTemporalStat(x_train, y_train, likelihood).cuda()
    likelihood.noise = 1e-1
    model.train()
    likelihood.train()
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr)
    mll = gpytorch.mlls.ExactMarginalLogLikelihood(likelihood, model)
    xd, yd = model.train_inputs[0], model.train_targets
    for i in range(args.iters):
        optimizer.zero_grad()
        loss = -mll(model(xd), yd)
        loss.backward()
        if i % max(1, args.iters // 4) == 0:
            print('Iter %d/%d - Loss: %.3f  noise: %.3f' % (i + 1, args.iters, loss.item(), model.likelihood.noise.item()),
                  flush=True)
        optimizer.step()
    model.eval()
    likelihood.eval()
    with torch.no_grad():
        pred = likelihood(model(x_test))
    print('RMSE test =  %.4f' % float(rmse(pred.loc, y_test, stdy.cuda())))
    print('NLPD test = %.4f' % float(nlpd(pred, y_test, stdy.cuda())))


if __name__ == '__main__':
    main()
