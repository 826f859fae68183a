#!/usr/bin/env python3
"""DSVI deep-GP regression on data/uib_spatial.csv through the reference's class surface.

The caller side of the hot path: the flow of the reference's
experiments/deepgp_spatial_bench.py:45-117 -- sklearn shuffle(random_state=r), whitening
(utils/dataprep.py:35-43), first-80 % split (dataprep.py:45-52), DeepGP(num_layers, train_x.shape),
DeepApproximateMLL(VariationalELBO(likelihood, model, N_train)), Adam(lr=0.01) over all parameters,
minibatches of 315, S likelihood samples, then model.predict() and rmse*stdy / nlpd
(utils/metrics.py:36-45) -- written against this repo's drop-in packages.  Prints one line per split and
the mean +- standard error over splits; `--out` writes the `,pred,std,lat,lon`-style prediction CSV of
the last split (the layout of the reference's results/*.csv).

    python examples/deepgp_spatial.py --splits 2 --epochs 100 --layers 1 --samples 3
"""
import argparse
import math

import _path  # noqa: F401
import numpy as np
import pandas as pd
import torch
from sklearn.utils import shuffle
from torch.utils.data import DataLoader, TensorDataset

import models.dgps as m                 # importing `models` registers nsgp.gp as `gpytorch` if the real one is absent
import gpytorch                         # noqa: E402
import utils.dataprep as dp             # noqa: E402
from gpytorch.mlls import DeepApproximateMLL, VariationalELBO      # noqa: E402
from utils.config import DATASET_DIR
from utils.metrics import nlpd, rmse


def run_split(dataset, random_state, args, device):
    data = shuffle(dataset, random_state=random_state)
    x_tr, y_tr, meanx, stdx, meany, stdy = dp.whitening_transform(data)
    train_x, train_y, test_x, test_y = dp.train_test_split(x_tr, y_tr, 0.8)
    torch.manual_seed(random_state)
    model = m.DeepGP(args.layers, train_x.shape, num_inducing=args.inducing).to(device)
    mll = DeepApproximateMLL(VariationalELBO(model.likelihood, model, train_x.shape[-2]))
    train_x, train_y, test_x, test_y = (t.to(device) for t in (train_x, train_y, test_x, test_y))
    loader = DataLoader(TensorDataset(train_x, train_y), batch_size=args.batch, shuffle=True)
    model.train()
    optimizer = torch.optim.Adam([{'params': model.parameters()}], lr=args.lr)
    loss = None
    for epoch in range(args.epochs):
        for x_batch, y_batch in loader:
            with gpytorch.settings.num_likelihood_samples(args.samples):
                optimizer.zero_grad()
                loss = -mll(model(x_batch), y_batch)
                loss.backward()
                optimizer.step()
        if args.verbose and epoch % max(1, args.epochs // 10) == 0:
            print(f'  split {random_state} epoch {epoch:4d} loss {float(loss):.4f}', flush=True)
    model.eval()
    with torch.no_grad(), gpytorch.settings.num_likelihood_samples(args.samples):
        pred_y, y_means, y_var, test_lls = model.predict(DataLoader(TensorDataset(test_x, test_y), batch_size=args.batch))
    rmse_test = float(rmse(y_means, test_y, stdy.to(device)))
    nlpd_test = float(nlpd(pred_y, test_y, stdy.to(device)).mean())
    frame = None
    if args.out:
        raw_x = (test_x.cpu() * stdx + meanx).numpy()
        frame = pd.DataFrame({'pred': (y_means.mean(0).cpu() * stdy + meany).numpy(),
                              'std': (y_var.mean(0).sqrt().cpu() * stdy).numpy(),
                              'lat': raw_x[:, 0], 'lon': raw_x[:, 1]})
    if getattr(args, 'predict_all', False):
        # predictions at every row of the CSV in the reference's results/*.csv layout (index, pred, std, lat, lon):
        # what experiments/spatial_exp.py:252 reads back as results/f_mean_sigma_dgp2.csv
        all_x = x_tr.to(device)
        with torch.no_grad(), gpytorch.settings.num_likelihood_samples(args.samples):
            _, a_means, a_var, _ = model.predict(DataLoader(TensorDataset(all_x, y_tr.to(device)), batch_size=args.batch))
        raw_all = (x_tr * stdx + meanx).numpy()
        args.frame_all = pd.DataFrame({'pred': (a_means.mean(0).cpu() * stdy + meany).numpy(),
                                       'std': (a_var.mean(0).sqrt().cpu() * stdy).numpy(),
                                       'lat': raw_all[:, 0], 'lon': raw_all[:, 1]})
    return rmse_test, nlpd_test, float(loss.detach()), frame


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--csv', default=str(DATASET_DIR / 'uib_spatial.csv'))
    ap.add_argument('--splits', type=int, default=10)
    ap.add_argument('--epochs', type=int, default=400)
    ap.add_argument('--layers', type=int, default=4)
    ap.add_argument('--samples', type=int, default=3)
    ap.add_argument('--inducing', type=int, default=250)
    ap.add_argument('--batch', type=int, default=315)
    ap.add_argument('--lr', type=float, default=0.01)
    ap.add_argument('--out', default=None)
    ap.add_argument('--verbose', action='store_true')
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit('examples/deepgp_spatial.py needs the MI355X: nsgp has no CPU path')
    device = torch.device('cuda', 0)
    dataset = dp.download_data(args.csv)
    rmses, nlpds = [], []
    frame = None
    for r in range(args.splits):
        rm, nl, loss, frame = run_split(dataset, r, args, device)
        print(f'split {r}: RMSE test = {rm:.4f}  NLPD test = {nl:.4f}  final loss = {loss:.4f}', flush=True)
        rmses.append(rm)
        nlpds.append(nl)
    k = math.sqrt(max(len(rmses), 1))
    print(f'Final RMSE across splits: {np.mean(rmses):.4f} +- {np.std(rmses) / k:.4f}')
    print(f'Final NLPD across splits: {np.mean(nlpds):.4f} +- {np.std(nlpds) / k:.4f}')
    if args.out and frame is not None:
        frame.to_csv(args.out)
        print('wrote', args.out)


if __name__ == '__main__':
    main()
