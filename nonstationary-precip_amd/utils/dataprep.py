"""Host-side data preparation with the reference's semantics (utils/dataprep.py:9-52): CSV -> float32
tensor, z-scoring with the UNBIASED std (torch.std_mean), Box-Cox, ordered first-k train/test split."""
import math

import pandas as pd
import scipy as sp
import scipy.stats
import torch


def download_data(filepath):
    return torch.Tensor(pd.read_csv(filepath).values)


def prep_inputs(data):
    x = data[:, :-1]
    stdx, meanx = torch.std_mean(x, dim=-2)
    return (x - meanx) / stdx


def prep_outputs(data):
    y_tr, bc_param = sp.stats.boxcox(data[:, -1])
    return y_tr, bc_param


def box_cox_transform(data):
    return prep_inputs(data), prep_outputs(data)


def whitening_transform(data):
    x, y = data[:, :-1], data[:, -1]
    stdx, meanx = torch.std_mean(x, dim=-2)
    stdy, meany = torch.std_mean(y)
    return (x - meanx) / stdx, (y - meany) / stdy, meanx, stdx, meany, stdy


def train_test_split(X, y, train_prop):
    n = int(math.floor(train_prop * len(X)))
    return X[:n, :].contiguous(), y[:n].contiguous(), X[n:, :].contiguous(), y[n:].contiguous()
