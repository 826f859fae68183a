"""Restatement of the host-side data preparation (oracle; test infrastructure only).

Follows ``utils/dataprep.py`` of the reference: download_data :9-12, whitening_transform :35-43
(unbiased std via torch.std_mean), train_test_split :45-52 (ordered first-k split).
Pinned against the reference itself by tests/golden/ref_dataprep.npz.
"""
import math
import pandas as pd
import torch


def download_data(filepath):
    """CSV -> float32 tensor of all columns (utils/dataprep.py:9-12)."""
    return torch.Tensor(pd.read_csv(filepath).values)


def whitening_transform(data):
    """z-score inputs (all but last col) and target (last col) (utils/dataprep.py:35-43)."""
    x, y = data[:, :-1], data[:, -1]
    stdx, meanx = torch.std_mean(x, dim=-2)
    stdy, meany = torch.std_mean(y)
    return (x - meanx) / stdx, (y - meany) / stdy, meanx, stdx, meany, stdy


def train_test_split(X, y, train_prop):
    """Ordered split: first floor(p*N) rows train, rest test (utils/dataprep.py:45-52)."""
    n = int(math.floor(train_prop * len(X)))
    return X[:n].contiguous(), y[:n].contiguous(), X[n:].contiguous(), y[n:].contiguous()
