"""Same as utils.metrics except rmse is NOT rescaled by Y_std (reference utils/metrics2.py:36-38)."""
import torch

from utils.metrics import print_trainable_param_names, get_trainable_param_names, nlpd  # noqa: F401


def rmse(Y_pred_mean, Y_test, Y_std):
    return torch.sqrt(torch.mean((Y_pred_mean - Y_test) ** 2)).detach()
