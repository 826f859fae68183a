"""RMSE / NLPD and parameter listing with the reference's names (utils/metrics.py:11-52).
`rmse` here multiplies by Y_std (metrics.py:36-38); utils.metrics2.rmse does not (metrics2.py:36-38).
The parameter table needs no prettytable (absent in this image)."""
import torch


def _table(model):
    rows = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
    w = max([len('Modules')] + [len(n) for n, _ in rows])
    lines = [f"{'Modules':<{w}} | Parameters", '-' * (w + 13)]
    lines += [f'{n:<{w}} | {k}' for n, k in rows]
    return '\n'.join(lines), sum(k for _, k in rows)


def print_trainable_param_names(model):
    text, total = _table(model)
    print(text)
    print(f'Total Trainable Params: {total}')


def get_trainable_param_names(model):
    return [n for n, p in model.named_parameters() if p.requires_grad]


def rmse(Y_pred_mean, Y_test, Y_std):
    return Y_std.item() * torch.sqrt(torch.mean((Y_pred_mean - Y_test) ** 2)).detach()


def nlpd(Y_test_pred, Y_test, Y_std):
    lpd = Y_test_pred.log_prob(Y_test)
    return -(lpd.detach() / len(Y_test) - torch.log(torch.as_tensor(Y_std)))


def negative_log_predictive_density(test_y, predicted_mean, predicted_var):
    lpd = torch.distributions.Normal(predicted_mean, torch.sqrt(predicted_var)).log_prob(test_y)
    return -torch.mean(lpd)
