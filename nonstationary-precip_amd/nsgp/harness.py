"""Run harness for the exact / sparse GP experiments, modelled on the training section of the
reference's experiments/precipitation_baselines.py:236-397:

  * the objective is evaluated and back-propagated, THEN the run stops if |loss - previous loss| < threshold
    (before the optimiser step, :281-293, :389-390);
  * the parameters with the best objective so far are written to `best.tar`
    (model state_dict, iteration, optimiser state, objective; :376-383) and the last ones to `final.tar` (:393-397);
  * a log of the objective (and named scalar hyper-parameters) is kept -- a JSON-lines file here instead of the
    reference's TensorBoard writer (tensorboard is not part of this image);
  * `freeze(...)` is the "fix this hyper-parameter" idiom of :262-270.

Host logic only (device-agnostic): the model, the objective and the optimiser come from the caller, so the same
loop drives the MI355X models (`models.*`) and -- in the CPU tests -- a plain torch module.  Checkpoints are plain
`torch.save` dictionaries of tensors; load them with `torch.load(..., weights_only=True)`."""
import json
import math
import os

import torch


def freeze(module_or_params):
    """requires_grad = False on a module's parameters (or an iterable of parameters)."""
    params = module_or_params.parameters() if hasattr(module_or_params, 'parameters') else module_or_params
    for p in params:
        p.requires_grad = False


class RunLog:
    """JSON-lines log: one object per logged iteration, `{"i": ..., "objective": ..., <scalars>}`."""

    def __init__(self, path):
        self.path = path
        self._fh = open(path, 'w') if path else None

    def write(self, i, objective, scalars=None):
        if self._fh is None:
            return
        rec = {'i': int(i), 'objective': float(objective)}
        for k, v in (scalars or {}).items():
            rec[k] = float(v)
        self._fh.write(json.dumps(rec) + '\n')
        self._fh.flush()

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


def save_checkpoint(path, model, optimizer, i, objective=None):
    state = {'model': model.state_dict(), 'i': int(i), 'optim_state': optimizer.state_dict()}
    if objective is not None:
        state['objective'] = float(objective)
    torch.save(state, path)


def load_checkpoint(path, model, optimizer=None, map_location=None):
    """Restores `model` (and `optimizer`) from a checkpoint written by this harness; returns the stored dict."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(state['model'])
    if optimizer is not None and 'optim_state' in state:
        optimizer.load_state_dict(state['optim_state'])
    return state


def fit(model, loss_fn, optimizer, max_iters=1000, threshold=1e-6, logdir=None, log_interval=1, scalars=None,
        callback=None):
    """Minimise `loss_fn()` (a closure returning the scalar objective, e.g. `-mll(model(x), y)`).

    Returns a dict(iterations, objective, best_objective, best_iteration, stopped_early).  With `logdir`, writes
    `log.jsonl`, `best.tar` and `final.tar` there.  `scalars`: optional closure -> {name: value} logged every
    `log_interval` iterations; `callback(i, loss)`: optional hook (e.g. test metrics)."""
    if logdir:
        os.makedirs(logdir, exist_ok=True)
    log = RunLog(os.path.join(logdir, 'log.jsonl') if logdir else None)
    best, best_i = math.inf, -1
    loss_val, i, early = math.inf, -1, False
    try:
        for i in range(int(max_iters)):
            optimizer.zero_grad()
            old = loss_val
            loss = loss_fn()
            loss_val = float(loss.detach())
            if not math.isfinite(loss_val):
                # a NaN/inf objective never satisfies |change| < threshold: without this check the loop would keep
                # stepping on NaN gradients until max_iters and checkpoint NaN parameters
                raise FloatingPointError(f'objective is {loss_val} at iteration {i}: stopping before the optimiser '
                                         'step (last finite parameters are in best.tar when logdir is set)')
            change = abs(old - loss_val)
            loss.backward()
            if i % max(1, int(log_interval)) == 0:
                log.write(i, loss_val, scalars() if scalars is not None else None)
            if callback is not None:
                callback(i, loss_val)
            if loss_val < best:
                best, best_i = loss_val, i
                if logdir:
                    save_checkpoint(os.path.join(logdir, 'best.tar'), model, optimizer, i, loss_val)
            if change < float(threshold):          # not making progress: stop before the step, like the reference
                early = True
                break
            optimizer.step()
        if logdir:
            save_checkpoint(os.path.join(logdir, 'final.tar'), model, optimizer, max(i, 0))
    finally:
        log.close()
    return {'iterations': i + 1, 'objective': loss_val, 'best_objective': best, 'best_iteration': best_i,
            'stopped_early': early}
