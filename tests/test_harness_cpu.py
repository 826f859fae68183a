"""SURVEY 8f.4 run harness (nsgp.harness): early stopping on |delta loss| < threshold BEFORE the optimiser step,
best-by-objective and final checkpoints, JSON-lines log -- the behaviour of the reference's
experiments/precipitation_baselines.py:281-293,376-397.  Host logic, exercised on a plain CPU torch module."""
import json
import os

import torch


def _problem():
    torch.manual_seed(0)
    x = torch.randn(64, 3)
    w_true = torch.tensor([1.0, -2.0, 0.5])
    y = x @ w_true + 0.01 * torch.randn(64)
    model = torch.nn.Linear(3, 1)
    return model, (lambda: ((model(x).squeeze(-1) - y) ** 2).mean())


def test_fit_logs_checkpoints_and_stops_on_small_change(tmp_path):
    from nsgp.harness import fit, load_checkpoint
    model, loss_fn = _problem()
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    seen = []
    res = fit(model, loss_fn, opt, max_iters=2000, threshold=1e-7, logdir=str(tmp_path), log_interval=10,
              scalars=lambda: {'bias': model.bias.item()}, callback=lambda i, l: seen.append(l))
    assert res['stopped_early'] and res['iterations'] < 2000
    assert res['best_objective'] <= min(seen) + 1e-12 and res['objective'] < 1e-3
    # early stop happens BEFORE the optimiser step of that iteration: last two objectives differ by < threshold
    assert abs(seen[-1] - seen[-2]) < 1e-7
    lines = [json.loads(l) for l in open(os.path.join(str(tmp_path), 'log.jsonl'))]
    assert lines[0]['i'] == 0 and lines[1]['i'] == 10 and 'bias' in lines[0]
    assert lines[0]['objective'] > lines[-1]['objective']
    # best.tar restores the best parameters and the optimiser state; tensors only (weights_only loader)
    model2, loss2 = _problem()
    opt2 = torch.optim.Adam(model2.parameters(), lr=0.05)
    state = load_checkpoint(os.path.join(str(tmp_path), 'best.tar'), model2, opt2)
    assert abs(float(loss2()) - state['objective']) < 1e-6 and state['i'] == res['best_iteration']
    final = torch.load(os.path.join(str(tmp_path), 'final.tar'), weights_only=True)
    assert set(final) == {'model', 'i', 'optim_state'} and final['i'] == res['iterations'] - 1


def test_fit_runs_to_max_iters_without_logdir_and_freeze_fixes_parameters():
    from nsgp.harness import fit, freeze
    model, loss_fn = _problem()
    freeze([model.bias])
    b0 = model.bias.detach().clone()
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05)
    res = fit(model, loss_fn, opt, max_iters=25, threshold=0.0)
    assert res['iterations'] == 25 and not res['stopped_early']
    assert torch.equal(model.bias.detach(), b0)
    assert res['best_iteration'] >= 0


def test_fit_raises_on_a_non_finite_objective_before_stepping(tmp_path):
    """a NaN loss makes `change < threshold` false forever; fit() must stop before optimizer.step() instead of
    writing NaN parameters to final.tar."""
    import pytest
    from nsgp.harness import fit
    model, loss_fn = _problem()
    calls = {'n': 0}

    def nan_after_3():
        calls['n'] += 1
        loss = loss_fn()
        return loss * float('nan') if calls['n'] > 3 else loss
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    with pytest.raises(FloatingPointError, match='objective is nan at iteration 3'):
        fit(model, nan_after_3, opt, max_iters=50, threshold=0.0, logdir=str(tmp_path))
    assert all(torch.isfinite(p).all() for p in model.parameters())
    best = torch.load(os.path.join(str(tmp_path), 'best.tar'), weights_only=True)
    assert best['i'] <= 2 and all(torch.isfinite(v).all() for v in best['model'].values())


def test_fused_adam_state_dict_round_trip_and_bucket_homing():
    """FusedAdam.state_dict()/load_state_dict() (plain tensors, in-place restore) and FlatBucket noticing a
    parameter that was moved out of the flat buffer.  CPU-level: no kernel is launched."""
    import pytest
    from nsgp.optim import FlatBucket, FusedAdam
    torch.manual_seed(1)
    lin = torch.nn.Linear(5, 3)
    opt = FusedAdam(lin.parameters(), lr=0.02, capturable=False)
    opt.exp_avg.normal_(); opt.exp_avg_sq.uniform_(); opt.steps = 17
    sd = opt.state_dict()
    assert set(sd) == {'step', 'exp_avg', 'exp_avg_sq', 'lr', 'betas', 'eps', 'numel', 'layout'}
    buf = __import__('io').BytesIO()
    torch.save(sd, buf); buf.seek(0)
    sd2 = torch.load(buf, weights_only=True)                       # loads without unpickling arbitrary objects
    lin2 = torch.nn.Linear(5, 3)
    opt2 = FusedAdam(lin2.parameters(), lr=0.5)
    ptr = opt2.exp_avg.data_ptr()
    opt2.load_state_dict(sd2)
    assert opt2.exp_avg.data_ptr() == ptr                            # in place: a captured hipGraph stays valid
    assert opt2.steps == 17 and opt2.lr == 0.02 and torch.equal(opt2.exp_avg, opt.exp_avg)
    assert torch.equal(opt2.exp_avg_sq, opt.exp_avg_sq) and opt2.param_groups[0]['lr'] == 0.02
    with pytest.raises(ValueError, match='parameters in the checkpoint'):
        FusedAdam(torch.nn.Linear(2, 2).parameters()).load_state_dict(sd2)
    bucket = FlatBucket(list(lin2.parameters()))
    bucket.check_homed()
    lin2.double()                                                    # re-allocates p.data outside the flat buffer
    with pytest.raises(RuntimeError, match='moved out of the flat buffer'):
        bucket.check_homed()
