"""The callers either side of the hot path: examples/deepgp_spatial.py and
examples/gibbs_spatial.py drive the drop-in `models.*` / `utils.*` / `gpytorch` surface exactly like the
reference's experiments/deepgp_spatial_bench.py and experiments/spatial_exp.py.  Short runs on the bundled
uib_spatial.csv; the bands are sanity bands (the reference's results/*.csv come from unseeded 400-epoch runs:
trained 2-layer DGP RMSE 0.5-0.6 in raw units), not parity assertions."""
import importlib.util
import math
import os
import sys
import types

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    ex = os.path.join(ROOT, 'examples')
    if ex not in sys.path:
        sys.path.insert(0, ex)
    spec = importlib.util.spec_from_file_location(name, os.path.join(ex, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_deepgp_spatial_example_trains_and_predicts(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    mod = _load('deepgp_spatial')
    import utils.dataprep as dp
    from utils.config import DATASET_DIR
    args = types.SimpleNamespace(layers=1, inducing=64, batch=315, lr=0.01, epochs=120, samples=3,
                                 verbose=False, out=str(tmp_path / 'pred.csv'))
    dataset = dp.download_data(str(DATASET_DIR / 'uib_spatial.csv'))
    rm, nl, loss, frame = mod.run_split(dataset, 0, args, torch.device('cuda', 0))
    assert list(frame.columns) == ['pred', 'std', 'lat', 'lon'] and len(frame) == 79
    # whitened-target ELBO loss starts near 1.6; a trained model is well below 1.2 and predicts better than the
    # target's standard deviation (raw std of tp on this CSV is ~1.1): sanity band only
    assert loss < 1.2
    assert 0.2 < rm < 0.9
    assert abs(nl) < 5.0


def test_trained_dgp_agrees_with_the_predictions_the_reference_saved(tmp_path):
    """results/f_mean_sigma_dgp2.csv is the one model output the reference repository holds: a 2-layer DGP's predictive
    mean / std at the 394 points of uib_spatial.csv (copied as data to tests/golden/ref_results/; unseeded 400-epoch
    run, so no entry-wise parity).  Against tp it has RMSE 0.554 and correlation 0.92.  The same recipe on this
    path (2-layer DGP, 250 inducing points, Adam 0.01, minibatch 315) must land in that band AND reproduce the saved
    field itself: correlation > 0.9 and RMS difference < 0.55 between the two prediction maps."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import numpy as np
    import pandas as pd
    mod = _load('deepgp_spatial')
    import utils.dataprep as dp
    from utils.config import DATASET_DIR
    args = types.SimpleNamespace(layers=1, inducing=250, batch=315, lr=0.01, epochs=300, samples=3, verbose=False,
                                 out=None, predict_all=True)
    dataset = dp.download_data(str(DATASET_DIR / 'uib_spatial.csv'))
    mod.run_split(dataset, 0, args, torch.device('cuda', 0))
    ours = args.frame_all
    ref = pd.read_csv(os.path.join(ROOT, 'tests', 'golden', 'ref_results', 'f_mean_sigma_dgp2.csv'), index_col=0)
    raw = pd.read_csv(str(DATASET_DIR / 'uib_spatial.csv'))
    if ours['lat'].mean() > 60:                                # the example labels the columns by position like the reference
        ours = ours.rename(columns={'lat': 'lon', 'lon': 'lat'})    # (dataprep keeps the CSV order lon, lat)
    key = lambda f: f.assign(lat=f['lat'].astype(float).round(2), lon=f['lon'].astype(float).round(2))
    m = key(ours).merge(key(ref), on=['lat', 'lon'], suffixes=('', '_ref')).merge(key(raw), on=['lat', 'lon'])
    assert len(m) == 394
    rm_ours = float(np.sqrt(((m['pred'] - m['tp']) ** 2).mean()))
    rm_ref = float(np.sqrt(((m['pred_ref'] - m['tp']) ** 2).mean()))
    assert abs(rm_ref - 0.5537) < 1e-3                       # the fixture is the file SURVEY 8d quotes
    assert 0.3 < rm_ours < 0.75, rm_ours                     # the reference's own run: 0.554
    corr = float(np.corrcoef(m['pred'], m['pred_ref'])[0, 1])
    diff = float(np.sqrt(((m['pred'] - m['pred_ref']) ** 2).mean()))
    assert corr > 0.9, corr
    assert diff < 0.55, diff


def test_gibbs_spatial_example_runs_exact_and_sparse(capsys, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    mod = _load('gibbs_spatial')
    for extra in ([], ['--inference', 'sparse', '--M', '64']):
        monkeypatch.setattr(sys, 'argv', ['gibbs_spatial.py', '--splits', '1', '--iters', '60'] + extra)
        mod.main()
        out = capsys.readouterr().out
        line = [l for l in out.splitlines() if l.startswith('split 0:')][0]
        rm = float(line.split('RMSE test =')[1].split()[0])
        assert 0.1 < rm < 1.0, line


def test_temporal_example_runs(capsys, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    mod = _load('temporal')
    monkeypatch.setattr(sys, 'argv', ['temporal.py', '--iters', '40'])
    mod.main()
    out = capsys.readouterr().out
    assert 'RMSE test' in out and 'NLPD test' in out
    rm = float([l for l in out.splitlines() if l.startswith('RMSE test')][0].split('=')[1])
    assert math.isfinite(rm) and rm > 0
