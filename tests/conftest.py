import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'nonstationary-precip_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
DATA = os.path.join(GOLDEN, 'data')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu via gpurun)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def data_dir():
    return DATA


def measured(name, got, ref, rtol, atol):
    """allclose-style check that also PRINTS the achieved error next to its bound (run with -s): the worst value of
    |got - ref| / (atol + rtol |ref|) -- below 1 passes.  Tolerances in the GPU parity tests are set to ~3x what this
    printed on MI355X."""
    import torch
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    diff = (got - ref).abs()
    ratio = float((diff / (atol + rtol * ref.abs())).max())
    print(f'[measured] {name}: max|diff| {float(diff.max()):.3g}, max|ref| {float(ref.abs().max()):.3g}, '
          f'worst diff/(atol + rtol|ref|) = {ratio:.3g} at rtol {rtol:g} atol {atol:g}')
    return ratio < 1.0
