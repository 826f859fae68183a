"""Run-wide constants under the names the reference's scripts import from `utils.config`
(`from utils.config import BASE_SEED, EPSILON, DATASET_DIR`, experiments/deepgp_spatial_bench.py:18 and
experiments/spatial_exp.py:28): seeds and tolerances, library versions, device census, data / results locations.
The bundled CSVs live under tests/golden/data/ in this repository (they double as test fixtures)."""
import pathlib

import torch

import nsgp.gp as _gp

__all__ = ['BASE_SEED', 'EPSILON', 'TORCH_VERSION', 'GPYTORCH_VERSION', 'AVAILABLE_GPU', 'GPU_ACTIVE', 'BASE_PATH',
           'RESULTS_DIR', 'DATASET_DIR']

# reproducibility / numerics
BASE_SEED: int = 173          # split i of an experiment uses BASE_SEED + i
EPSILON: float = 1e-5         # passed to gpytorch.settings.cholesky_jitter by the scripts


def _repo_root() -> pathlib.Path:
    here = pathlib.Path(__file__).resolve()
    return here.parents[2]    # <repo>/nonstationary-precip_amd/utils/config.py -> <repo>


# locations
BASE_PATH = _repo_root()
DATASET_DIR = BASE_PATH.joinpath('tests', 'golden', 'data')      # uib_spatial.csv, uib_spatio_temporal.csv, khyber_time_series.csv
RESULTS_DIR = BASE_PATH.joinpath('results')

# environment census
AVAILABLE_GPU: int = torch.cuda.device_count()
GPU_ACTIVE: bool = AVAILABLE_GPU > 0
TORCH_VERSION: str = torch.__version__
GPYTORCH_VERSION: str = _gp.__version__      # the nsgp.gp namespace stands in for gpytorch
