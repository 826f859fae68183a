"""Constants with the reference's names (utils/config.py:10-20)."""
from pathlib import Path

import torch

import nsgp.gp as gpytorch

TORCH_VERSION = torch.__version__
GPYTORCH_VERSION = gpytorch.__version__

AVAILABLE_GPU = torch.cuda.device_count()
GPU_ACTIVE = bool(AVAILABLE_GPU)
EPSILON = 1e-5
BASE_SEED = 173

BASE_PATH = Path(__file__).resolve().parent.parent.parent
RESULTS_DIR = BASE_PATH / 'results'
DATASET_DIR = BASE_PATH / 'tests' / 'golden' / 'data'      # the bundled uib_* / khyber_* CSVs
