import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")
WEIGHTS = os.path.join(REPO, "weights")
MODELS = ["imdn_baseline", "rfdn_baseline", "team04_rlfn", "team18_bsrn"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_sd_numpy(name):
    from safetensors.numpy import load_file
    return load_file(os.path.join(WEIGHTS, name + ".safetensors"))


def load_sd_torch(name, device="cpu"):
    from safetensors.torch import load_file
    return load_file(os.path.join(WEIGHTS, name + ".safetensors"), device=device)


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


def rel_err(y, ref, data_range):
    """max |y - ref| / data_range -- the parity measure of SURVEY.md section 8c."""
    return float(np.max(np.abs(np.asarray(y, np.float64) - np.asarray(ref, np.float64))) / data_range)
