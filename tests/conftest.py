"""pytest configuration: markers, import paths, fixture helpers."""
import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "sesr-pytorch-quantize_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(pattern="*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN, pattern)))


def load_fixture(path):
    fx = np.load(path, allow_pickle=False)
    meta = json.loads(str(fx["meta"]))
    return fx, meta


def fixture_input(fx, meta):
    """The fp32 input of a fixture: stored inline for crops, shared .npy for full frames."""
    if fx["x"].size:
        return fx["x"]
    name = "rand_SR_Input_80x960.npy" if meta["mflag"] == 5 else "rand_DM_Input_80x960.npy"
    return np.load(os.path.join(GOLDEN, name))
