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


def big_cases():
    """[(json path, record)] of the BASELINE-size natural frames the reference itself ran (make_golden.py *_nat cases)."""
    return [(p, json.load(open(p))) for p in sorted(glob.glob(os.path.join(GOLDEN, "*.big.json")))]


def big_input(rec):
    import hashlib
    sys.path.insert(0, GOLDEN)
    from natural import natural_frame
    x = natural_frame(rec["in_shape"][1], rec["in_shape"][2], rec["in_shape"][3], rec["nat_seed"])
    assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == rec["x_sha256"], "natural frame differs from the one the reference ran on"
    return x


def fixture_input(fx, meta):
    """The fp32 input of a fixture: stored inline for crops, shared .npy for full frames."""
    if fx["x"].size:
        return fx["x"]
    if meta.get("input") == "natural":      # regenerated (tests/golden/natural.py is pure IEEE arithmetic + a seeded torch generator) and checked
        import hashlib
        sys.path.insert(0, GOLDEN)
        from natural import natural_frame
        x = natural_frame(1 if meta["mflag"] == 5 else 3, meta["H"], meta["W"], meta["nat_seed"])
        assert hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest() == meta["x_sha256"], "natural frame differs from the one the reference ran on"
        return x
    name = "rand_SR_Input_80x960.npy" if meta["mflag"] == 5 else "rand_DM_Input_80x960.npy"
    return np.load(os.path.join(GOLDEN, name))
