"""Hardware stimulus writers against text files written by the reference's own dump code
(quan_func.py weight dump, output.py, output_end2end.py; tests/golden/*.stimtxt.npz): byte-for-byte."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_fixture
import sesrq
from sesrq import stimulus as S
from sesrq.bundle import Bundle

CASES = ["sesr_x4", "nrdm_3"]


def _golden(case):
    z = np.load(os.path.join(GOLDEN, f"{case}.stimtxt.npz"), allow_pickle=False)
    return {k.replace("|", "/"): bytes(z[k]).decode() for k in z.files}


def _taps_from_fixture(fx):
    taps = {f"input{k}": fx[f"input{k}"] for k in range(6)}
    for k in range(5):
        taps[f"pe_out{k}"] = fx[f"pe_out{k}"]
        taps[f"pe_add{k}"] = fx[f"pe_add{k}"]
    return taps


@pytest.mark.parametrize("case", CASES)
def test_writers_reproduce_reference_text(case, tmp_path):
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.stim.npz"))
    b = Bundle.load(os.path.join(GOLDEN, f"{case}.stim.npz"))
    gold = _golden(case)
    files = S.write_all(b, _taps_from_fixture(fx), root=str(tmp_path / "output_txt"))
    assert len(files) == 5 + 6 + 1 + 20 + 5 + 1
    for rel, want in gold.items():
        if not rel.startswith("output/"):
            continue
        got = open(tmp_path / "output_txt" / rel[len("output/"):]).read()
        assert got == want, rel
    S.write_all(b, _taps_from_fixture(fx), root=str(tmp_path / "e2e"), end2end=True)
    for k in (0, 5):
        assert open(tmp_path / "e2e" / "input" / f"input.{k}.txt").read() == gold[f"end2end/input/input.{k}.txt"]


def test_float_to_hex_table():
    assert S.float_to_hex(-1, 8) == "ff" and S.float_to_hex(5, 8) == "05"
    assert S.float_to_hex(-131072, 18) == "20000" and S.float_to_hex(131071, 18) == "1ffff"
    assert S.float_to_hex(-524288, 20) == "80000" and S.float_to_hex(-32768, 16) == "8000"
    assert S.float_to_hex(22, np.log2(32)) == "16"


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_writers_from_device_taps(case, tmp_path):
    """Same files, but every tensor comes from the device engine's debug taps (sesrq_forward_debug)."""
    import torch
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.stim.npz"))
    b = Bundle.load(os.path.join(GOLDEN, f"{case}.stim.npz"))
    e = sesrq.Engine(b, torch.device("cuda:0"))
    res = e.forward_debug(torch.from_numpy(fx["x"]).cuda())
    r = b.pixel_shuffle
    taps = {}
    for k in range(5):
        taps[f"input{k}"] = res[f"input{k}"].cpu().numpy()
        taps[f"pe_out{k}"] = res[f"pe_out{k}"][0].cpu().numpy()
        taps[f"pe_add{k}"] = res[f"pe_add{k}"].cpu().numpy()
    q = res["q_out"].cpu().numpy()
    N, C, Ho, Wo = q.shape
    taps["input5"] = q.reshape(N, C, Ho // r, r, Wo // r, r).transpose(0, 1, 3, 5, 2, 4).reshape(N, C * r * r, Ho // r, Wo // r)
    S.write_all(b, taps, root=str(tmp_path / "output_txt"))
    gold = _golden(case)
    for rel, want in gold.items():
        if rel.startswith("output/"):
            assert open(tmp_path / "output_txt" / rel[len("output/"):]).read() == want, rel
