"""Pins the numpy oracle (oracle/sesrq_oracle.py) to the reference: every stage of every
golden fixture (produced by running the reference itself, tests/golden/make_golden.py) must
be reproduced bit-for-bit."""
import hashlib
import os

import numpy as np
import pytest

from conftest import golden_files, load_fixture, fixture_input, big_cases, big_input, GOLDEN
from oracle import sesrq_oracle as O

STAGE_FILES = [f for f in golden_files() if not f.endswith((".params.npz", "tables.npz", ".stimtxt.npz", ".anchor.npz"))]


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("path", STAGE_FILES, ids=[os.path.basename(p)[:-4] for p in STAGE_FILES])
def test_forward_matches_reference_stage_by_stage(path):
    fx, meta = load_fixture(path)
    net = O.net_from_fixture(fx)
    x = fixture_input(fx, meta)
    st = O.forward(net, x, keep=True)
    inv = {"out": "y"}
    for name, want_sha in meta["sha"].items():
        got = st[inv.get(name, name)]
        if name.startswith("input") and name != "input4_special":
            got = got.astype(np.int8)
        assert _sha(got) == want_sha, f"{name} differs from the reference"
    # stored arrays (crops keep everything) compared element-wise too, for readable failures
    for name in fx.files:
        if name in ("meta", "x") or name.startswith(("Wq", "add_const")):
            continue
        got = st[inv.get(name, name)]
        np.testing.assert_array_equal(np.asarray(got).astype(fx[name].dtype), fx[name], err_msg=name)
    assert list(st["y"].shape) == meta["out_shape"]


def test_c_oracle_matches_the_reference_at_config2_size():
    """BASELINE config 2 at full size: the reference's own sim path was run on bench.py's pool frame 0 (1x3x1080x1920, the seeded
    random-init x2 net) in the build container (make_golden.py --case time_x2_1080p); the C oracle reproduces the SHA-256 of its int8
    and fp32 4K frames."""
    import json
    import torch
    from oracle import c_oracle as CO
    ref = json.load(open(os.path.join(GOLDEN, "reference_x2_1080p.json")))
    fx, meta = load_fixture(os.path.join(GOLDEN, "sesr_x2_rand.crop.npz"))
    net = O.net_from_fixture(fx)
    x = torch.rand((1, 3, 1080, 1920), generator=torch.Generator().manual_seed(1), dtype=torch.float32).numpy()
    assert _sha(x) == ref["x_sha256"]
    r = CO.forward(net, x, threads=min(os.cpu_count() or 1, 8), want_f=True)
    assert list(r["q_out"].shape) == ref["out_shape"]
    assert _sha(r["q_out"]) == ref["out_q_sha256"] and _sha(r["y"]) == ref["out_f_sha256"]


BIG = big_cases()


@pytest.mark.parametrize("rec", [r for _, r in BIG], ids=[r["case"] for _, r in BIG])
def test_c_oracle_matches_the_reference_on_baseline_size_natural_frames(rec):
    """Round 5: natural-ish frames at BASELINE sizes (SESR-x2 1080p = config 2, nrdm_3 540p = config 3, SESR-x4 540p = config 4's frame)
    through the reference's own sim path, calibrated by the reference on a natural frame (zero_0 < -128 from calibration; the x2 net's
    18-bit PE clamp fires on it: the reference printed max_overflow).  The C oracle reproduces the SHA-256 of the int8 and fp32 results."""
    from oracle import c_oracle as CO
    fx, meta = load_fixture(os.path.join(GOLDEN, rec["bundle"]))
    assert meta["zero"][0] < -128
    net = O.net_from_fixture(fx)
    r = CO.forward(net, big_input(rec), threads=min(os.cpu_count() or 1, 8), want_f=True)
    assert list(r["q_out"].shape) == rec["out_shape"]
    assert _sha(r["q_out"]) == rec["out_q_sha256"] and _sha(r["y"]) == rec["out_f_sha256"]


def test_qconst_table():
    t = np.load(os.path.join(GOLDEN, "tables.npz"))
    for r, M, n in zip(t["r"], t["M"], t["n"]):
        assert O.qconst(float(r)) == (int(M), int(n)), r


def test_weight_quantiser_table():
    t = np.load(os.path.join(GOLDEN, "tables.npz"))
    for w, q, s in zip(t["wq_in"], t["wq_out"], t["wq_scale"]):
        gq, gs = O.quantize_weight(w)
        assert gs == float(s)
        np.testing.assert_array_equal(gq, q)


PARAM_FILES = golden_files("*.params.npz")


@pytest.mark.parametrize("path", PARAM_FILES, ids=[os.path.basename(p)[:-11] for p in PARAM_FILES])
def test_parameter_derivation(path):
    """float collapsed convs + calibration (min,max) -> the integer bundle the reference wrote."""
    p, pm = load_fixture(path)
    fx, meta = load_fixture(path.replace(".params.npz", ".crop.npz"))
    sz = [O.calib_scale_zero(0.0 if i == 5 else pm["min"][i], pm["max"][i]) for i in range(6)]
    assert [s for s, _ in sz] == pm["scale"] == meta["scale"]
    assert [z for _, z in sz] == pm["zero"] == meta["zero"]
    ps = {5: 4, 6: 2, 3: 1}[pm["mflag"]]
    net = O.derive_net([p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)], pm["scale"], pm["zero"], ps)
    for k in range(5):
        np.testing.assert_array_equal(net.layers[k].wq, fx[f"Wq{k}"], err_msg=f"Wq{k}")
        np.testing.assert_array_equal(net.layers[k].add_const, fx[f"add_const{k}"], err_msg=f"add_const{k}")
        assert (net.layers[k].M, net.layers[k].n) == (meta["M"][k], meta["n"][k])
    assert (net.M_res, net.n_res) == (meta["M_res"], meta["n_res"])


@pytest.mark.parametrize("path", STAGE_FILES, ids=[os.path.basename(p)[:-4] for p in STAGE_FILES])
def test_c_oracle_matches_reference(path):
    """oracle/sesrq_oracle.c (the timed CPU baseline 'port') against the same golden vectors."""
    from oracle import c_oracle as CO
    fx, meta = load_fixture(path)
    net = O.net_from_fixture(fx)
    r = CO.forward(net, fixture_input(fx, meta), keep=True)
    for k in range(5):
        for nm in (f"input{k}", f"pe_out{k}", f"pe_add{k}"):
            assert _sha(r[nm]) == meta["sha"][nm], nm
    assert _sha(r["y"]) == meta["sha"]["out"]
    assert _sha(O.forward(net, fixture_input(fx, meta))["q_out"]) == _sha(r["q_out"])


def test_c_oracle_vs_numpy_oracle_batches_and_depth():
    from oracle import c_oracle as CO
    rng = np.random.default_rng(5)
    for kind, nb, hard in (("sesr_x2", 3, True), ("nrdm", 6, False), ("sesr_x4", 3, True)):
        net = O.synth_net(kind, 3, n_blocks=nb, hard=hard)
        x = rng.random((2, net.layers[0].wq.shape[1], 19, 23), dtype=np.float32)
        a, b = O.forward(net, x), CO.forward(net, x, threads=2)
        np.testing.assert_array_equal(a["q_out"], b["q_out"])
        np.testing.assert_array_equal(a["y"], b["y"])
