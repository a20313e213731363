"""One-fma requant (sesrq_layer_one_fma, csrc/sesrq_verify.hip: prove_direct_requant).

A layer that requantises into a -128 domain computes q = clamp8(rint(fl(fl(s*M) * 2^-n - 128))) (myQL/quan_func.py:280; output
layer :601).  The MFMA kernels may replace it by cvt_u8(fl(s*M) * 2^-n) - 128 -- one fused multiply-add and the saturating
byte convert -- where sesrq_create has checked, for the layer's (M, n), every accumulator value s whose result is not saturated.
Here: the claim restated in numpy (the oracle's arithmetic, no library), its analytic corner (n <= 17 can never fail), a
counter-example that shows the check is not vacuous, and -- on the GPU -- that the library's verdict equals numpy's for every
golden bundle, that the forward with the form switched off (SESRQ_DIRECT=0, a second process: the knob is read once) produces the
same bits, and that both are the oracle's."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_files, load_fixture
from helpers import bundle_from_oracle, fixture_case
from oracle import sesrq_oracle as O


def one_fma_verdict(M: int, n: int):
    """(all equal?, first s that differs or None): reference form vs one-fma form over every s with t' in [-2, 258]."""
    if M == 0 or 3 * M >= (1 << 18):
        return False, None
    scale = M * 2.0 ** -n
    lim = (1 << 22) - 1
    lo, hi = max(-lim, int(np.floor(-2.0 / scale))), min(lim, int(np.ceil(258.0 / scale)))
    first = None
    for a in range(lo, hi + 1, 1 << 20):
        s = np.arange(a, min(hi + 1, a + (1 << 20)), dtype=np.int64)
        t = (s * M).astype(np.float32)                      # fl(s * M): exact integer product, one rounding
        tp = t * np.float32(2.0 ** -n)                      # exact (power of two)
        v = tp + np.float32(-128.0)                         # fl(t' - 128): the reference's add of the zero point
        ref = np.clip(np.rint(v), -128, 127) + 128
        one = np.clip(np.rint(tp), 0, 255)
        bad = np.nonzero(ref != one)[0]
        if len(bad):
            first = int(s[bad[0]])
            break
    return first is None, first


def single_rounding_verdict(M: int, n: int) -> bool:
    """The output layer's second candidate: v = fl(s*M*2^-n - 128) with ONE rounding (exact value in float64), then + 128."""
    if M == 0 or 3 * M >= (1 << 18):
        return False
    scale = M * 2.0 ** -n
    lim = (1 << 22) - 1
    lo, hi = max(-lim, int(np.floor(-2.0 / scale))), min(lim, int(np.ceil(258.0 / scale)))
    s = np.arange(lo, hi + 1, dtype=np.int64)
    tp = (s * M).astype(np.float32) * np.float32(2.0 ** -n)
    ref = np.clip(np.rint(tp + np.float32(-128.0)), -128, 127) + 128
    exact = (s * M).astype(np.float64) * 2.0 ** -n - 128.0          # < 2^40 * 2^-n: exact in float64
    one = np.clip(np.rint(exact.astype(np.float32) + np.float32(128.0)), 0, 255)
    return bool(np.all(ref == one))


def test_shift_up_to_17_can_never_fail():
    """t' < 64 lies on the 2^-n grid (s*M < 2^23 is exact), so t' - 128 is exact for n <= 17: the two forms are the same number."""
    rng = np.random.default_rng(5)
    for n in (8, 12, 16, 17):
        for M in rng.integers(1 << 10, (1 << 16) - 1, 6):
            ok, first = one_fma_verdict(int(M), n)
            assert ok, (M, n, first)


def test_the_check_is_not_vacuous():
    """A (M, n) for which the forms differ exists (t' = k + 1/2 +- 2^-18 below 64: fl(t' - 128) lands ON the tie)."""
    found = None
    for M in range(40000, 40400):
        ok, first = one_fma_verdict(M, 24)
        if not ok:
            found = (M, first)
            break
    assert found is not None
    M, s = found
    t = np.float32(np.int64(s) * M)
    tp = t * np.float32(2.0 ** -24)
    assert tp < 64 and abs((float(tp) % 1.0) - 0.5) <= 2.0 ** -18 and (float(tp) % 1.0) != 0.5, (M, s, float(tp))


PARAM_FILES = [f for f in golden_files("*.npz") if not f.endswith((".params.npz", "tables.npz", ".stimtxt.npz", ".anchor.npz"))]


def test_host_proof_equals_numpy_restatement():
    """sesrq_requant_form (the proof sesrq_create runs; a host function, no device) against the numpy restatement: every (M, n)
    of the golden bundles, the counter-example range, and a seeded sample of 16-bit multipliers with the shifts calibration gives."""
    import sesrq
    pairs = set()
    for path in PARAM_FILES:
        fx, meta = load_fixture(path)
        net = O.net_from_fixture(fx)
        pairs |= {(l.M, l.n) for l in net.layers}
    pairs |= {(M, 24) for M in range(40000, 40040)}
    rng = np.random.default_rng(11)
    pairs |= {(int(rng.integers(1 << 15, 1 << 16)), int(rng.integers(18, 27))) for _ in range(60)}
    forms = {0: 0, 1: 0, 2: 0}
    for M, n in sorted(pairs):
        one = one_fma_verdict(M, n)[0]
        want_hidden = 1 if one else 0
        want_out = 1 if one else (2 if single_rounding_verdict(M, n) else 0)
        assert sesrq.requant_form(M, n) == want_hidden, (M, n)
        assert sesrq.requant_form(M, n, output_layer=True) == want_out, (M, n)
        forms[want_out] += 1
    assert forms[1] > 0 and forms[2] > 0, forms
    assert sesrq.requant_form(0, 20) == 0 and sesrq.requant_form(1 << 17, 20) == 0          # outside the biased form's own range


def expected_flags(net: O.Net):
    L = len(net.layers)
    out = []
    for k, l in enumerate(net.layers):
        zt = net.zero[L] if k == L - 1 else net.zero[1 if k == 0 else k + 1]
        # the residual-merging layer L-2: its first requant goes into the fixed -128 domain of ic, whatever the zero points
        f = 1 if ((k == L - 2 or zt == -128) and one_fma_verdict(l.M, l.n)[0]) else 0
        if f == 0 and k == L - 1 and zt == -128 and single_rounding_verdict(l.M, l.n):
            f = 2                                                       # output layer: the single-rounding form
        out.append(f)
    return out


@pytest.mark.gpu
def test_library_verdict_equals_numpy_on_every_golden_bundle():
    import torch
    import sesrq
    seen = set()
    some_true = False
    for path in PARAM_FILES:
        fx, meta, net, x = fixture_case(path)
        key = tuple((l.M, l.n) for l in net.layers) + tuple(net.zero)
        if key in seen:
            continue
        seen.add(key)
        e = sesrq.Engine(bundle_from_oracle(net), torch.device("cuda:0"))
        want = expected_flags(net)
        assert e.one_fma_layers() == want, (os.path.basename(path), e.one_fma_layers(), want)
        some_true |= any(want)
    assert some_true, "no golden bundle exercises the one-fma form"


_CHILD = r"""
import hashlib, sys, numpy as np, torch
sys.path[:0] = [{root!r}, {root!r} + "/sesr-pytorch-quantize_amd", {root!r} + "/tests"]
from helpers import bundle_from_oracle, fixture_case
import sesrq
fx, meta, net, x = fixture_case({path!r})
e = sesrq.Engine(bundle_from_oracle(net), torch.device("cuda:0"))
q, y = e.forward(torch.from_numpy(x).cuda())
print("FLAGS", int(any(e.one_fma_layers())), "SHA", hashlib.sha256(q.cpu().numpy().tobytes()).hexdigest())
"""


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sesr_x4.crop", "sesr_x2_rand.crop", "nrdm_3.crop"])
def test_forward_bits_do_not_depend_on_the_form(name):
    """Same fixture, one process per setting of SESRQ_DIRECT: identical int8 output, and it is the reference's own."""
    import hashlib
    cands = [p for p in PARAM_FILES if os.path.basename(p) == name + ".npz"]
    assert cands, name
    path = cands[0]
    fx, meta, net, x = fixture_case(path)
    if not any(expected_flags(net)):
        pytest.skip("no layer of this bundle passes the one-fma proof")
    out = {}
    for knob in ("1", "0"):
        env = dict(os.environ, SESRQ_DIRECT=knob)
        r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, path=path)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("FLAGS")][-1].split()
        out[knob] = (int(line[1]), line[3])
    assert out["1"][0] == 1 and out["0"][0] == 0, out
    assert out["1"][1] == out["0"][1], out
    st = O.forward(net, x)
    want = st["q_out"] if "q_out" in st else None
    if want is not None:
        assert hashlib.sha256(np.ascontiguousarray(want).tobytes()).hexdigest() == out["1"][1]
