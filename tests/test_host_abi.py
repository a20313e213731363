"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/sesrq.h declares, and its host-scalar entry points (load-time arithmetic of the path)
agree with the golden tables and with the oracle.  No device calls here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, GOLDEN, golden_files, load_fixture
from oracle import sesrq_oracle as O
import sesrq
from sesrq import _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sesrq.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sesrq_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 12
    handle = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(handle, n), f"libsesrq.so lacks {n}"
    assert sorted(_lib.SYMBOLS) == names, "python binding and header disagree"
    assert _lib.lib().sesrq_version() == _lib.ABI_VERSION == 4


def test_requant_const_matches_reference_table():
    t = np.load(os.path.join(GOLDEN, "tables.npz"))
    for r, M, n in zip(t["r"], t["M"], t["n"]):
        assert sesrq.requant_const(float(r)) == (int(M), int(n)), r


def test_requant_const_errors():
    with pytest.raises(ValueError):
        sesrq.requant_const(0.0)
    with pytest.raises(ValueError):
        sesrq.requant_const(1.0, 16, 16)       # reference asserts data_bit < shift_max
    with pytest.raises(ValueError):
        sesrq.requant_const(1e9)


def test_weight_quantiser_matches_reference_table():
    t = np.load(os.path.join(GOLDEN, "tables.npz"))
    for w, q, s in zip(t["wq_in"], t["wq_out"], t["wq_scale"]):
        gq, gs = sesrq.quantize_weight(w)
        assert gs == float(s)
        np.testing.assert_array_equal(gq, q)
    with pytest.raises(ValueError, match="all zero"):
        sesrq.quantize_weight(np.zeros((2, 2, 3, 3), np.float32))


def test_calib_scale_zero_vs_oracle():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        mn = float(np.float32(rng.uniform(-3, 1)))
        mx = float(np.float32(mn + rng.uniform(1e-3, 5)))
        assert sesrq.calib_scale_zero(mn, mx) == O.calib_scale_zero(mn, mx)
    with pytest.raises(ValueError):
        sesrq.calib_scale_zero(1.0, 1.0)


PARAM_FILES = golden_files("*.params.npz")


@pytest.mark.parametrize("path", PARAM_FILES, ids=[os.path.basename(p)[:-11] for p in PARAM_FILES])
def test_derive_bundle_reproduces_reference_bundle(path):
    p, pm = load_fixture(path)
    fx, meta = load_fixture(path.replace(".params.npz", ".crop.npz"))
    sz = [sesrq.calib_scale_zero(0.0 if i == 5 else pm["min"][i], pm["max"][i]) for i in range(6)]
    assert [s for s, _ in sz] == meta["scale"] and [z for _, z in sz] == meta["zero"]
    ps = {5: 4, 6: 2, 3: 1}[pm["mflag"]]
    b = sesrq.derive_bundle([p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)], pm["scale"], pm["zero"], ps)
    for k in range(5):
        np.testing.assert_array_equal(b.layers[k].wq, fx[f"Wq{k}"])
        np.testing.assert_array_equal(b.layers[k].add_const, fx[f"add_const{k}"])
        assert (b.layers[k].M, b.layers[k].n) == (meta["M"][k], meta["n"][k])
        assert b.layers[k].w_scale == meta["wscale"][k]
    assert (b.M_res, b.n_res) == (meta["M_res"], meta["n_res"])


def test_add_const_with_odd_zero_points():
    rng = np.random.default_rng(3)
    for _ in range(50):
        oc, ic, k = 16, int(rng.integers(1, 17)), int(rng.choice([3, 5]))
        wq = rng.integers(-128, 128, (oc, ic, k, k)).astype(np.int8)
        b = (rng.standard_normal(oc) * 0.3).astype(np.float32)
        s_in, s_w, z = float(rng.uniform(1e-3, 0.05)), float(rng.uniform(1e-4, 0.01)), int(rng.integers(-160, -90))
        np.testing.assert_array_equal(sesrq.add_const(b, wq, s_in, z, s_w), O.add_const(b, wq, s_in, z, s_w))


def test_bundle_roundtrip(tmp_path):
    from helpers import bundle_from_oracle
    b = bundle_from_oracle(O.synth_net("sesr_x2", 3, hard=True))
    f = str(tmp_path / "b.npz")
    b.save(f)
    c = sesrq.Bundle.load(f)
    assert c.zero == b.zero and c.scale == b.scale and (c.M_res, c.n_res) == (b.M_res, b.n_res)
    for x, y in zip(b.layers, c.layers):
        np.testing.assert_array_equal(x.wq, y.wq)
        np.testing.assert_array_equal(x.add_const, y.add_const)
        assert (x.M, x.n, x.relu) == (y.M, y.n, y.relu)
    # golden fixtures load as bundles directly
    g = sesrq.Bundle.load(os.path.join(GOLDEN, "sesr_x4.crop.npz"))
    assert g.pixel_shuffle == 4 and g.L == 5 and g.in_channels == 1 and g.out_channels == 1


def test_engine_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from helpers import bundle_from_oracle
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sesrq.Engine(bundle_from_oracle(O.synth_net("nrdm", 0)))


def test_submission_pool_survives_fork():
    """Round 5: the pool of sesrq_forward_many is per process.  A child forked AFTER the parent's pool has threads gets a working pool of
    its own on its first call (pthread_atfork drops the parent's objects: round 4's child published jobs to threads that do not exist in
    it and spun forever) -- no device needed: sesrq_submit_selftest drives the same hand-off with jobs that only count themselves."""
    import signal
    lib = _lib.lib()
    assert lib.sesrq_submit_selftest(4, 50) == 200          # the parent's pool has three threads now
    pid = os.fork()
    if pid == 0:
        signal.alarm(20)                                     # a hang in the child must not hang the test session
        ok = lib.sesrq_submit_selftest(4, 50) == 200 and lib.sesrq_submit_selftest(6, 5) == 30
        os._exit(0 if ok else 1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, f"forked child: status {status:#x}"
    assert lib.sesrq_submit_selftest(4, 50) == 200          # and the parent's is untouched
    assert lib.sesrq_submit_selftest(1, 1) == -1 and "n_streams" in _lib.last_error()


def test_instance_registry_lists_every_kernel_family():
    """The library enumerates the kernel instantiations it can launch (no device needed: filled when the library is loaded)."""
    inst = _lib.instances()
    fam = {}
    for n in inst:
        fam[n.split("<")[0]] = fam.get(n.split("<")[0], 0) + 1
    for f in ("conv_dot4_kernel", "mfma_f5_kernel_w4", "mfma_f5_kernel", "mfma_h3_kernel", "mfma_h5_kernel", "mfma_h5p_kernel", "mfma_trio_kernel",
              "unpack_nhwc16_kernel", "verify_fastdiv_kernel", "calib_conv_kernel", "calib_minmax_kernel", "calib_hist_kernel", "calib_fakequant_kernel"):
        assert fam.get(f, 0) >= 1, (f, fam)
    assert fam["mfma_trio_kernel"] == 9 and "mfma_trio_kernel<1, 15>" in inst and "mfma_h5_kernel<1, 2, 22, 3, 0>" in inst      # what bench.py times
    assert all(v == 0 for k, v in inst.items() if k.startswith("mfma_")), "nothing has been launched in a CPU session"


def test_per_channel_weight_scales_host_side():
    """Round 5: one weight scale per OUTPUT channel as a bundle option -- NOT in the reference (its quantiser is per
    tensor, quan_func.py:58-71; BASELINE's north star names per-channel weights): PARITY UNPINNED.  Host side: the library's per-channel
    quantiser and the derived per-channel requant constants equal the numpy oracle's definition; a bundle keeps them through save / load;
    the per-tensor derivation is untouched (no M_oc)."""
    p, pm = load_fixture(os.path.join(GOLDEN, "sesr_x2_rand.params.npz"))
    Wf = [p[f"Wf{k}"] for k in range(5)]
    bf = [p[f"bf{k}"] for k in range(5)]
    for w in Wf:
        q, s = sesrq.quantize_weight_per_channel(w)
        oq, os_ = O.quantize_weight_per_channel(w)
        np.testing.assert_array_equal(q, oq)
        assert list(s) == list(os_)
        assert all(np.abs(q[o].astype(np.int32)).max() >= 127 for o in range(q.shape[0])), "every channel uses its full int8 range"
    b = sesrq.derive_bundle(Wf, bf, pm["scale"], pm["zero"], 2, per_channel=True)
    net = O.derive_net(Wf, bf, pm["scale"], pm["zero"], 2, per_channel=True)
    for lb, ln in zip(b.layers, net.layers):
        np.testing.assert_array_equal(lb.wq, ln.wq)
        np.testing.assert_array_equal(lb.add_const, ln.add_const)
        assert list(lb.M_oc) == list(ln.M_oc) and list(lb.n_oc) == list(ln.n_oc)
        assert len(set(zip(lb.M_oc.tolist(), lb.n_oc.tolist()))) > 1, "the channels really differ"
    assert (b.M_res, b.n_res) == (net.M_res, net.n_res)
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        b.save(os.path.join(d, "pc.npz"))
        b2 = sesrq.Bundle.load(os.path.join(d, "pc.npz"))
    assert all(list(x.M_oc) == list(y.M_oc) and list(x.n_oc) == list(y.n_oc) for x, y in zip(b.layers, b2.layers))
    bt = sesrq.derive_bundle(Wf, bf, pm["scale"], pm["zero"], 2)
    assert all(l.M_oc is None and l.n_oc is None for l in bt.layers)
    with pytest.raises(ValueError, match="output channel 1"):
        w0 = Wf[1].copy()
        w0[1] = 0
        sesrq.quantize_weight_per_channel(w0)
