"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI, against
(1) the golden vectors produced by the reference itself and (2) the pinned oracle on seeded
inputs.  Bar: bit-exact on every int8/int32 tensor; the fp32 output is a deterministic
function of the int8 tensor ((q - z) * f32(scale), one rounding) and must be bit-equal too."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import golden_files
from helpers import bundle_from_oracle, fixture_case, rand_frame
from oracle import sesrq_oracle as O
import sesrq
from sesrq import _lib

pytestmark = pytest.mark.gpu

STAGE_FILES = [f for f in golden_files() if not f.endswith((".params.npz", "tables.npz", ".stimtxt.npz", ".anchor.npz"))]
# kernel families behind the same ABI: dot4 (one lane per pixel), mfma (one launch per layer), trio (MFMA kernels with every
# eligible run of three hidden 3x3 layers fused into one launch: the default)
ENGINES = [("dot4", dict(engine=_lib.ENGINE_DOT4)), ("mfma", dict(engine=_lib.ENGINE_MFMA, fuse_hidden=0)),
           ("trio", dict(engine=_lib.ENGINE_MFMA, fuse_hidden=1))]


def make_engine(net, eng, **kw):
    return sesrq.Engine(bundle_from_oracle(net), _dev(), **eng[1], **kw)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _cmp(name, got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    if got.shape != want.shape:
        raise AssertionError(f"{name}: shape {got.shape} != {want.shape}")
    bad = np.argwhere(got != want)
    if len(bad):
        i = tuple(bad[0])
        raise AssertionError(f"{name}: {len(bad)} mismatches, first at {i}: got {got[i]} want {want[i]}")


@pytest.mark.parametrize("eng", ENGINES, ids=[e[0] for e in ENGINES])
@pytest.mark.parametrize("path", STAGE_FILES, ids=[os.path.basename(p)[:-4] for p in STAGE_FILES])
def test_golden_stage_by_stage(path, eng):
    """Every tensor the reference dumped (input.K, pe_outputK_P, pe_add_outputK, final) on the
    reference's own inputs, including the adversarial zero-point / saturating-weight runs."""
    fx, meta, net, x = fixture_case(path)
    e = make_engine(net, eng)
    xt = torch.from_numpy(x).to(_dev())
    # the PE dump taps: dot4 kernels on the dot4 engine, the per-PE MFMA kernels themselves (GEN_TAP) on the MFMA engine
    use_pe = eng[0] in ("dot4", "mfma")
    res = e.forward_debug(xt, pe=use_pe, special=True)       # special: shortcut_tensor.pt / input.4.spcial.pt (quan_func.py:549, :254)
    if eng[0] != "dot4":
        assert all(s.startswith("mfma") for s in e.layer_engines()), e.layer_engines()
        names = e.layer_engines()
        hidden_merged = all(l.M > 0 for l in net.layers) and not any("general" in s_ or "hybrid" in s_ for s_ in
                                                                     sesrq.Engine(bundle_from_oracle(net), _dev(), fuse_hidden=0).layer_engines()[1:4])
        if eng[0] == "mfma":
            assert not any("trio" in s_ for s_ in names), names
        else:
            assert ("mfma-trio-merged" in names) == hidden_merged, names
    r = net.pixel_shuffle
    got = {k: v.cpu().numpy() for k, v in res.items()}
    # un-shuffle q_out to compare with input5
    N, C, Ho, Wo = got["q_out"].shape
    q5 = got["q_out"].reshape(N, C, Ho // r, r, Wo // r, r).transpose(0, 1, 3, 5, 2, 4).reshape(N, C * r * r, Ho // r, Wo // r)
    got["input5"] = q5
    got["out"] = got["y"]
    assert got["shortcut"].dtype == np.float32 and got["input4_special"].dtype == np.int8
    for k in range(5):
        if use_pe:
            got[f"pe_out{k}"] = got[f"pe_out{k}"][0]
    for name, want_sha in meta["sha"].items():
        if name.startswith("pe_") and not use_pe:
            continue
        assert _sha(got[name]) == want_sha, f"{name} differs from the reference"
    for name in fx.files:
        if name in got and name not in ("x",):
            _cmp(name, got[name], fx[name])
    if eng[0] == "mfma":     # without the input taps layer 0 keeps its MFMA kernel too: its PE taps must be the same bytes
        res2 = e.forward_debug(xt, pe=True, acts=False)
        for nm in ("pe_out0", "pe_add0", "pe_out4", "pe_add4", "q_out"):
            assert torch.equal(res2[nm], res[nm]), nm
    # production call (no taps; this is where the fused hidden trio runs) must give the same result -- in each of its three output kinds:
    # they are different kernel instantiations of the last layer (round 4's wrong bytes lived in the int8-only one)
    for wq, wf in ((True, False), (False, True), (True, True)):
        q, y = e.forward(xt, want_q=wq, want_f=wf)
        assert (q is None) == (not wq) and (y is None) == (not wf)
        if wq:
            _cmp(f"q_out(production, q={wq}, f={wf})", q, got["q_out"])
        if wf:
            _cmp(f"y(production, q={wq}, f={wf})", y, got["y"])
    q, y = e.forward(xt)
    _cmp("q_out(production) vs golden input5", np.ascontiguousarray(q.cpu().numpy().reshape(N, C, Ho // r, r, Wo // r, r).transpose(0, 1, 3, 5, 2, 4).reshape(q5.shape)), fx["input5"])


SIZES = [(1, 1, 1), (1, 3, 5), (1, 8, 32), (1, 9, 33), (2, 17, 70), (1, 40, 129), (3, 31, 64), (1, 26, 121)]


@pytest.mark.parametrize("eng", ENGINES, ids=[e[0] for e in ENGINES])
@pytest.mark.parametrize("kind", ["sesr_x4", "sesr_x2", "nrdm"])
@pytest.mark.parametrize("hard", [False, True], ids=["plain", "hard"])
def test_synthetic_nets_vs_oracle(kind, hard, eng):
    """Seeded random bundles (plain: saturation-free -> merged kernels; hard: wide weights and odd
    zero points -> general kernels with the 18/20-bit clamps firing), ragged sizes, batches."""
    for seed in range(2):
        net = O.synth_net(kind, seed, hard=hard)
        e = make_engine(net, eng)
        cin = net.layers[0].wq.shape[1]
        for (N, H, W) in SIZES:
            x = rand_frame((N, cin, H, W), 1000 * seed + H * W)
            want = O.forward(net, x)
            q, y = e.forward(torch.from_numpy(x).to(_dev()))
            _cmp(f"{net.name} {N}x{H}x{W} q_out", q, want["q_out"])
            _cmp(f"{net.name} {N}x{H}x{W} y", y, want["y"])


@pytest.mark.parametrize("eng", ENGINES, ids=[e[0] for e in ENGINES])
def test_general_and_merged_kernels_agree(eng):
    """force_general runs the per-PE kernels on a saturation-free bundle: same bits."""
    net = O.synth_net("sesr_x2", 5)
    x = torch.from_numpy(rand_frame((2, 3, 37, 91), 9)).to(_dev())
    e0 = make_engine(net, eng)
    e1 = make_engine(net, eng, force_general=True)
    assert any("merged" in s for s in e0.layer_engines())
    q0, y0 = e0.forward(x)
    q1, y1 = e1.forward(x)
    assert torch.equal(q0, q1) and torch.equal(y0, y1)
    want = O.forward(net, x.cpu().numpy())
    _cmp("q_out", q0, want["q_out"])


@pytest.mark.parametrize("cin", [2, 4])
def test_first_layer_with_two_and_four_input_channels(cin):
    """The first-layer kernels are specialised for 1 and 3 input channels (the reference's nets); any other count up to the
    PE width (a 4-channel Bayer RGGB frame, say) takes the run-time channel path of the same kernels: fp32 and int8 frames,
    every launch plan, merged and per-PE accumulation."""
    rng = np.random.default_rng(70 + cin)
    for hard in (False, True):
        net = O.synth_net("nrdm", 30 + cin, hard=hard)
        w0 = net.layers[0].wq
        net.layers[0].wq = rng.integers(-128, 128, size=(w0.shape[0], cin, 5, 5)).astype(np.int8) if hard else \
            np.clip(np.rint(rng.standard_normal((w0.shape[0], cin, 5, 5)) * 14.0), -128, 127).astype(np.int8)
        x = rng.random((2, cin, 37, 91), dtype=np.float32)
        want = O.forward(net, x)
        q0 = O.quantize_input(x, net.scale[0], net.zero[0])
        for name, kw in ENGINES:
            e = make_engine(net, (name, kw))
            q, y = e.forward(torch.from_numpy(x).to(_dev()))
            _cmp(f"cin {cin} hard {hard} {name} {e.layer_engines()[0]}", q, want["q_out"])
            _cmp("y", y, want["y"])
            q8, _ = e.forward(torch.from_numpy(q0).to(_dev()), want_f=False)
            _cmp(f"cin {cin} hard {hard} {name}: int8 frame", q8, want["q_out"])


@pytest.mark.parametrize("acc_bits,add_bits", [(16, 18), (17, 17), (20, 22), (24, 26)])
def test_other_pe_bit_widths(acc_bits, add_bits):
    """define.py's PE_ACC_BIT / PE_ADD_BIT are configuration, not constants: narrower accumulators saturate often (and with
    add_bits <= acc_bits + 1 the adder clamp is no longer a provable no-op), wider ones never -- the kernels take the widths
    from the bundle (literal clamps only for the reference's 18 / 20).  Every engine vs the oracle, C oracle vs numpy."""
    from oracle import c_oracle as CO
    for kind in ("sesr_x2", "nrdm"):
        net = O.synth_net(kind, 11, hard=True)
        net.acc_bits, net.add_bits = acc_bits, add_bits
        x = rand_frame((1, 3, 30, 77), 5)
        want = O.forward(net, x, keep=True)
        sat = [int((np.abs(want[f"pe_raw{k}"]) >= 2 ** (acc_bits - 1)).sum()) for k in range(net.L)]
        assert (sum(sat) > 0) == (acc_bits < 20), sat                   # the narrow cases really saturate, the wide ones never
        _cmp("C oracle vs numpy oracle", CO.forward(net, x)["q_out"], want["q_out"])
        for name, kw in ENGINES:
            e = make_engine(net, (name, kw))
            q, y = e.forward(torch.from_numpy(x).to(_dev()))
            _cmp(f"{kind} acc {acc_bits} add {add_bits} {name} {e.layer_engines()}", q, want["q_out"])
            _cmp("y", y, want["y"])


def test_int8_input_path():
    """The boundary also accepts an already-quantised q0 (input.0.pt) instead of the fp32 frame."""
    net = O.synth_net("nrdm", 2, hard=True)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = rand_frame((1, 3, 20, 45), 4)
    q0 = O.quantize_input(x, net.scale[0], net.zero[0])
    qa, _ = e.forward(torch.from_numpy(x).to(_dev()))
    qb, _ = e.forward(torch.from_numpy(q0).to(_dev()))
    assert torch.equal(qa, qb)


def test_deeper_net_positional_roles():
    """nrdm_6-shaped net (8 convs): roles generalised by position; parity UNPINNED w.r.t. the
    reference (it cannot int-simulate this depth) -- oracle vs HIP self-consistency only."""
    net = O.synth_net("nrdm", 11, n_blocks=6)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = rand_frame((1, 3, 33, 47), 5)
    want = O.forward(net, x)
    q, y = e.forward(torch.from_numpy(x).to(_dev()))
    _cmp("q_out", q, want["q_out"])
    _cmp("y", y, want["y"])


def test_full_size_properties_1080p():
    """BASELINE config 2 size (1x3x1080x1920 -> 3x2160x3840): size-independent properties.
    (a) translation consistency: a crop that keeps a 7-pixel halo reproduces the interior of the
        full-frame result bit-for-bit (receptive field 2+1+1+1+2 = 7);
    (b) batch independence: frames in a batch equal the same frames run alone;
    (c) determinism: two runs give identical bytes."""
    net = O.synth_net("sesr_x2", 0)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    assert all(s.startswith("mfma") for s in e.layer_engines()), "default engine = MFMA kernels"
    x = torch.from_numpy(rand_frame((1, 3, 1080, 1920), 2)).to(_dev())
    q, y = e.forward(x)
    q2, _ = e.forward(x)
    assert torch.equal(q, q2)
    y0, x0, h, w = 500, 900, 64, 96
    crop = x[:, :, y0 - 7:y0 + h + 7, x0 - 7:x0 + w + 7].contiguous()
    qc, _ = e.forward(crop)
    assert torch.equal(qc[:, :, 14:14 + 2 * h, 14:14 + 2 * w], q[:, :, 2 * y0:2 * (y0 + h), 2 * x0:2 * (x0 + w)])
    want = O.forward(net, crop.cpu().numpy())
    _cmp("crop vs oracle", qc, want["q_out"])
    xb = torch.cat([x[:, :, :270, :480], x[:, :, 270:540, 480:960]], 0).contiguous()
    qb, _ = e.forward(xb)
    qa0, _ = e.forward(xb[0:1].contiguous())
    qa1, _ = e.forward(xb[1:2].contiguous())
    assert torch.equal(qb[0:1], qa0) and torch.equal(qb[1:2], qa1)


def test_4k_input_frame_large_offsets():
    """A 2160x3840 input (-> 4320x7680, 99.5 MB int8 / 398 MB fp32 out; 133 MB per NHWC16 tensor): the largest frame class the
    32-bit buffer offsets are specified for.  Every launch plan gives the same bytes, fp32 and int8 outputs agree, and crops
    at the four corners and the centre (7-pixel halo, so the frame borders are the crop's own borders) match the oracle."""
    net = O.synth_net("sesr_x2", 5, hard=True)
    b = bundle_from_oracle(net)
    H, W = 2160, 3840
    x = torch.rand((1, 3, H, W), generator=torch.Generator().manual_seed(9)).to(_dev())
    e = sesrq.Engine(b, _dev())
    q, y = e.forward(x)
    q2, _ = sesrq.Engine(b, _dev(), fuse_hidden=0).forward(x, want_f=False)
    assert torch.equal(q, q2)
    zL, sL = net.zero[net.L], np.float32(net.scale[net.L])
    assert torch.equal(y, (q.float() - zL) * float(sL))
    for (y0, x0) in [(0, 0), (0, W - 160), (H - 120, 0), (H - 120, W - 160), (1000, 1900)]:
        ya, xa = max(y0 - 7, 0), max(x0 - 7, 0)
        yb, xb = min(y0 + 120 + 7, H), min(x0 + 160 + 7, W)
        crop = x[:, :, ya:yb, xa:xb].contiguous()
        want = O.forward(net, crop.cpu().numpy())["q_out"]
        oy, ox = y0 - ya, x0 - xa
        _cmp(f"4K frame, window at ({y0},{x0})", q[:, :, 2 * y0:2 * (y0 + 120), 2 * x0:2 * (x0 + 160)],
             want[:, :, 2 * oy:2 * (oy + 120), 2 * ox:2 * (ox + 160)])
    del q, y, q2, x
    torch.cuda.empty_cache()
    # one step further: H*W = 2^24 pixels is refused before anything is launched (32-bit buffer offsets), not computed wrongly
    big = torch.zeros((1, 3, 4096, 4096), device=_dev())
    with pytest.raises(RuntimeError, match="too large for 32-bit buffer offsets"):
        e.forward(big, want_f=False)
    del big
    torch.cuda.empty_cache()


def test_error_conventions():
    net = O.synth_net("nrdm", 0)
    b = bundle_from_oracle(net)
    e = sesrq.Engine(b, _dev())
    with pytest.raises(ValueError, match="dimension"):
        e.forward(torch.zeros(3, 8, 8, device=_dev()))
    with pytest.raises(ValueError, match="channels"):
        e.forward(torch.zeros(1, 1, 8, 8, device=_dev()))
    bad = bundle_from_oracle(net)
    bad.layers[1].M = 1 << 16
    with pytest.raises(ValueError, match="requant"):
        sesrq.Engine(bad, _dev())
    bad = bundle_from_oracle(net)
    bad.pe_num = 8
    with pytest.raises(ValueError, match="pe_num"):
        sesrq.Engine(bad, _dev())
    # option values outside their range are refused by sesrq_create (the net would otherwise be built on a guess)
    # (fuse_hidden = 2 was the fused front of rounds 2-3, retired in round 4)
    for kw, msg in ((dict(fuse_hidden=2), "fuse_hidden"), (dict(fuse_hidden=-1), "fuse_hidden"), (dict(engine=9), "engine")):
        with pytest.raises(ValueError, match=msg):
            sesrq.Engine(b, _dev(), **kw)
    # a scale the reciprocal form cannot represent (1/s0 overflows): exact_div = 2 is refused, the default falls back to the
    # division (first layer on the dot4 kernel) and still matches the oracle
    tiny = O.synth_net("nrdm", 0)
    tiny.scale[0] = 1e-39
    with pytest.raises(ValueError, match="exact_div = 2"):
        sesrq.Engine(bundle_from_oracle(tiny), _dev(), reciprocal_division=True)
    et = sesrq.Engine(bundle_from_oracle(tiny), _dev())
    assert not et.fast_division_proven() and et.layer_engines()[0].startswith("dot4")
    x = rand_frame((1, 3, 9, 21), 2) * np.float32(2e-37)
    _cmp("unproven scale: division on the dot4 kernel", et.forward(torch.from_numpy(x).to(_dev()))[0], O.forward(tiny, x)["q_out"])


def test_fast_division_is_proven_and_equals_exact_division():
    """The 3-instruction reciprocal form of the input quantiser is used only after sesrq_create proved it
    exhaustively for the net's (scale, zero); with the proof on or forced off the bits are the same, on
    adversarial inputs too (ties of x/s+z at .5, denormals, huge / negative values)."""
    net = O.synth_net("sesr_x2", 1)
    b = bundle_from_oracle(net)
    e_fast = sesrq.Engine(b, _dev(), engine=_lib.ENGINE_MFMA)
    e_exact = sesrq.Engine(b, _dev(), engine=_lib.ENGINE_MFMA, exact_division=True)
    assert e_fast.fast_division_proven() and e_fast.layer_engines()[0].startswith("mfma-f5")
    # the MFMA first-layer kernels carry only the proven form: with the division forced, layer 0 runs on the dot4 kernel
    assert e_exact.fast_division_proven() and e_exact.layer_engines()[0].startswith("dot4") and \
        e_exact.layer_engines()[1:] == e_fast.layer_engines()[1:]
    rng = np.random.default_rng(0)
    s0, z0 = np.float32(net.scale[0]), net.zero[0]
    x = rng.random((1, 3, 64, 96), dtype=np.float32)
    ties = ((np.arange(-140, 140, dtype=np.float64) + 0.5 - z0) * float(s0)).astype(np.float32)   # x/s+z near k+.5
    x[0, 0, 0, :len(ties[:96])] = ties[:96]
    x[0, 1, 1, :96] = np.nextafter(ties[96:192], np.float32(1e9))
    x[0, 2, 2, :88] = np.nextafter(ties[192:280], np.float32(-1e9))
    x[0, 0, 3, :8] = np.array([0.0, -0.0, 1e-42, -1e-42, 3.0e38, -3.0e38, 1e-30, 123456.0], np.float32)
    xt = torch.from_numpy(x).to(_dev())
    qa, _ = e_fast.forward(xt)
    qb, _ = e_exact.forward(xt)
    assert torch.equal(qa, qb)
    _cmp("vs oracle", qa, O.forward(net, x)["q_out"])
    for tag in ("sesr_x4", "nrdm_3", "sesr_x2_rand"):
        fx, meta, gnet, gx = fixture_case(os.path.join(os.path.dirname(STAGE_FILES[0]), f"{tag}.crop.npz"))
        assert sesrq.Engine(bundle_from_oracle(gnet), _dev()).fast_division_proven(), tag


def test_reciprocal_division_option_is_the_gpu_run_reference_quantiser():
    """sesrq_options.exact_div = 2: x * fl(1/s0) instead of the quotient (what torch evaluates for tensor / scalar on a GPU,
    where the reference's scripts run).  PARITY UNPINNED w.r.t. the reference (its goldens are CPU runs): every first-layer
    kernel vs the numpy restatement, q0 within one LSB of the default quantiser's and different from it on some tie."""
    net = O.synth_net("sesr_x2", 1)
    net.scale[0] = 0.0039            # the synthetic 1/255 has fl(1/s0) == 255: both forms agree on every tie; this one does not
    b = bundle_from_oracle(net)
    s0, z0 = np.float32(net.scale[0]), net.zero[0]
    rng = np.random.default_rng(5)
    x = rng.random((1, 3, 40, 300), dtype=np.float32)
    k = rng.integers(-128, 127, size=(3, 39, 300))
    x[0, :, 1:, :] = ((k + 0.5 - z0) * float(s0)).astype(np.float32)                 # x/s + z at k + .5, to an ulp
    q0d, q0r = O.quantize_input(x, s0, z0), O.quantize_input(x, s0, z0, reciprocal=True)
    d = q0r.astype(int) - q0d.astype(int)
    assert np.abs(d).max() == 1 and 0 < np.count_nonzero(d) < d.size // 2 and not d[0, :, 0].any()   # ties only
    xt = torch.from_numpy(x).to(_dev())
    ref = sesrq.Engine(b, _dev())
    want_q, _ = ref.forward(torch.from_numpy(q0r).to(_dev()))                       # an int8 frame IS q0 (test_int8_input_path)
    base_q, _ = ref.forward(xt)
    assert not torch.equal(want_q, base_q)
    for kw in (dict(engine=_lib.ENGINE_DOT4), dict(fuse_hidden=0), dict(fuse_hidden=1)):
        e = sesrq.Engine(b, _dev(), reciprocal_division=True, **kw)
        assert e.fast_division_proven()                                             # the proof is reported whatever form runs
        if kw.get("engine") == _lib.ENGINE_DOT4:
            _cmp("q0 (input.0 tap)", e.forward_debug(xt, pe=False)["input0"], q0r)
        q, _ = e.forward(xt)
        assert torch.equal(q, want_q), kw
    with pytest.raises(ValueError, match="exclude"):
        sesrq.Engine(b, _dev(), reciprocal_division=True, exact_division=True)


@pytest.mark.parametrize("layer,arch", [(0, "sesr_x2"), (1, "sesr_x2"), (3, "sesr_x2"), (4, "sesr_x4"), (4, "sesr_x2"), (4, "nrdm")])
def test_hybrid_single_risky_pe(layer, arch):
    """Exactly one PE of a layer can saturate -> merged chain + that PE's chain ('hybrid' kernels).  The
    weights of one (oc, PE) pair are blown up so that the 18-bit clamp really fires."""
    net = O.synth_net(arch, 7)
    w = (net.layers[layer].wq.astype(np.int32) // 2).astype(np.int8)     # keep the other three PEs provably safe
    pe = 2 if layer == 0 else 1
    w[5 % w.shape[0], pe::4, :, :] = 127
    w[7 % w.shape[0], pe::4, :, :] = -128
    net.layers[layer].wq = w
    e = sesrq.Engine(bundle_from_oracle(net), _dev(), engine=_lib.ENGINE_MFMA)
    # a last layer with OC <= 4 runs the pe-split kernel (all four PE sums come out of one chain anyway)
    assert ("h5p-general" if (layer == 4 and arch == "nrdm") else "hybrid") in e.layer_engines()[layer], e.layer_engines()
    x = rand_frame((2, net.layers[0].wq.shape[1], 41, 77), 21)
    x[0, :, :20] = 1.0                       # bright region: large positive PE sums
    want = O.forward(net, x, keep=False)
    q, y = e.forward(torch.from_numpy(x).to(_dev()))
    _cmp("q_out", q, want["q_out"])
    st = O.forward(net, x[:1], keep=True)
    pe_out = st[f"pe_out{layer}"]
    assert (np.abs(pe_out) >= 131071).any(), "test net does not reach the 18-bit clamp"
    e2 = sesrq.Engine(bundle_from_oracle(net), _dev(), engine=_lib.ENGINE_MFMA, force_general=True)
    q2, _ = e2.forward(torch.from_numpy(x).to(_dev()))
    assert torch.equal(q, q2)


@pytest.mark.parametrize("case", ["sesr_x4", "nrdm_3", "sesr_x2_rand"])
def test_calibration_pass_matches_reference_ranges(case):
    """exe_mode 0 on the GPU: running min/max of every quantiser input on the reference's own random frame
    vs the ranges the reference observed (tests/golden/*.params.npz).  fp32 summation order differs from
    oneDNN's, so the bar is a tolerance (1e-4 of the range), written here; the derived integer bundle must
    still reproduce the reference's requant constants."""
    from sesrq.calibrate import Calibrator
    from conftest import GOLDEN, load_fixture
    p, pm = load_fixture(os.path.join(GOLDEN, f"{case}.params.npz"))
    ps = {5: 4, 6: 2, 3: 1}[pm["mflag"]]
    cal = Calibrator([p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)], ps, _dev())
    x = np.load(os.path.join(GOLDEN, "rand_SR_Input_80x960.npy" if pm["mflag"] == 5 else "rand_DM_Input_80x960.npy"))
    y = cal.observe(torch.from_numpy(x).to(_dev()))
    r = ps
    assert tuple(y.shape) == (1, p["Wf4"].shape[0] // (r * r), 80 * r, 960 * r)
    for k in range(6):
        span = pm["max"][k] - pm["min"][k]
        assert abs(cal.run_min[k] - pm["min"][k]) <= 1e-4 * span, (k, cal.run_min[k], pm["min"][k])
        assert abs(cal.run_max[k] - pm["max"][k]) <= 1e-4 * span, (k, cal.run_max[k], pm["max"][k])
    scale, zero = cal.finalize()
    assert zero == pm["zero"]
    np.testing.assert_allclose(scale, pm["scale"], rtol=2e-4)
    b = cal.bundle()
    fx, meta = load_fixture(os.path.join(GOLDEN, f"{case}.crop.npz"))
    for k in range(5):
        np.testing.assert_array_equal(b.layers[k].wq, fx[f"Wq{k}"])
        assert b.layers[k].n == meta["n"][k] and abs(b.layers[k].M - meta["M"][k]) <= max(2, meta["M"][k] * 3e-4)
    # a second frame only widens the ranges; reset() forgets them
    cal.observe(torch.from_numpy(np.ascontiguousarray(x[:, :, ::-1, :]) * 0.5).to(_dev()))
    assert all(cal.run_max[k] >= pm["max"][k] * (1 - 1e-4) for k in range(6))
    cal.reset()
    assert cal.run_min[0] is None


def test_config4_shape_batch_of_x4_frames():
    """BASELINE config 4 per-GPU share: SESR-x4 540p -> 4K, 4 frames in one call.  Frame 0 against the C oracle
    (full size), the others through batch independence."""
    from oracle import c_oracle as CO
    fx, meta, net, _ = fixture_case(os.path.join(os.path.dirname(STAGE_FILES[0]), "sesr_x4.crop.npz"))
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = rand_frame((4, 1, 540, 960), 4)
    xt = torch.from_numpy(x).to(_dev())
    q, y = e.forward(xt)
    assert tuple(q.shape) == (4, 1, 2160, 3840)
    want = CO.forward(net, x[:1], want_f=False)["q_out"]
    _cmp("frame 0 vs C oracle", q[:1], want)
    for i in range(1, 4):
        qi, _ = e.forward(xt[i:i + 1].contiguous())
        assert torch.equal(qi, q[i:i + 1])


def test_config5_shape_nrdm6_then_sesr_x2_chain():
    """BASELINE config 5 shape: an 8-conv NRDM net, its float output handed to SESR-x2 (float hand-off between
    nets).  The reference has no integer path for either the 8-conv depth or the chain: parity
    UNPINNED -- oracle chain vs HIP chain self-consistency."""
    nr = O.synth_net("nrdm", 21, n_blocks=6)
    sr = O.synth_net("sesr_x2", 22)
    e1 = sesrq.Engine(bundle_from_oracle(nr), _dev())
    e2 = sesrq.Engine(bundle_from_oracle(sr), _dev())
    x = rand_frame((2, 3, 45, 83), 6)
    _, y1 = e1.forward(torch.from_numpy(x).to(_dev()))
    q2, y2 = e2.forward(y1)
    w1 = O.forward(nr, x)
    w2 = O.forward(sr, w1["y"])
    _cmp("nrdm_6 float output", y1, w1["y"])
    _cmp("chain int8 output", q2, w2["q_out"])
    assert tuple(q2.shape) == (2, 3, 90, 166)
    # int8 hand-off: the second net takes the first one's int8 output and re-quantises it while staging
    # (sesrq_options.i8_in_scale/zero) -- same bits as the fp32 hand-off, a quarter of the bytes, on every first-layer kernel
    q1, _ = e1.forward(torch.from_numpy(x).to(_dev()), want_f=False)
    for kw in (dict(), dict(fuse_hidden=0), dict(engine=_lib.ENGINE_DOT4)):
        e2i = sesrq.Engine(bundle_from_oracle(sr), _dev(), upstream=bundle_from_oracle(nr), **kw)
        q2i, _ = e2i.forward(q1)
        _cmp("chain int8 hand-off", q2i, w2["q_out"])
        # without the upstream domain an int8 frame is taken as q0 itself: a different result
        assert not torch.equal(e2.forward(q1)[0], q2i)


def test_config5_real_weights_nrdm6_then_sesr_x2_540p():
    """BASELINE config 5 with the reference's own nrdm_6_G.pth: collapsed by this package (tests/golden/unpinned/
    nrdm_6.collapsed.npz), calibrated by this package's Calibrator on rand_DM_Input_80x960 (nrdm_6.bundle.npz), chained into
    the reference-calibrated SESR-x2 bundle, 960x540 -> 1920x1080, int8 hand-off.  PARITY UNPINNED: the reference cannot
    int-simulate 8 convs nor the chain; this checks HIP vs the C oracle on the same bundle, full frame."""
    from oracle import c_oracle as CO
    from conftest import GOLDEN
    from sesrq.bundle import Bundle
    path = os.path.join(GOLDEN, "unpinned", "nrdm_6.bundle.npz")
    if not os.path.isfile(path):
        pytest.skip("nrdm_6.bundle.npz not generated yet (tools/make_nrdm6_bundle.py on the GPU box)")
    b1 = Bundle.load(path)
    assert b1.L == 8 and b1.pixel_shuffle == 1
    fx, meta, sr, _ = fixture_case(os.path.join(GOLDEN, "sesr_x2_rand.crop.npz"))
    b2 = bundle_from_oracle(sr)
    e1 = sesrq.Engine(b1, _dev())
    e2 = sesrq.Engine(b2, _dev(), upstream=b1)
    assert e1.launch_plan() == [(0, 1), (1, 3), (4, 3), (7, 1)]
    x = rand_frame((1, 3, 540, 960), 5)
    q1, _ = e1.forward(torch.from_numpy(x).to(_dev()), want_f=False)
    q2, _ = e2.forward(q1, want_f=False)
    nr = O.Net(layers=[O.Layer(wq=l.wq, add_const=l.add_const, M=l.M, n=l.n, relu=l.relu) for l in b1.layers], scale=b1.scale,
               zero=b1.zero, M_res=b1.M_res, n_res=b1.n_res, pixel_shuffle=1, name="nrdm_6")
    w1 = CO.forward(nr, x, want_f=False)["q_out"]
    _cmp("nrdm_6 540p int8 output", q1, w1)
    y1 = (w1.astype(np.float32) - np.float32(b1.zero[8])) * np.float32(b1.scale[8])
    w2 = CO.forward(sr, y1, want_f=False)["q_out"]
    _cmp("chain output 1080p", q2, w2)
    assert tuple(q2.shape) == (1, 3, 1080, 1920)


def test_entropy_calibration_variant():
    """The KL-entropy calibration variant BASELINE's north star names (the reference only has min/max): PARITY UNPINNED.
    (a) the device histogram equals numpy's on the same fp32 bin formula, values outside the range land in the edge bins;
    (b) a two-pass calibration of nrdm_3 on the reference's random frame: ranges lie inside the min/max ranges, the bundle
        is valid, and the integer forward on it matches the oracle bit for bit (whatever the domains, the path is exact);
    (c) on a frame with a few hot pixels min/max wastes the input range on them, the entropy ranges do not."""
    from sesrq.calibrate import Calibrator
    from conftest import GOLDEN, load_fixture
    rng = np.random.default_rng(3)
    x = rng.normal(0.2, 1.0, 1_000_003).astype(np.float32)
    x[:5] = [np.nan, 50.0, -50.0, 3.0, -2.0]
    lo, hi, B = np.float32(-2.0), np.float32(3.0), 2048
    hist = torch.zeros(B, dtype=torch.int32, device=_dev())
    xt = torch.from_numpy(x).to(_dev())
    _lib.check(_lib.lib().sesrq_calib_histogram(xt.data_ptr(), xt.numel(), float(lo), float(hi), B, hist.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream))
    inv_w = np.float32(B) / (hi - lo)
    v = x[~np.isnan(x)]
    idx = np.clip(np.floor((v - lo) * inv_w), 0, B - 1).astype(np.int64)
    np.testing.assert_array_equal(hist.cpu().numpy(), np.bincount(idx, minlength=B))
    with pytest.raises(RuntimeError, match="bins"):
        _lib.check(_lib.lib().sesrq_calib_histogram(xt.data_ptr(), xt.numel(), 0.0, 1.0, 5000, hist.data_ptr(), None))

    p, pm = load_fixture(os.path.join(GOLDEN, "nrdm_3.params.npz"))
    frame = np.load(os.path.join(GOLDEN, "rand_DM_Input_80x960.npy"))
    hot = frame.copy()
    hot[0, :, 5, 5:9] = 40.0                                   # four hot pixels per channel
    results = {}
    for tag, fr in (("plain", frame), ("hot", hot)):
        cal = Calibrator([p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)], 1, _dev(), method="entropy")
        with pytest.raises(RuntimeError, match="min/max pass"):
            cal.begin_histogram_pass()
        ft = torch.from_numpy(fr).to(_dev())
        cal.observe(ft)
        with pytest.raises(RuntimeError, match="histogram pass"):
            cal.finalize()
        mm = list(zip(cal.run_min, cal.run_max))
        cal.begin_histogram_pass()
        cal.observe(ft)
        assert list(zip(cal.run_min, cal.run_max)) == mm                       # the second pass does not widen the ranges
        assert all(int(h.sum()) > 0 for h in cal.hist)
        scale, zero = cal.finalize()
        for k, (lo_k, hi_k) in enumerate(cal.ranges):
            assert mm[k][0] - 1e-6 <= lo_k < hi_k <= mm[k][1] + 1e-6, (k, lo_k, hi_k, mm[k])
        results[tag] = (cal, scale, zero, mm)
    cal, scale, zero, mm = results["hot"]
    assert mm[0][1] == 40.0 and cal.ranges[0][1] < 10.0 and scale[0] < 0.25 * (mm[0][1] - mm[0][0]) / 255
    b = results["plain"][0].bundle("nrdm_3_entropy")
    net = O.Net(layers=[O.Layer(wq=l.wq, add_const=l.add_const, M=l.M, n=l.n, relu=l.relu) for l in b.layers], scale=b.scale,
                zero=b.zero, M_res=b.M_res, n_res=b.n_res, pixel_shuffle=b.pixel_shuffle, pe=b.pe_num, acc_bits=b.pe_acc_bits,
                add_bits=b.pe_add_bits, name=b.name)
    q, _ = sesrq.Engine(b, _dev()).forward(torch.from_numpy(frame).to(_dev()))
    _cmp("integer forward on the entropy-calibrated bundle", q, O.forward(net, frame)["q_out"])


@pytest.mark.parametrize("eng", ENGINES, ids=[e[0] for e in ENGINES])
def test_x2_anchor_add(eng):
    """the x2 eval loop adds the nearest-upsampled input to the float output (test.py:148-155);
    fused into the last epilogue as an option.  One fp32 add per output value -> bit-exact vs numpy."""
    net = O.synth_net("sesr_x2", 3)
    e = make_engine(net, eng, anchor_add=True)
    x = rand_frame((2, 3, 37, 70), 12)
    q, y = e.forward(torch.from_numpy(x).to(_dev()))
    want = O.forward(net, x)
    _cmp("int8 output unaffected", q, want["q_out"])
    _cmp("y + upsampled input", y, want["y"] + np.repeat(np.repeat(x, 2, axis=2), 2, axis=3))
    with pytest.raises(RuntimeError, match="anchor"):       # the anchor is the fp32 frame: an int8 input cannot provide it
        e.forward(torch.from_numpy(O.quantize_input(x, net.scale[0], net.zero[0])).to(_dev()))


@pytest.mark.parametrize("eng", ENGINES, ids=[e[0] for e in ENGINES])
def test_x2_anchor_add_against_the_reference(eng):
    """The same option pinned on REFERENCE-MADE data (tests/golden/make_golden.py --case anchor): the reference's AnchorOp
    (models/sesr_arch.py:171-205) + PixelShuffle of the input, added to the reference's x2 float result exactly as its eval loop
    does (test.py:148-155), for the 24x40 crop (stored) and the 80x960 frame (SHA-256)."""
    import json
    from conftest import GOLDEN
    anc = np.load(os.path.join(GOLDEN, "sesr_x2_rand.anchor.npz"), allow_pickle=False)
    for tag in ("crop", "full"):
        fx, meta, net, x = fixture_case(os.path.join(GOLDEN, f"sesr_x2_rand.{tag}.npz"))
        q, y = make_engine(net, eng, anchor_add=True).forward(torch.from_numpy(x).to(_dev()))
        want = json.loads(str(anc[f"sha_{tag}"]))
        y = y.cpu().numpy()
        assert list(y.shape) == want["shape"]
        if tag == "crop":
            _cmp("gfake + inps_x2 (reference-made)", y, anc["sum_crop"])
            _cmp("AnchorOp + PixelShuffle (reference-made) == nearest upsampling", np.repeat(np.repeat(x, 2, axis=2), 2, axis=3), anc["up_crop"])
        assert _sha(y.astype(np.float32)) == want["sum"], tag


def test_randomised_shapes_against_c_oracle():
    """Stress the tile / chunk / strip boundaries of the persistent kernels: random frame sizes (around multiples of
    8 / 16 rows and 64 columns, tiny and tall), batches, all three topologies, merged / hybrid / general kernels.
    SESRQ_STRESS_TRIALS=<n> runs more trials (same generator, longer sequence)."""
    from oracle import c_oracle as CO
    rng = np.random.default_rng(2024)
    heights = [1, 3, 7, 8, 9, 15, 16, 17, 23, 24, 25, 31, 32, 33, 47, 48, 49, 63, 64, 65, 95, 97, 129, 255, 257, 300]
    widths = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 191, 193, 250]
    kinds = ["sesr_x4", "sesr_x2", "nrdm"]
    for trial in range(int(os.environ.get("SESRQ_STRESS_TRIALS", "24"))):
        kind = kinds[trial % 3]
        hard = (trial // 3) % 2 == 1
        net = O.synth_net(kind, 100 + trial, hard=hard)
        if trial % 4 == 3:                      # force exactly one risky PE in a random layer -> hybrid kernel
            k = int(rng.integers(0, 5))
            w = (net.layers[k].wq.astype(np.int32) // 2).astype(np.int8)
            w[int(rng.integers(w.shape[0])), int(rng.integers(min(4, w.shape[1])))::4] = 127
            net.layers[k].wq = w
        e = sesrq.Engine(bundle_from_oracle(net), _dev(), fuse_hidden=trial % 2)      # launch plans alternate: per layer / fused trio
        H, W, N = int(rng.choice(heights)), int(rng.choice(widths)), int(rng.choice([1, 1, 2, 3]))
        x = rng.random((N, net.layers[0].wq.shape[1], H, W), dtype=np.float32)
        want = CO.forward(net, x)
        q, y = e.forward(torch.from_numpy(x).to(_dev()))
        _cmp(f"trial {trial} {net.name} {N}x{H}x{W} {e.layer_engines()} q", q, want["q_out"])
        _cmp(f"trial {trial} y", y, want["y"])


@pytest.mark.parametrize("hard", [False, True], ids=["merged-hybrid", "general"])
def test_forward_is_deterministic_at_full_size(hard):
    """Three forwards of the same 1080p frame give the same bytes, int8-only and int8+fp32 output kinds alike.  (Round 3: an inline-asm
    register write directly behind a 16-byte store made ~11 k bytes of the first layer's output differ from run to run on the
    general kernels -- a store-data hazard hipcc does not pad, tools/store_hazard_probe.hip; every other test compared ONE run
    with the oracle and only the 4K-frame test happened to run a frame twice.)"""
    net = O.synth_net("sesr_x2", 5, hard=hard)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = torch.rand((1, 3, 1080, 1920), generator=torch.Generator().manual_seed(9)).to(_dev())
    q0, y0 = e.forward(x)
    for _ in range(2):
        q1, y1 = e.forward(x)
        assert torch.equal(q0, q1) and torch.equal(y0, y1)
        q2, _ = e.forward(x, want_f=False)
        assert torch.equal(q0, q2)


@pytest.mark.parametrize("hard", [False, True], ids=["merged", "general"])
@pytest.mark.parametrize("oc_last,ps", [(12, 1), (8, 2), (9, 3), (5, 1), (7, 1), (11, 1), (12, 2), (13, 1), (16, 2)])
def test_last_layer_channel_counts(oc_last, ps, hard):
    """Every way the last MFMA layer lays out its output rows (last_slot_oc): up to 12 channels -> three real rows per lane group
    (generic map, 1-byte stores; the 12 x PixelShuffle(2) pair map), more -> four; merged and per-PE kernels, int8-only (the
    compile-time store flavours) and int8 + fp32 outputs, against the C oracle."""
    from oracle import c_oracle as CO
    net = O.synth_net("sesr_x2", 300 + oc_last + 16 * ps, hard=hard)
    rng = np.random.default_rng(oc_last * 7 + ps)
    last = net.layers[-1]
    if hard:
        w = rng.choice(np.array([-128, -100, 90, 127], dtype=np.int64), size=(oc_last, 16, 5, 5))
    else:
        w = np.clip(np.rint(rng.standard_normal((oc_last, 16, 5, 5)) * 14.0), -128, 127)
    net.layers[-1] = O.Layer(wq=w.astype(np.int8), add_const=rng.integers(-6000, 6000, oc_last).astype(np.int32), M=last.M, n=last.n, relu=False)
    net.pixel_shuffle = ps
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    assert e.layer_engines()[-1].startswith("mfma-h5-"), e.layer_engines()
    x = rand_frame((2, 3, 21, 70), oc_last + ps)
    want = CO.forward(net, x)
    xt = torch.from_numpy(x).to(_dev())
    q, y = e.forward(xt)
    _cmp("q (int8 + fp32 outputs)", q, want["q_out"])
    _cmp("y", y, want["y"])
    q2, _ = e.forward(xt, want_f=False)
    _cmp("q (int8 only)", q2, want["q_out"])


def test_empty_and_degenerate_inputs():
    net = O.synth_net("nrdm", 1)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    for shape in [(0, 3, 8, 8), (1, 3, 0, 8), (1, 3, 8, 0)]:
        with pytest.raises(ValueError, match="positive"):
            e.forward(torch.zeros(shape, device=_dev()))
    # a constant frame is fine in the integer path (only the calibration observer rejects "all equal")
    q, _ = e.forward(torch.full((1, 3, 5, 7), 0.25, device=_dev()))
    _cmp("constant frame", q, O.forward(net, np.full((1, 3, 5, 7), 0.25, np.float32))["q_out"])


# ---------------------------------------------------------------- fused hidden trio (sesrq_trio.hip)

TRIO_SHAPES = [(1, 1, 1), (1, 7, 59), (1, 8, 60), (1, 9, 61), (2, 16, 119), (1, 17, 120), (1, 23, 121), (1, 64, 180),
               (3, 41, 250), (1, 130, 62), (2, 200, 33)]


@pytest.mark.parametrize("budget", [0, 1, 6], ids=["one-round", "one-wg-per-strip", "budget6"])
def test_trio_walk_shapes_and_chunking(budget):
    """Strips of 60 valid columns, steps of 8 rows, cold chunk starts, the steady multi-step walk (wg_budget makes a
    workgroup walk a whole strip even on small frames), frame borders inside the 3-layer halo: fused trio vs the
    per-layer kernels vs the C oracle."""
    from oracle import c_oracle as CO
    for seed, kind in enumerate(["sesr_x2", "nrdm", "sesr_x4"]):
        net = O.synth_net(kind, 40 + seed)
        et = sesrq.Engine(bundle_from_oracle(net), _dev(), wg_budget=budget)
        el = sesrq.Engine(bundle_from_oracle(net), _dev(), fuse_hidden=0, wg_budget=budget)
        assert et.layer_engines()[1:4] == ["mfma-trio-merged"] * 3 and et.launch_plan() == [(0, 1), (1, 3), (4, 1)]
        assert el.launch_plan() == [(k, 1) for k in range(5)]
        cin = net.layers[0].wq.shape[1]
        for (N, H, W) in TRIO_SHAPES:
            x = rand_frame((N, cin, H, W), 31 * H + W)
            xt = torch.from_numpy(x).to(_dev())
            q, y = et.forward(xt)
            q2, y2 = el.forward(xt)
            want = CO.forward(net, x)
            _cmp(f"{net.name} {N}x{H}x{W} budget {budget}: trio vs oracle", q, want["q_out"])
            _cmp(f"{net.name} {N}x{H}x{W} budget {budget}: per-layer vs oracle", q2, want["q_out"])
            _cmp("y", y, want["y"])
            # an already-quantised frame gives the same bytes (an upstream net's int8 output: test_config5_*)
            q4, _ = et.forward(torch.from_numpy(O.quantize_input(x, net.scale[0], net.zero[0])).to(_dev()))
            _cmp(f"{net.name} {N}x{H}x{W} budget {budget}: trio, int8 input", q4, want["q_out"])


def test_trio_with_separate_residual_tensor_and_odd_zero_points():
    """zero[1] != -128: the residual operand is its own tensor (layer 0 writes it); pad values of the inner layers
    differ per layer (zc = max(zero, -128)); zero points above -128 on the hidden domains."""
    net = O.synth_net("sesr_x2", 51)
    net.zero[1], net.zero[2], net.zero[3], net.zero[4] = -120, -101, -128, -77
    e = sesrq.Engine(bundle_from_oracle(net), _dev(), wg_budget=2)
    assert "mfma-trio-merged" in e.layer_engines() and e.launch_plan() == [(0, 1), (1, 3), (4, 1)]
    for (N, H, W) in [(1, 19, 70), (2, 33, 121)]:
        x = rand_frame((N, 3, H, W), H)
        want = O.forward(net, x)
        q, y = e.forward(torch.from_numpy(x).to(_dev()))
        _cmp("q_out", q, want["q_out"])


def test_trio_on_deeper_net_two_trios():
    """8-conv net (nrdm_6 shape): hidden layers 1-3 run as a trio with a plain third epilogue, 4-6 as the trio that
    merges the residual.  Parity unpinned w.r.t. the reference (no integer path at this depth): oracle vs HIP."""
    net = O.synth_net("nrdm", 61, n_blocks=6)
    e = sesrq.Engine(bundle_from_oracle(net), _dev(), wg_budget=3)
    assert e.launch_plan() == [(0, 1), (1, 3), (4, 3), (7, 1)], e.launch_plan()
    x = rand_frame((2, 3, 45, 130), 8)
    want = O.forward(net, x)
    q, y = e.forward(torch.from_numpy(x).to(_dev()))
    _cmp("q_out", q, want["q_out"])
    _cmp("y", y, want["y"])


# ---------------------------------------------------------------- full frames at the BASELINE sizes

def _full_frame_case(fixture, shape, seed, engines_expected, **kw):
    from oracle import c_oracle as CO
    fx, meta, net, _ = fixture_case(os.path.join(os.path.dirname(STAGE_FILES[0]), fixture))
    e = sesrq.Engine(bundle_from_oracle(net), _dev(), **kw)
    assert e.layer_engines() == engines_expected, e.layer_engines()
    x = rand_frame(shape, seed)
    q, _ = e.forward(torch.from_numpy(x).to(_dev()))
    want = CO.forward(net, x, want_f=False)["q_out"]
    _cmp(f"{fixture} {shape} whole frame vs C oracle", q, want)
    return e, x, q


def test_config2_full_frame_1080p_on_the_timed_kernels():
    """BASELINE config 2 on the bundle bench.py times (reference random-init x2 net calibrated by the reference):
    the WHOLE 1x3x1080x1920 -> 3x2160x3840 frame against the C oracle, borders (pad value zc,
    myQL/quan_func.py:351-356) and every chunk boundary of the full-chip grid included; the same frame on the
    per-layer kernels must give the same bytes."""
    names = ["mfma-f5-hybrid", "mfma-trio-merged", "mfma-trio-merged", "mfma-trio-merged", "mfma-h5-general"]
    e, x, q = _full_frame_case("sesr_x2_rand.crop.npz", (1, 3, 1080, 1920), 1, names)
    # round 4: the REFERENCE ITSELF was run on this very frame in the build container (tests/golden/make_golden.py --case
    # time_x2_1080p -> reference_x2_1080p.json): the whole int8 and fp32 4K frame against the reference's own output, by SHA-256
    import json
    ref = json.load(open(os.path.join(os.path.dirname(STAGE_FILES[0]), "reference_x2_1080p.json")))
    assert _sha(x) == ref["x_sha256"]
    q1, y1 = e.forward(torch.from_numpy(x).to(_dev()), want_q=True, want_f=True)
    assert _sha(q1.cpu().numpy()) == ref["out_q_sha256"] and _sha(q.cpu().numpy()) == ref["out_q_sha256"]
    assert _sha(y1.cpu().numpy()) == ref["out_f_sha256"]
    e2 = sesrq.Engine(e.bundle, _dev(), fuse_hidden=0)
    assert e2.layer_engines() == ["mfma-f5-hybrid", "mfma-h3-merged", "mfma-h3-merged", "mfma-h3-merged", "mfma-h5-general"]
    q2, _ = e2.forward(torch.from_numpy(x).to(_dev()))
    assert torch.equal(q, q2)


def test_config3_full_frame_nrdm3_540p():
    """BASELINE config 3: nrdm_3 (reference checkpoints -- nrdm_3_qat_G.pth, the one config 3 names, and nrdm_3_raw_G.pth --
    reference calibration) 1x3x540x960, whole frame."""
    _full_frame_case("nrdm_3_qat.crop.npz", (1, 3, 540, 960), 3, ["mfma-f5-merged"] + ["mfma-trio-merged"] * 3 + ["mfma-h5p-general"])
    _full_frame_case("nrdm_3.crop.npz", (1, 3, 540, 960), 3, ["mfma-f5-merged"] + ["mfma-trio-merged"] * 3 + ["mfma-h5p-merged"])


def test_batch_larger_than_one_chip_round():
    """N x strips exceeds the workgroup slots of the chip: every workgroup walks a whole strip (chunk = all tiles)."""
    from oracle import c_oracle as CO
    fx, meta, net, _ = fixture_case(os.path.join(os.path.dirname(STAGE_FILES[0]), "sesr_x2_rand.crop.npz"))
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = rand_frame((72, 3, 36, 1030), 5)          # 72 frames x 18 strips (60 columns) / 17 (64 columns) > 1024 slots
    q, _ = e.forward(torch.from_numpy(x).to(_dev()))
    want = CO.forward(net, x, want_f=False)["q_out"]
    _cmp("72 x 36 x 1030", q, want)


def test_forward_many_gives_sesrq_forward_bytes():
    """sesrq_forward_many (round 4): many independent frames, distinct caller buffers, several streams, ONE call -- per frame the bytes
    of sesrq_forward; windows that start anywhere in the cycle, more frames than the list holds, fp32 and int8 frames, an fp32 output."""
    net = O.synth_net("sesr_x4", 17)
    b = bundle_from_oracle(net)
    e = sesrq.Engine(b, _dev(), wg_budget=64)
    S, F = 3, 6
    xs = [torch.from_numpy(rand_frame((1, 1, 45, 130), 100 + k)).to(_dev()) for k in range(F)]
    want = [e.forward(x) for x in xs]
    streams = [torch.cuda.Stream(device=_dev()) for _ in range(S)]
    outs = [torch.zeros_like(want[0][0]) for _ in range(F)]
    outf = [torch.zeros_like(want[0][1]) for _ in range(F)]
    torch.cuda.synchronize()
    sub = e.submission(xs, outs, streams, outs_f=outf)
    sub.enqueue(F)
    torch.cuda.synchronize()
    for k in range(F):
        assert torch.equal(outs[k], want[k][0]) and torch.equal(outf[k], want[k][1]), k
    # a window in the middle of the cycle, longer than the list: frames 4, 5, 0, 1, ..., each on ITS stream of the cycle
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    sub.enqueue(2 * F + 1, first=4)
    torch.cuda.synchronize()
    for k in range(F):
        assert torch.equal(outs[k], want[k][0]), k
    # int8 frames (already q0), no fp32 output
    q0 = [torch.from_numpy(O.quantize_input(x.cpu().numpy(), net.scale[0], net.zero[0])).to(_dev()) for x in xs]
    outs8 = [torch.zeros_like(want[0][0]) for _ in range(F)]
    e.submission(q0, outs8, streams).enqueue(F)
    torch.cuda.synchronize()
    for k in range(F):
        assert torch.equal(outs8[k], want[k][0]), k
    # grouping: a workspace for G frames lets the library run up to G consecutive frames of a stream as the images of ONE launch sequence
    # (pointer table in the kernel arguments of the first and the last layer): same bytes, also for a remainder group and an fp32 output
    for kind, shape, G in (("sesr_x4", (1, 1, 45, 130), 4), ("sesr_x2", (1, 3, 33, 70), 8), ("nrdm", (1, 3, 40, 64), 3)):
        net2 = O.synth_net(kind, 23)
        e2 = sesrq.Engine(bundle_from_oracle(net2), _dev(), wg_budget=64)
        F2 = 14
        xs2 = [torch.from_numpy(rand_frame(shape, 300 + k)).to(_dev()) for k in range(F2)]
        want2 = [e2.forward(x) for x in xs2]
        o2 = [torch.zeros_like(want2[0][0]) for _ in range(F2)]
        f2 = [torch.zeros_like(want2[0][1]) for _ in range(F2)]
        st2 = streams[:2]
        torch.cuda.synchronize()
        sub2 = e2.submission(xs2, o2, st2, outs_f=f2, group=G)
        assert sub2.ws[0].numel() >= G * e2.workspace(1, shape[2], shape[3], 9).numel() - 4096
        sub2.enqueue(F2)                                  # 7 frames per stream: groups of G and a remainder
        torch.cuda.synchronize()
        for k in range(F2):
            assert torch.equal(o2[k], want2[k][0]) and torch.equal(f2[k], want2[k][1]), (kind, G, k)
        for o in o2:
            o.zero_()
        torch.cuda.synchronize()
        sub2.enqueue(F2 - 3, first=2)                    # a window inside the cycle: every frame still lands in ITS buffers
        torch.cuda.synchronize()
        for k in range(2, F2 - 1):
            assert torch.equal(o2[k], want2[k][0]), (kind, G, k)
        if kind == "sesr_x2":      # already-quantised int8 frames through the table too
            q02 = [torch.from_numpy(O.quantize_input(x.cpu().numpy(), net2.scale[0], net2.zero[0])).to(_dev()) for x in xs2]
            for o in o2:
                o.zero_()
            torch.cuda.synchronize()
            e2.submission(q02, o2, st2, group=G).enqueue(F2)
            torch.cuda.synchronize()
            for k in range(F2):
                assert torch.equal(o2[k], want2[k][0]), (kind, "int8", k)
    # the x2 anchor add reads each frame's OWN input through the table
    neta = O.synth_net("sesr_x2", 3)
    ea = sesrq.Engine(bundle_from_oracle(neta), _dev(), anchor_add=True)
    xa = [torch.from_numpy(rand_frame((1, 3, 20, 50), 400 + k)).to(_dev()) for k in range(4)]
    wa = [ea.forward(x) for x in xa]
    oa = [torch.zeros_like(wa[0][0]) for _ in range(4)]
    fa = [torch.zeros_like(wa[0][1]) for _ in range(4)]
    torch.cuda.synchronize()
    ea.submission(xa, oa, streams[:1], outs_f=fa, group=4).enqueue(4)
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(oa[k], wa[k][0]) and torch.equal(fa[k], wa[k][1]), k
    # errors: a bad frame is reported with its index
    import ctypes as C
    io = (_lib.FrameIO * 2)(_lib.FrameIO(xs[0].data_ptr(), outs[0].data_ptr(), None), _lib.FrameIO(xs[1].data_ptr(), None, None))
    ws = e.workspace(1, 45, 130, 0)
    rc = _lib.lib().sesrq_forward_many(e._h, io, 2, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(ws.data_ptr()), ws.numel(),
                                       (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream), 1, 1)
    assert rc != 0 and "frame 1" in _lib.last_error()
    torch.cuda.synchronize()
    # round 5 (ABI v4): the group size is an ARGUMENT, validated -- a workspace that happens to hold more frames changes nothing
    big = e.workspace(8, 45, 130, 7)
    st1 = (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream)
    io2 = (_lib.FrameIO * 4)(*[_lib.FrameIO(xs[k].data_ptr(), outs[0].data_ptr(), None) for k in range(4)])      # ONE output buffer, stream-ordered
    for o in outs:
        o.zero_()
    assert _lib.lib().sesrq_forward_many(e._h, io2, 4, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(big.data_ptr()), big.numel(), st1, 1, 1) == 0
    torch.cuda.synchronize()
    assert torch.equal(outs[0], want[3][0])          # group 1: legal, the last frame of the stream wins
    # ... and grouped frames that share an output buffer are REFUSED (they would be written concurrently), nothing of the group enqueued
    outs[0].zero_()
    torch.cuda.synchronize()
    rc = _lib.lib().sesrq_forward_many(e._h, io2, 4, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(big.data_ptr()), big.numel(), st1, 1, 4)
    assert rc != 0 and "share a launch sequence" in _lib.last_error(), _lib.last_error()
    torch.cuda.synchronize()
    assert int(outs[0].abs().sum()) == 0
    with pytest.raises(ValueError, match="share an output buffer"):
        e.submission(xs[:4], [outs[0]] * 4, streams[:1], group=4)
    for bad_group, msg in ((0, "group must be"), (9, "group must be")):
        assert _lib.lib().sesrq_forward_many(e._h, io2, 4, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(big.data_ptr()), big.numel(), st1, 1, bad_group) != 0
        assert msg in _lib.last_error()
    small = e.workspace(1, 45, 130, 8)
    assert _lib.lib().sesrq_forward_many(e._h, io2, 4, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(small.data_ptr()), small.numel(), st1, 1, 2) != 0
    assert "workspace too small for this group" in _lib.last_error()
    ed = sesrq.Engine(b, _dev(), engine=_lib.ENGINE_DOT4)
    assert _lib.lib().sesrq_forward_many(ed._h, io2, 4, _lib.F32, 1, 45, 130, (C.c_void_p * 1)(big.data_ptr()), big.numel(), st1, 1, 2) != 0
    assert "MFMA first- and last-layer kernels" in _lib.last_error()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_forward_many_workers_wake_from_sleep():
    """The submission threads of sesrq_forward_many spin for 2 ms after a batch and then sleep on a condition variable: calls spaced
    around that boundary (0 ... 5 ms apart) must hand every batch over -- a lost wake-up would leave the caller spinning (the timeout)."""
    import time
    net = O.synth_net("sesr_x4", 29)
    e = sesrq.Engine(bundle_from_oracle(net), _dev(), wg_budget=64)
    S, F = 3, 12
    xs = [torch.from_numpy(rand_frame((1, 1, 24, 64), 500 + k)).to(_dev()) for k in range(F)]
    want = [e.forward(x, want_f=False)[0] for x in xs]
    streams = [torch.cuda.Stream(device=_dev()) for _ in range(S)]
    outs = [torch.zeros_like(want[0]) for _ in range(F)]
    sub = e.submission(xs, outs, streams)
    torch.cuda.synchronize()
    for it in range(60):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        sub.enqueue(F)                     # 4 launch sequences per stream: the pooled path
        torch.cuda.synchronize()
        assert all(torch.equal(outs[k], want[k]) for k in range(F)), it
        time.sleep((0.0, 0.0015, 0.0019, 0.002, 0.0021, 0.0025, 0.005)[it % 7])


def test_side_stream_with_non_contiguous_input():
    """forward(stream=s): the side stream is ordered behind the producer of x, temporaries are recorded on it."""
    net = O.synth_net("sesr_x2", 9)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    base = torch.from_numpy(rand_frame((2, 3, 50, 140), 3)).to(_dev())
    xnc = base.transpose(2, 3).contiguous().transpose(2, 3)       # same values, non-contiguous view
    assert not xnc.is_contiguous()
    s = torch.cuda.Stream(device=_dev())
    q, y = e.forward(xnc, stream=s)
    s.synchronize()
    want = O.forward(net, base.cpu().numpy())
    _cmp("q_out", q, want["q_out"])
    _cmp("y", y, want["y"])


def test_captured_graph_replays_the_same_bits():
    """Engine.capture: the forward (and a two-net chain) as a HIP graph -- same kernels, so the same bytes as the eager call;
    a new frame written into the captured input buffer is what the next replay processes."""
    net = O.synth_net("sesr_x2", 3)
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    x = torch.from_numpy(rand_frame((2, 3, 37, 150), 1)).to(_dev())
    g = e.capture(x, want_q=True, want_f=True)
    q0, y0 = e.forward(x)
    q, y = g.replay()
    torch.cuda.synchronize()
    assert torch.equal(q, q0) and torch.equal(y, y0)
    x2 = rand_frame((2, 3, 37, 150), 2)
    x.copy_(torch.from_numpy(x2))
    q, _ = g.replay()
    _cmp("graph replay on a new frame", q, O.forward(net, x2)["q_out"])
    side = torch.cuda.Stream(device=_dev())
    side.wait_stream(torch.cuda.current_stream(_dev()))
    with torch.cuda.stream(side):
        q, _ = g.replay()                                  # on another stream
    side.synchronize()
    _cmp("graph replay on a side stream", q, O.forward(net, x2)["q_out"])
    # chain: nrdm_6 -> SESR-x2 with the int8 hand-off, one graph
    nr = O.synth_net("nrdm", 21, n_blocks=6)
    e1 = sesrq.Engine(bundle_from_oracle(nr), _dev())
    e2 = sesrq.Engine(bundle_from_oracle(net), _dev(), upstream=bundle_from_oracle(nr))
    xc = torch.from_numpy(rand_frame((1, 3, 30, 70), 4)).to(_dev())
    gc = e1.capture(xc, downstream=[e2])
    qa, _ = e2.forward(e1.forward(xc, want_f=False)[0], want_f=False)
    qb, _ = gc.replay()
    torch.cuda.synchronize()
    assert torch.equal(qa, qb) and tuple(qb.shape) == (1, 3, 60, 140)
    with pytest.raises(ValueError, match="contiguous"):
        e.capture(x[:, :, :, ::2])


def test_overflow_counters_mirror_the_reference_prints():
    """sesrq_taps.overflow counts PE sums outside the 18-bit accumulator range before saturation -- the events the
    reference prints as max_overflow / min_overflow (myQL/quan_func.py:358-361); golden: the saturating-weight
    fixtures made the reference print them, the ordinary ones did not."""
    for tag, expect in (("sesr_x4.crop.npz", False), ("sesr_x4.satw.npz", True)):
        fx, meta, net, x = fixture_case(os.path.join(os.path.dirname(STAGE_FILES[0]), tag))
        e = sesrq.Engine(bundle_from_oracle(net), _dev())
        res = e.forward_debug(torch.from_numpy(x).to(_dev()), pe=False, overflow=True)
        ovf = res["overflow"].cpu().numpy()
        st = O.forward(net, x, keep=True)
        for k in range(net.L if hasattr(net, "L") else len(net.layers)):
            raw = st.get(f"pe_raw{k}")
            if raw is not None:
                assert ovf[k, 0] == int((raw > 131071).sum()) and ovf[k, 1] == int((raw < -131072).sum())
        assert bool(ovf.any()) == expect, (tag, ovf)
        _cmp("q_out with the counter tap", res["q_out"], st["q_out"])


def test_per_channel_weight_scales_vs_oracle():
    """Round 5: per-output-channel requant constants (sesrq_layer_desc.M_oc / n_oc) -- NOT in the reference, PARITY
    UNPINNED: HIP against the numpy oracle's definition only.  Nets derived per channel from the reference's float weights (golden
    params) and the reference's calibrated activation domains; a per-channel layer runs on the dot4 kernels (engine name says so), its
    per-tensor neighbours keep their MFMA kernels; the per-tensor bundle of the same weights is a different result (the option is live)."""
    from conftest import GOLDEN, load_fixture
    for case, ps in (("sesr_x4", 4), ("nrdm_3", 1), ("sesr_x2_rand", 2), ("sesr_x2_rand_nat", 2)):
        p, pm = load_fixture(os.path.join(GOLDEN, f"{case}.params.npz"))
        Wf, bf = [p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)]
        net = O.derive_net(Wf, bf, pm["scale"], pm["zero"], ps, per_channel=True)
        e = sesrq.Engine(bundle_from_oracle(net), _dev())
        assert all(n.startswith("dot4-") and n.endswith("-perchannel") for n in e.layer_engines()), e.layer_engines()
        cin = Wf[0].shape[1]
        for (N, H, W) in ((1, 24, 40), (2, 9, 33), (1, 1, 1)):
            x = rand_frame((N, cin, H, W), 7 * H + W)
            want = O.forward(net, x, keep=True)
            q, y = e.forward(torch.from_numpy(x).to(_dev()))
            _cmp(f"{case} per-channel {N}x{H}x{W} q", q, want["q_out"])
            _cmp(f"{case} per-channel {N}x{H}x{W} y", y, want["y"])
        res = e.forward_debug(torch.from_numpy(x).to(_dev()), pe=True, special=True)      # the taps too (N == 1 case left in x)
        for k in range(5):
            _cmp(f"input{k}", res[f"input{k}"], want[f"input{k}"])
            _cmp(f"pe_add{k}", res[f"pe_add{k}"], want[f"pe_add{k}"])
        _cmp("shortcut", res["shortcut"], want["shortcut"])
        # mixed: only the middle hidden layer per channel -> that layer on dot4, no fused trio, the others on their MFMA kernels
        tens = O.derive_net(Wf, bf, pm["scale"], pm["zero"], ps)
        mixed = O.Net(**{**tens.__dict__, "layers": [net.layers[k] if k == 2 else tens.layers[k] for k in range(5)]})
        em = sesrq.Engine(bundle_from_oracle(mixed), _dev())
        names = em.layer_engines()
        assert names[2].endswith("-perchannel") and all(n.startswith("mfma-") for i, n in enumerate(names) if i != 2), names
        x = rand_frame((1, cin, 31, 70), 11)
        wm = O.forward(mixed, x)
        qm, ym = em.forward(torch.from_numpy(x).to(_dev()))
        _cmp(f"{case} mixed q", qm, wm["q_out"])
        _cmp(f"{case} mixed y", ym, wm["y"])
        qt, _ = sesrq.Engine(bundle_from_oracle(tens), _dev()).forward(torch.from_numpy(x).to(_dev()))
        assert not torch.equal(qt, qm), "per-channel constants change the result: the option is live"
    # grouping needs the MFMA first / last layers: refused for a per-channel net, with the reason
    import ctypes as C
    ws = e.workspace(2, 8, 8, 3)
    xs = torch.zeros((1, cin, 8, 8), device=_dev())
    out = torch.zeros(e.out_shape(1, 8, 8), dtype=torch.int8, device=_dev())
    io = (_lib.FrameIO * 2)(_lib.FrameIO(xs.data_ptr(), out.data_ptr(), None), _lib.FrameIO(xs.data_ptr(), out.data_ptr(), None))
    rc = _lib.lib().sesrq_forward_many(e._h, io, 2, _lib.F32, 1, 8, 8, (C.c_void_p * 1)(ws.data_ptr()), ws.numel(),
                                       (C.c_void_p * 1)(torch.cuda.current_stream().cuda_stream), 1, 2)
    assert rc != 0 and "MFMA first- and last-layer kernels" in _lib.last_error()


def _big_cases():
    from conftest import big_cases
    return big_cases()


@pytest.mark.parametrize("rec", [r for _, r in _big_cases()], ids=[r["case"] for _, r in _big_cases()])
def test_baseline_size_natural_frames_match_the_reference(rec):
    """Round 5: natural-ish frames at BASELINE sizes (SESR-x2 1080p -> 4K = config 2, nrdm_3 540p = config 3, SESR-x4 540p = config 4's frame)
    that the REFERENCE itself ran through its sim path with a calibration it made on a natural frame (zero_0 < -128; the x2 net's 18-bit
    PE clamp fires: it printed max_overflow).  The HIP path -- the production launch plan, each output kind -- must reproduce the SHA-256 of
    the reference's int8 and fp32 results (tests/golden/*_nat.big.json; generator: make_golden.py)."""
    from conftest import GOLDEN, big_input
    fx, meta, net, _ = fixture_case(os.path.join(GOLDEN, rec["bundle"]))
    assert meta["zero"][0] < -128
    x = torch.from_numpy(big_input(rec)).to(_dev())
    e = sesrq.Engine(bundle_from_oracle(net), _dev())
    assert any("trio" in n for n in e.layer_engines())
    q, y = e.forward(x)
    assert list(q.shape) == rec["out_shape"]
    assert _sha(q.cpu().numpy()) == rec["out_q_sha256"], "int8 frame differs from the reference's"
    assert _sha(y.cpu().numpy()) == rec["out_f_sha256"], "fp32 frame differs from the reference's"
    q1, _ = e.forward(x, want_f=False)
    _, y1 = e.forward(x, want_q=False)
    assert torch.equal(q1, q) and torch.equal(y1, y), "the int8-only / fp32-only kernel flavours give the same frames"
    # the bench plan too: half-chip launches (wg_budget 512-like: two slots per CU)
    e2 = sesrq.Engine(bundle_from_oracle(net), _dev(), wg_budget=2 * torch.cuda.get_device_properties(_dev()).multi_processor_count)
    q2, _ = e2.forward(x, want_f=False)
    assert torch.equal(q2, q)
