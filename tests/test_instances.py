"""Every kernel instantiation the library can select is RUN and CHECKED here (-m gpu).

Round 4 shipped 9.5 M wrong bytes in an instantiation (the int8-only last layer) that 168 green tests never selected.  The library
now enumerates what it can launch (sesrq_instance_count / _name / _launches: every launch goes through launch_kernel<KERN>, which
registers KERN when the library is loaded), and this file drives a matrix of cases until every entry has been launched:

  tier A  the reference-made crops (tests/golden/*.{crop,zeros,satw,satw_zeros,stim}.npz, incl. the natural-frame fixtures calibrated by the
          reference) under every option that changes the kernel selection without changing the result: launch plans (dot4 / per-layer
          MFMA / fused trio), force_general, PE widths 19 / 21 where no clamp fires, sesrq_options.reduced_forms (the trio's five
          epilogue modes, the last layer's three requant forms), the three output kinds, fp32 and int8 (the reference's own input.0)
          frames, grouped pointer-table launches, the debug forward's taps -- all compared with what THE REFERENCE produced;
  tier B  seeded synthetic nets for what no reference net selects (2- and 4-channel frames, int8 hand-off from an upstream net, 5x5 hidden
          layers, 16-channel outputs with PixelShuffle 1 / 2, crafted saturation patterns for the hybrid kernels, 8-conv nets with two
          trios) -- compared with the numpy oracle, which tier A's fixtures pin.

The last test fails for any instantiation that no case launched, by name, and writes the table (instance, launches, pinned by) to
gpurun_out/instance_coverage.txt.
"""
import os

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_files
from helpers import bundle_from_oracle, fixture_case, rand_frame
from oracle import sesrq_oracle as O
import sesrq
from sesrq import _lib

pytestmark = pytest.mark.gpu

DEV = None
REF_PINNED = set()       # instances launched by a case whose expected result came from the reference
ORACLE_ONLY = set()      # ... by a case checked against the oracle only


def dev():
    global DEV
    if DEV is None:
        assert torch.cuda.is_available(), "GPU tests need a HIP device"
        DEV = torch.device("cuda:0")
    return DEV


class Track:
    """Attributes the instances launched inside the block to a tier."""

    def __init__(self, pinned):
        self.pinned = pinned

    def __enter__(self):
        self.before = _lib.instances()
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            torch.cuda.synchronize()
            after = _lib.instances()
            hit = {k for k, v in after.items() if v > self.before.get(k, 0)}
            (REF_PINNED if self.pinned else ORACLE_ONLY).update(hit)
        return False


def shuffle(q5, r):
    return O.pixel_shuffle(q5, r)


def eq(name, got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    assert got.shape == want.shape, f"{name}: shape {got.shape} != {want.shape}"
    bad = np.argwhere(got != want)
    if len(bad):
        i = tuple(bad[0])
        raise AssertionError(f"{name}: {len(bad)} mismatches, first at {i}: got {got[i]} want {want[i]}")


OUT_KINDS = [(True, False), (False, True), (True, True)]      # (want_q, want_f): the three output kinds of the boundary


def run_all_kinds(tag, e, inputs, want_q, want_f):
    """inputs: [(label, tensor)]; every output kind; returns nothing, raises on the first difference."""
    for lbl, xt in inputs:
        for wq, wf in OUT_KINDS:
            q, y = e.forward(xt, want_q=wq, want_f=wf)
            if wq:
                eq(f"{tag} [{lbl}, q={wq}, f={wf}] q_out", q, want_q)
            if wf:
                eq(f"{tag} [{lbl}, q={wq}, f={wf}] y", y, want_f)


RF_ALL = 63
# reduced_forms masks: trio modes 0 / 1 / 3 / 7 / 15 (bits 1, 2, 4, 8) and the last layer's forms 1 / 2 / none (bits 16, 32)
RF_MASKS = [0, 1, RF_ALL & ~4 & ~8, RF_ALL & ~8, RF_ALL, RF_ALL & ~16, RF_ALL & ~16 & ~32]

CROPS = [f for f in golden_files() if f.endswith((".crop.npz", ".zeros.npz", ".satw.npz", ".satw_zeros.npz", ".stim.npz"))]


def plan_variants():
    v = [dict(), dict(fuse_hidden=0), dict(engine=_lib.ENGINE_DOT4), dict(force_general=True)]
    v += [dict(reduced_forms=m) for m in RF_MASKS] + [dict(reduced_forms=m, fuse_hidden=0) for m in (0, RF_ALL & ~16)]
    return v


@pytest.mark.parametrize("path", CROPS, ids=[os.path.basename(p)[:-4] for p in CROPS])
def test_reference_crops_under_every_selection_option(path):
    """Tier A.  The same reference-made frame, bundle and expected tensors under every option that only changes WHICH kernel computes them."""
    fx, meta, net, x = fixture_case(path)
    r = net.pixel_shuffle
    want_q, want_f = shuffle(fx["input5"], r), fx["out"]
    xt = torch.from_numpy(x).to(dev())
    q0 = torch.from_numpy(fx["input0"]).to(dev())              # the reference's own input.0.pt: the int8 entry of the boundary
    inputs = [("f32", xt), ("i8", q0)]
    b = bundle_from_oracle(net)
    for kw in plan_variants():
        with Track(pinned=True):
            e = sesrq.Engine(b, dev(), **kw)
            run_all_kinds(f"{meta['case']}.{meta['tag']} {kw}", e, inputs, want_q, want_f)
    # other PE widths: where no 18-/20-bit clamp fires on this frame (the oracle says so: same result at 19 / 21 bits) the reference's
    # tensors stay the expected ones and the run-time-bounds instantiations (GEN_ANY) meet reference-made data
    wide = O.Net(**{**net.__dict__, "acc_bits": 19, "add_bits": 21})
    ow = O.forward(wide, x)
    pinned = np.array_equal(ow["q_out"], want_q)
    bw = bundle_from_oracle(wide)
    for kw in (dict(), dict(fuse_hidden=0), dict(force_general=True), dict(engine=_lib.ENGINE_DOT4), dict(force_general=True, engine=_lib.ENGINE_DOT4)):
        with Track(pinned=pinned):
            e = sesrq.Engine(bw, dev(), **kw)
            run_all_kinds(f"{meta['case']}.{meta['tag']} 19/21 bits {kw}", e, inputs, ow["q_out"], ow["y"])
    # the debug forward: PE taps written by the per-PE MFMA kernels (GEN_TAP) / the dot4 kernels, and the NHWC16 unpack of the input taps
    # (acts=False keeps layer 0 / L-2 on their MFMA tap kernels; the int8 frame = the reference's input.0 reaches the first layer's int8 tap instance)
    for kw, acts, xin in ((dict(fuse_hidden=0), True, xt), (dict(fuse_hidden=0), False, xt), (dict(fuse_hidden=0), False, q0), (dict(engine=_lib.ENGINE_DOT4), True, xt)):
        with Track(pinned=True):
            e = sesrq.Engine(b, dev(), **kw)
            res = e.forward_debug(xin, pe=True, acts=acts, special=acts)
            for k in range(5):
                eq(f"pe_out{k}", res[f"pe_out{k}"][0], fx[f"pe_out{k}"])
                eq(f"pe_add{k}", res[f"pe_add{k}"], fx[f"pe_add{k}"])
                if acts:
                    eq(f"input{k}", res[f"input{k}"], fx[f"input{k}"])
            if acts:
                eq("shortcut", res["shortcut"], fx["shortcut"])
                eq("input4_special", res["input4_special"], fx["input4_special"])
            eq("q_out(debug)", res["q_out"], want_q)
            eq("y(debug)", res["y"], want_f)


@pytest.mark.parametrize("path", [p for p in CROPS if p.endswith(".crop.npz")], ids=[os.path.basename(p)[:-4] for p in CROPS if p.endswith(".crop.npz")])
def test_reference_crops_as_grouped_launches(path):
    """Tier A, sesrq_forward_many: G frames of a stream as the images of one launch sequence (pointer table in the kernel arguments of the
    first and the last layer), every output kind, fp32 and int8 frames -- each image must be the reference's tensors."""
    fx, meta, net, x = fixture_case(path)
    r = net.pixel_shuffle
    want_q, want_f = shuffle(fx["input5"], r), fx["out"]
    b = bundle_from_oracle(net)
    streams = [torch.cuda.Stream(device=dev()) for _ in range(2)]
    for kw in (dict(), dict(fuse_hidden=0), dict(reduced_forms=RF_ALL & ~16), dict(reduced_forms=0)):
        e = sesrq.Engine(b, dev(), **kw)
        for src in ("f32", "i8"):
            for G in (2, 3):
                F = 2 * G * 2
                frames = [(torch.from_numpy(x) if src == "f32" else torch.from_numpy(fx["input0"])).to(dev()).clone() for _ in range(F)]
                for wq, wf in OUT_KINDS:
                    oq = [torch.zeros(want_q.shape, dtype=torch.int8, device=dev()) for _ in range(F)]
                    of = [torch.zeros(want_f.shape, dtype=torch.float32, device=dev()) for _ in range(F)] if wf else None
                    torch.cuda.synchronize()
                    with Track(pinned=True):
                        sub = e.submission(frames, oq if wq else None, streams, outs_f=of, group=G)
                        sub.enqueue(F)
                        torch.cuda.synchronize()
                        for k in range(F):
                            if wq:
                                eq(f"group {G} {src} frame {k} q", oq[k], want_q)
                            if wf:
                                eq(f"group {G} {src} frame {k} y", of[k], want_f)


# ------------------------------------------------------------------------------------------------ tier B: synthetic nets vs the oracle
def pick_Mn(rng, target, want_form=None, last=False):
    """(M, n) near `target` whose load-time proof gives the wanted requant form (None = any)."""
    lib = _lib.lib()
    for _ in range(400):
        M, n = O.qconst(float(target * rng.uniform(0.6, 1.6)))
        if want_form is None or lib.sesrq_requant_form(M, n, 1 if last else 0) == want_form:
            return M, n
    raise AssertionError(f"no (M, n) near {target} with requant form {want_form}")


def craft_net(seed, cin, cout, ps, ks=(5, 3, 3, 3, 5), risky=None, zeros=None, forms=None, bits=(18, 20), wide_all=False):
    """A seeded integer bundle with a chosen saturation pattern.
    risky: {layer: (PEs, output channels)} -- those (oc, PE) weight groups are wide enough for the 18-bit PE clamp to be possible, every
    other group is provably safe (128 * sum|w| <= 131071): picks merged / hybrid / general kernels per layer.
    forms: {layer: requant form wanted from the load-time proof}."""
    rng = np.random.default_rng(seed)
    L = len(ks)
    chans = [cin] + [16] * (L - 1)
    layers = []
    for k in range(L):
        ic, oc, kk = chans[k], (cout if k == L - 1 else 16), ks[k]
        per_group = kk * kk * max(1, (ic + 3) // 4) if k > 0 else kk * kk      # weights per (oc, PE); first layer: one channel per PE
        wsafe = max(1, min(127, 1023 // per_group))
        w = rng.integers(-wsafe, wsafe + 1, size=(oc, ic, kk, kk))
        if wide_all:
            w = rng.choice(np.array([-128, -100, 90, 127]), size=w.shape)
        pes, ocs = (risky or {}).get(k, ((), ()))
        for p in pes:
            for o in ocs:
                if o < oc:
                    w[o, p::4] = rng.choice(np.array([-128, -110, 100, 127]), size=w[o, p::4].shape)
        fan = ic * kk * kk
        wr = float(np.sqrt(np.mean(w.astype(np.float64) ** 2))) + 1e-9
        tgt = 60.0 / (np.sqrt(fan) * wr * 74.0)
        M, n = pick_Mn(rng, tgt, (forms or {}).get(k), last=(k == L - 1))
        ac = rng.integers(-6000, 6000, oc).astype(np.int32)
        layers.append(O.Layer(wq=w.astype(np.int8), add_const=ac, M=M, n=n, relu=(k != L - 1)))
    zero = list(zeros) if zeros is not None else [-128] * (L + 1)
    scale = [float(s) for s in rng.uniform(0.003, 0.04, L + 1)]
    scale[0] = 1.0 / 255.0
    M_res, n_res = O.qconst(float(rng.uniform(0.2, 0.9)))
    return O.Net(layers=layers, scale=scale, zero=zero, M_res=M_res, n_res=n_res, pixel_shuffle=ps, acc_bits=bits[0], add_bits=bits[1],
                 name=f"craft{seed}")


def check_vs_oracle(tag, net, kws, sizes=((1, 21, 70), (2, 9, 33)), int8_too=True, upstream=None, seed=3):
    cin = net.layers[0].wq.shape[1]
    b = bundle_from_oracle(net)
    for kw in kws:
        e = sesrq.Engine(b, dev(), upstream=bundle_from_oracle(upstream) if upstream is not None else None, **kw)
        for (N, H, W) in sizes:
            if upstream is not None:
                # an upstream net's int8 output frame in its own output domain; the oracle takes the float hand-off
                rng = np.random.default_rng(seed + H)
                qin = rng.integers(-128, 128, size=(N, cin, H, W)).astype(np.int8)
                x = ((qin.astype(np.float32) - np.float32(upstream.zero[upstream.L])) * np.float32(upstream.scale[upstream.L])).astype(np.float32)
                inputs = [("i8d", torch.from_numpy(qin).to(dev()))]
            else:
                x = rand_frame((N, cin, H, W), 100 * seed + H * W)
                inputs = [("f32", torch.from_numpy(x).to(dev()))]
                if int8_too:
                    inputs.append(("i8", torch.from_numpy(O.quantize_input(x, net.scale[0], net.zero[0])).to(dev())))
            want = O.forward(net, x)
            with Track(pinned=False):
                run_all_kinds(f"{tag} {kw} {N}x{H}x{W}", e, inputs, want["q_out"], want["y"])
    return b


PLANS = (dict(), dict(fuse_hidden=0), dict(engine=_lib.ENGINE_DOT4), dict(force_general=True))
ODD = [-140, -120, -131, -128, -150, -119]      # zero points: z0 < -128, z1 > -128 (separate residual tensor), mixed


def test_first_layer_instances_vs_oracle():
    """Tier B: channel counts 1 / 2 / 3 / 4 x {merged, hybrid (dense / sparse by clamp register), general, run-time bounds} x {fp32, int8,
    int8 hand-off} x {with / without the separate residual tensor}."""
    up = craft_net(90, 3, 3, 1)                       # an upstream net: only its output domain matters
    for cin in (1, 2, 3, 4):
        cout, ps = (cin * 4, 2)
        pats = [("merged", None), ("general", {0: ((0, 1, 2, 3), range(16))})]
        pats += [(f"hybrid pe{p}", {0: ((p,), range(16))}) for p in range(min(cin, 3))]
        if cin == 3:      # sparse hybrid: the channels that can saturate all in accumulator register i (channels 4i .. 4i+3) -> RR = i
            pats += [(f"hybrid reg{i}", {0: ((1,), range(4 * i, 4 * i + 4))}) for i in range(4)]
        for name, risky in pats:
            for zeros in (None, ODD):
                net = craft_net(100 + cin, cin, cout, ps, risky=risky, zeros=zeros)
                check_vs_oracle(f"first layer cin={cin} {name} zeros={'odd' if zeros else '-128'}", net, PLANS[:2] + (dict(reduced_forms=0),))
                check_vs_oracle(f"first layer cin={cin} {name} int8 hand-off", net, PLANS, upstream=up)
                if name in ("merged", "general"):      # the debug forward's first-layer tap kernel on an int8 hand-off frame
                    rng = np.random.default_rng(cin)
                    qin = rng.integers(-128, 128, size=(1, cin, 13, 37)).astype(np.int8)
                    xf = ((qin.astype(np.float32) - np.float32(up.zero[up.L])) * np.float32(up.scale[up.L])).astype(np.float32)
                    st = O.forward(net, xf, keep=True)
                    with Track(pinned=False):
                        res = sesrq.Engine(bundle_from_oracle(net), dev(), fuse_hidden=0, upstream=bundle_from_oracle(up)).forward_debug(
                            torch.from_numpy(qin).to(dev()), pe=True, acts=False)
                        eq("pe_out0 (hand-off)", res["pe_out0"][0], st["pe_out0"])
                        eq("q_out (hand-off)", res["q_out"], st["q_out"])
        for zeros in (None, ODD):      # run-time accumulator bounds (GEN_ANY)
            net = craft_net(120 + cin, cin, cout, ps, zeros=zeros, bits=(17, 19), wide_all=True)
            check_vs_oracle(f"first layer cin={cin} 17/19 bits", net, PLANS[:3])
            check_vs_oracle(f"first layer cin={cin} 17/19 bits int8 hand-off", net, PLANS[:3], upstream=up)


def test_hidden_layer_instances_vs_oracle():
    """Tier B: 3x3 and 5x5 hidden layers (mid and residual-merging) x {merged, one risky PE, general, run-time bounds, taps}."""
    for ks in ((5, 3, 3, 3, 5), (5, 5, 5, 5, 5), (3, 3, 5, 3, 3)):
        for name, risky in (("merged", None), ("hybrid", {1: ((2,), range(16)), 2: ((0,), (3, 7)), 3: ((1,), range(16))}),
                            ("general", {1: ((0, 3), range(16)), 2: ((1, 2), range(16)), 3: ((0, 1, 2, 3), range(16))})):
            for zeros in (None, ODD):
                net = craft_net(200 + len(name), 3, 12, 2, ks=ks, risky=risky, zeros=zeros)
                b = check_vs_oracle(f"hidden ks={ks} {name}", net, PLANS)
                with Track(pinned=False):      # PE taps of the per-PE kernels against the oracle's stage tensors
                    x = rand_frame((1, 3, 13, 37), 5)
                    st = O.forward(net, x, keep=True)
                    for kw, special in ((dict(fuse_hidden=0), True), (dict(fuse_hidden=0), False), (dict(engine=_lib.ENGINE_DOT4), True)):
                        res = sesrq.Engine(b, dev(), **kw).forward_debug(torch.from_numpy(x).to(dev()), pe=True, acts=special, special=special)
                        for k in range(5):
                            eq(f"pe_out{k}", res[f"pe_out{k}"][0], st[f"pe_out{k}"])
                            eq(f"pe_add{k}", res[f"pe_add{k}"], st[f"pe_add{k}"])
                        if special:
                            eq("shortcut", res["shortcut"], st["shortcut"])
                            eq("input4_special", res["input4_special"], st["input4_special"])
        net = craft_net(230, 3, 12, 2, ks=ks, bits=(17, 19), wide_all=True, zeros=ODD)
        check_vs_oracle(f"hidden ks={ks} 17/19 bits", net, PLANS[:3])


def test_last_layer_instances_vs_oracle():
    """Tier B: output widths 3 (pe-split) / 8 / 12 / 16 x PixelShuffle 1 / 2 / 4 x {merged, hybrid, general, run-time bounds} x the three
    requant forms of the output layer x the three output kinds."""
    shapes = [(3, 3, 1), (1, 4, 2), (2, 8, 2), (3, 12, 2), (4, 16, 2), (1, 16, 4), (3, 12, 1), (4, 16, 1), (2, 8, 1)]
    for cin, cout, ps in shapes:
        for name, risky in (("merged", None), ("hybrid", {4: ((1,), range(16))}), ("general", {4: ((0, 2), range(16))})):
            for form in (1, 2, None):
                net = craft_net(300 + cout + ps, cin, cout, ps, risky=risky, forms={4: form} if form else None)
                kws = [dict(), dict(reduced_forms=RF_ALL & ~16), dict(reduced_forms=RF_ALL & ~16 & ~32), dict(engine=_lib.ENGINE_DOT4)]
                check_vs_oracle(f"last layer {cin}->{cout} ps{ps} {name} form {form}", net, kws, int8_too=False)
            net = craft_net(320 + cout, cin, cout, ps, risky=risky, zeros=ODD)      # zero[L] != -128: the general store
            check_vs_oracle(f"last layer {cin}->{cout} ps{ps} {name} odd zeros", net, [dict(), dict(engine=_lib.ENGINE_DOT4)], int8_too=False)
        net = craft_net(340 + cout, cin, cout, ps, bits=(17, 19), wide_all=True)
        check_vs_oracle(f"last layer {cin}->{cout} ps{ps} 17/19 bits", net, PLANS[:3], int8_too=False)
    # 3x3 output layers (dot4 only): every padded width
    for cout in (3, 8, 12, 16):
        for risky in (None, {4: ((0, 1), range(16))}):
            check_vs_oracle(f"3x3 output layer cout={cout}", craft_net(360 + cout, 3, cout, 1, ks=(3, 3, 3, 3, 3), risky=risky), PLANS[:3], int8_too=False)
            check_vs_oracle(f"3x3 first layer, int8 hand-off, cout={cout}", craft_net(360 + cout, 3, cout, 1, ks=(3, 3, 3, 3, 3), risky=risky), PLANS[2:],
                            upstream=craft_net(90, 3, 3, 1))
            check_vs_oracle(f"5x5 everywhere cout={cout}", craft_net(370 + cout, 3, cout, 1, ks=(5, 5, 5, 5, 5), risky=risky), [dict(engine=_lib.ENGINE_DOT4)], int8_too=False)


def test_anchor_add_instances():
    """The x2 anchor add of the reference's eval loop (test.py:148-155: gfake + inps_x2) with the fp32 frame out: round 5's store flavour for
    the reference's one anchor topology (3 -> 12 channels, PixelShuffle 2).  Tier A: the reference-made sum for the SESR-x2 crop
    (tests/golden/sesr_x2_rand.anchor.npz) under every requant form and accumulate mode the options reach; tier B: crafted nets for the other
    modes, expected = the oracle's frame + the nearest-upsampled input (one fp32 add)."""
    fx, meta, net, x = fixture_case(os.path.join(ROOT, "tests", "golden", "sesr_x2_rand.crop.npz"))
    an = np.load(os.path.join(ROOT, "tests", "golden", "sesr_x2_rand.anchor.npz"), allow_pickle=False)
    xt = torch.from_numpy(x).to(dev())
    b = bundle_from_oracle(net)
    for kw in (dict(), dict(reduced_forms=RF_ALL & ~16), dict(reduced_forms=RF_ALL & ~16 & ~32), dict(force_general=True), dict(fuse_hidden=0),
               dict(engine=_lib.ENGINE_DOT4)):
        with Track(pinned=True):
            e = sesrq.Engine(b, dev(), anchor_add=True, **kw)
            for wq, wf in ((False, True), (True, True)):
                q, y = e.forward(xt, want_q=wq, want_f=wf)
                eq(f"anchor {kw} q={wq}", y, an["sum_crop"])
                if wq:
                    eq(f"anchor {kw} int8 frame unaffected", q, shuffle(fx["input5"], 2))
    for name, risky in (("merged", None), ("hybrid", {4: ((1,), range(16))}), ("general", {4: ((0, 2), range(16))})):
        for form in (1, 2, None):
            netc = craft_net(500 + len(name), 3, 12, 2, risky=risky, forms={4: form} if form else None)
            for kw in (dict(), dict(reduced_forms=RF_ALL & ~16), dict(reduced_forms=RF_ALL & ~16 & ~32)):
                e = sesrq.Engine(bundle_from_oracle(netc), dev(), anchor_add=True, **kw)
                for (N, H, W) in ((1, 21, 70), (2, 9, 33)):
                    xc = rand_frame((N, 3, H, W), 31 * H + W)
                    want = O.forward(netc, xc)
                    ya = (want["y"] + np.repeat(np.repeat(xc, 2, axis=2), 2, axis=3)).astype(np.float32)
                    with Track(pinned=False):
                        _, y = e.forward(torch.from_numpy(xc).to(dev()), want_q=False, want_f=True)
                        eq(f"anchor craft {name} form {form} {kw} {N}x{H}x{W}", y, ya)
    netw = craft_net(520, 3, 12, 2, bits=(17, 19), wide_all=True)      # run-time accumulator bounds (GEN_ANY) under the anchor flavour
    e = sesrq.Engine(bundle_from_oracle(netw), dev(), anchor_add=True)
    xc = rand_frame((1, 3, 21, 70), 77)
    want = O.forward(netw, xc)
    with Track(pinned=False):
        _, y = e.forward(torch.from_numpy(xc).to(dev()), want_q=False, want_f=True)
        eq("anchor craft 17/19 bits", y, (want["y"] + np.repeat(np.repeat(xc, 2, axis=2), 2, axis=3)).astype(np.float32))
    # ... and through the frame table of a grouped launch: every image adds ITS OWN input frame
    e = sesrq.Engine(b, dev(), anchor_add=True)
    xs = [torch.from_numpy(x * np.float32(0.5 + 0.1 * k)).to(dev()) for k in range(4)]
    wants = [e.forward(t, want_q=False)[1].clone() for t in xs]
    of = [torch.zeros_like(wants[0]) for _ in range(4)]
    torch.cuda.synchronize()
    with Track(pinned=False):
        e.submission(xs, None, [torch.cuda.Stream(device=dev())], outs_f=of, group=4).enqueue(4)
        torch.cuda.synchronize()
        for k in range(4):
            eq(f"anchor grouped frame {k}", of[k], wants[k].cpu().numpy())


def test_trio_instances_vs_oracle():
    """Tier B: 8-conv nets (two fused trios: a plain one, EPI_MID, and the residual-merging one whose residual operand is NOT its input)
    under the five epilogue modes."""
    for zeros in (None, [-128, -128, -128, -128, -120, -128, -128, -128, -128]):
        net = craft_net(400, 3, 3, 1, ks=(5, 3, 3, 3, 3, 3, 3, 5), zeros=zeros, forms=None if zeros else {k: 1 for k in range(1, 7)})
        kws = [dict(reduced_forms=m) for m in RF_MASKS[:5]] + [dict(wg_budget=24)]
        check_vs_oracle(f"two trios zeros={'odd' if zeros else '-128'}", net, kws, sizes=((1, 37, 130), (2, 9, 33)), int8_too=False)


def test_calibration_and_proof_kernels_run():
    """The calibration pass's kernels and the load-time proof kernel are instantiations too (their parity tests live in test_host_mirror.py /
    test_gpu_parity.py); here they only have to be launched so that the coverage table is complete."""
    from sesrq.calibrate import Calibrator
    fx, meta, net, x = fixture_case(os.path.join(ROOT, "tests", "golden", "sesr_x4.crop.npz"))
    p = np.load(os.path.join(ROOT, "tests", "golden", "sesr_x4.params.npz"), allow_pickle=False)
    with Track(pinned=True):
        for method in ("minmax", "entropy"):
            c = Calibrator([p[f"Wf{k}"] for k in range(5)], [p[f"bf{k}"] for k in range(5)], 4, dev(), method=method)
            xt = torch.from_numpy(x).to(dev())
            c.observe(xt)
            if method == "entropy":
                c.begin_histogram_pass()
                c.observe(xt)
            c.finalize()
        torch.cuda.synchronize()
    # the load-time proof of the input quantiser's division form runs once per (scale_0, zero_0) of a process (cached): a domain no other
    # test of the session has used makes sesrq_create launch it here, and the forward that relies on the proof is checked against the oracle
    fresh = O.Net(**{**net.__dict__, "scale": [net.scale[0] * (1.0 + 2.0 ** -10 + 2.0 ** -17)] + list(net.scale[1:])})      # used nowhere else
    with Track(pinned=False):
        e = sesrq.Engine(bundle_from_oracle(fresh), dev())
        assert e.fast_division_proven()
        q, y = e.forward(torch.from_numpy(x).to(dev()))
        want = O.forward(fresh, x)
        eq("fresh input domain q", q, want["q_out"])
        eq("fresh input domain y", y, want["y"])


def test_zz_every_kernel_instance_ran():
    """LAST in this file: every instantiation the library can launch has been launched by a checked case above.  Writes the table."""
    inst = _lib.instances()
    lines = []
    for name in sorted(inst):
        by = "reference-made data" if name in REF_PINNED else ("oracle only" if name in ORACLE_ONLY else "NOT RUN")
        lines.append(f"{inst[name]:8d}  {by:20s} {name}")
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "instance_coverage.txt"), "w") as f:
            f.write(f"# {len(inst)} kernel instantiations; {len(REF_PINNED)} checked on reference-made data, "
                    f"{len(ORACLE_ONLY - REF_PINNED)} against the oracle only\n# launches  checked against       instantiation\n" + "\n".join(lines) + "\n")
    except OSError:
        pass
    missing = sorted(n for n in inst if n not in REF_PINNED and n not in ORACLE_ONLY)
    assert not missing, f"{len(missing)} of {len(inst)} kernel instantiations were never launched by a checked case:\n  " + "\n  ".join(missing)
