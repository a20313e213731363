"""CPU ORACLE (numpy) -- TEST INFRASTRUCTURE ONLY, never on the product path.

An independent restatement of the reference's INT8 integer-inference arithmetic
(gui-yupeng/sesr-pytorch-quantize, `sim.py` path).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file.  The shipped package
(sesr-pytorch-quantize_amd/) must never import it and fails loudly when its HIP library
is missing.

Parity status: PINNED.  Every function below is checked bit-for-bit against fixtures under
tests/golden/ that were produced by importing and running the reference itself in the build
container (tests/golden/make_golden.py; tests/test_oracle_golden.py is the check).

Reference citations are relative to /root/reference (snapshot 2024-10-08).
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------- host scalars
def qconst(r: float, data_bit: int = 16, shift_max: int = 32):
    """real multiplier r -> (M < 2**data_bit, n <= shift_max) with r ~= M * 2**-n.

    Restates myQL/quan_func.py:495-515 (quan_layer_between_const): truncation, not rounding;
    for r >= 1 the integer part decides n, otherwise n = data_bit + number of leading
    binary zeros of the fraction, capped at shift_max.
    """
    assert data_bit < shift_max
    r = float(r)
    ip = int(r)
    if ip != 0:
        n = data_bit - math.ceil(math.log2(ip + 1))
    else:
        d = r * 2.0
        times = 0
        while int(d) == 0:
            times += 1
            d *= 2.0
        n = min(times + data_bit, shift_max)
    return int(r * (2.0 ** n)), n


def quantize_weight(w: np.ndarray, width: int = 8):
    """float conv weight -> (int8 weight, python-float scale).

    Restates myQL/quan_func.py:58-71: per-TENSOR symmetric, scale = 2*absmax/(2**width-1),
    Wq = clamp(rint(W / f32(scale))) (fp32 tensor divided by a python scalar).
    """
    w = np.asarray(w, dtype=F32)
    absmax = max(abs(float(w.max())), abs(float(w.min())))
    assert absmax > 0
    qmax, qmin = 2 ** (width - 1) - 1, -(2 ** (width - 1))
    scale = (absmax - (0 - absmax)) / (qmax - qmin)
    q = np.clip(np.rint(w / F32(scale)), qmin, qmax)
    return q.astype(np.int8), scale


def calib_scale_zero(min_val: float, max_val: float, width: int = 8):
    """running (min,max) -> (scale, zero).  Restates test.py:185-217 / quan_func.py:198-205."""
    qmax, qmin = 2 ** (width - 1) - 1, -(2 ** (width - 1))
    scale = (max_val - min_val) / (qmax - qmin)
    zero = qmin - round(min_val / scale)
    return scale, int(zero)


def add_const(bias_f: np.ndarray, wq: np.ndarray, s_in: float, z_in: int, s_w: float, bias_width: int = 16):
    """16-bit bias constant  clamp16( clamp16(rint(b/(s_in*s_w))) - z_in * sum(Wq) ).

    Restates myQL/quan_func.py:448-449,402 (bias quantiser) and :481-486 (constant with the
    UNCLAMPED zero point).
    """
    lo, hi = -(2 ** (bias_width - 1)), 2 ** (bias_width - 1) - 1
    bias_scale = s_in * s_w
    bq = np.clip(np.rint(np.asarray(bias_f, F32) / F32(bias_scale)), lo, hi)
    sw = wq.astype(np.int64).reshape(wq.shape[0], -1).sum(1)
    app = (sw.astype(F32) * F32(z_in))
    return np.clip(bq - app, lo, hi).astype(np.int32)


# --------------------------------------------------------------------------- net description
@dataclass
class Layer:
    wq: np.ndarray            # (OC, IC, k, k) int8
    add_const: np.ndarray     # (OC,) int32, already 16-bit saturated
    M: int
    n: int
    relu: bool
    # Per-OUTPUT-CHANNEL requant constants, or None = the reference's one (M, n) per layer.  NOT in the reference (its weight quantiser is per
    # tensor, quan_func.py:58-71); BASELINE's north star names per-channel weight scales.  PARITY UNPINNED: this is the definition the HIP
    # path is checked against, nothing of the reference pins it.  t[oc] = f32(f32(acc[oc]) * f32(M_oc[oc])) * 2^-n_oc[oc].
    M_oc: Optional[np.ndarray] = None
    n_oc: Optional[np.ndarray] = None


@dataclass
class Net:
    layers: List[Layer]
    scale: List[float]        # s_0 .. s_L   (python floats), s_L = output domain
    zero: List[int]           # z_0 .. z_L
    M_res: int
    n_res: int
    pixel_shuffle: int = 1    # 1 = none
    pe: int = 4
    acc_bits: int = 18
    add_bits: int = 20
    name: str = ""

    @property
    def L(self):
        return len(self.layers)


def quantize_weight_per_channel(w: np.ndarray, width: int = 8):
    """One symmetric scale per OUTPUT channel (no reference counterpart; parity unpinned): quantize_weight applied to every w[oc]."""
    w = np.asarray(w, dtype=F32)
    qs = [quantize_weight(w[o], width) for o in range(w.shape[0])]
    return np.stack([q for q, _ in qs]), np.array([s for _, s in qs], dtype=np.float64)


def derive_net(Wf: List[np.ndarray], bf: List[np.ndarray], scale: List[float], zero: List[int],
               pixel_shuffle: int, name: str = "", per_channel: bool = False) -> Net:
    """float collapsed convs + calibrated (scale, zero) -> integer parameter bundle.

    Role rules restate myQL/quan_func.py:523-609: layer 0 and layer L-2 requantise into
    domain 1 (the long-residual domain), layer L-1 into domain L, the others into k+1;
    residual multiplier = s_1 / s_{L-1} (quan_func.py:256-260).  For L == 5 this is exactly
    the reference; for other depths (nrdm_6) it is the positional generalisation (parity
    unpinned, no reference implementation exists).
    """
    L = len(Wf)
    layers = []
    for k in range(L):
        nxt = 1 if k in (0, L - 2) else k + 1
        if per_channel:      # unpinned variant: the layer's requant multiplier and bias constant per output channel
            wq, sws = quantize_weight_per_channel(Wf[k])
            Mn = [qconst(scale[k] / scale[nxt] * float(s_)) for s_ in sws]
            ac = np.concatenate([add_const(np.asarray(bf[k])[o:o + 1], wq[o:o + 1], scale[k], zero[k], float(sws[o])) for o in range(wq.shape[0])])
            layers.append(Layer(wq=wq, add_const=ac, M=Mn[0][0], n=Mn[0][1], relu=(k != L - 1),
                                M_oc=np.array([m for m, _ in Mn], np.int64), n_oc=np.array([n_ for _, n_ in Mn], np.int64)))
            continue
        wq, sw = quantize_weight(Wf[k])
        M, n = qconst(scale[k] / scale[nxt] * sw)
        layers.append(Layer(wq=wq, add_const=add_const(bf[k], wq, scale[k], zero[k], sw), M=M, n=n, relu=(k != L - 1)))
    M_res, n_res = qconst(scale[1] / scale[L - 1])
    return Net(layers=layers, scale=list(scale), zero=list(zero), M_res=M_res, n_res=n_res,
               pixel_shuffle=pixel_shuffle, name=name)


def net_from_fixture(fx) -> Net:
    """Build a Net straight from a tests/golden/*.npz fixture (integer bundle as harvested)."""
    meta = json.loads(str(fx["meta"]))
    L = 5
    layers = [Layer(wq=fx[f"Wq{k}"].astype(np.int8), add_const=fx[f"add_const{k}"].astype(np.int32),
                    M=meta["M"][k], n=meta["n"][k], relu=(k != L - 1)) for k in range(L)]
    ps = {5: 4, 6: 2, 3: 1}[meta["mflag"]]
    return Net(layers=layers, scale=meta["scale"], zero=meta["zero"], M_res=meta["M_res"], n_res=meta["n_res"],
               pixel_shuffle=ps, name=meta["case"])


# --------------------------------------------------------------------------- forward
def _sat(x, bits):
    return np.clip(x, -(2 ** (bits - 1)), 2 ** (bits - 1) - 1)


def quantize_input(x: np.ndarray, s0: float, z0: int, reciprocal: bool = False) -> np.ndarray:
    """q0 = clamp8(rint(x / f32(s0) + f32(z0)))   -- myQL/quan_func.py:222-225 (true fp32 division).
    reciprocal=True: x * f32(1 / f32(s0)) instead of the quotient -- how torch evaluates tensor / scalar on a GPU, where
    the reference's scripts run; no fixture pins it (the goldens come from a CPU run)."""
    x = np.asarray(x, F32)
    with np.errstate(over="ignore"):          # huge x / s0 -> inf -> clamps to +-127 like in the reference
        t = x * (F32(1) / F32(s0)) if reciprocal else x / F32(s0)
    return np.clip(np.rint(t + F32(z0)), -128, 127).astype(np.int8)


def conv_pe(q: np.ndarray, lay: Layer, z_in: int, pe: int, acc_bits: int, add_bits: int, return_raw: bool = False):
    """One layer's integer accumulate.  q: (N,IC,H,W) int8 -> (pe_out (N,pe,OC,H,W), acc (N,OC,H,W)) int64.

    pe_out[p] = clamp_acc( sum_{ic = p mod pe, taps} W*q )   with the image padded by zc = max(z_in,-128)
    acc       = clamp_add( sum_p pe_out[p] ) + add_const
    Restates myQL/quan_func.py:298-318 (channel split), the zero-padded nn.Conv2d on (q - zc),
    :338-356 (add back zc*sum(W_pe)), :370 (18-bit clamp after the complete PE sum), :380-386,
    :437 (20-bit clamp), :491 (constant added after the clamp, no further clamp).
    """
    N, IC, H, W = q.shape
    OC, _, k, _ = lay.wq.shape
    r = k // 2
    zc = max(int(z_in), -128)
    qp = np.full((N, IC, H + 2 * r, W + 2 * r), zc, dtype=np.float64)
    qp[:, :, r:r + H, r:r + W] = q
    w = lay.wq.astype(np.float64)
    pe_out = np.zeros((N, pe, OC, H, W), dtype=np.float64)
    for p in range(pe):
        ch = list(range(p, IC, pe))
        if not ch:
            continue
        for ky in range(k):
            for kx in range(k):
                patch = qp[:, ch, ky:ky + H, kx:kx + W]                # (N, c, H, W)
                pe_out[:, p] += np.einsum("oc,nchw->nohw", w[:, ch, ky, kx], patch, optimize=True)
    pe_raw = pe_out.astype(np.int64)          # before saturation: what the reference tests for its overflow prints (:358-361)
    pe_out = _sat(pe_raw, acc_bits)
    acc = _sat(pe_out.sum(1), add_bits) + lay.add_const.astype(np.int64)[None, :, None, None]
    if return_raw:
        return pe_out, acc, pe_raw
    return pe_out, acc


def requant(acc: np.ndarray, M, n) -> np.ndarray:
    """t = f32(f32(acc) * f32(M)) * 2**-n   -- myQL/quan_func.py:529,560,584,605.

    The fp32 rounding of the product is load-bearing (|acc*M| reaches 2**31..2**32).  M, n: the layer's scalars, or [OC] arrays for the
    unpinned per-output-channel variant (Layer.M_oc): the same two fp32 operations with the channel's own constants."""
    if np.ndim(M):
        Mv = np.asarray(M, dtype=np.float64).astype(F32)[None, :, None, None]
        sh = np.exp2(-np.asarray(n, dtype=np.float64)).astype(F32)[None, :, None, None]
        return (acc.astype(F32) * Mv) * sh
    return (acc.astype(F32) * F32(M)) * F32(2.0 ** (-n))


def _q8(v):
    return np.clip(np.rint(v), -128, 127)


def pixel_shuffle(a: np.ndarray, r: int) -> np.ndarray:
    """(N, C*r*r, H, W) -> (N, C, H*r, W*r); out[c, h*r+i, w*r+j] = in[c*r*r + i*r + j, h, w]."""
    if r == 1:
        return a
    N, C, H, W = a.shape
    c = C // (r * r)
    return a.reshape(N, c, r, r, H, W).transpose(0, 1, 4, 2, 5, 3).reshape(N, c, H * r, W * r)


def forward(net: Net, x: np.ndarray, keep: bool = False) -> Dict[str, np.ndarray]:
    """Full integer forward.  x: (N, Cin, H, W) fp32.  Returns q_out (int8, pixel-shuffled),
    y (fp32 dequantised, pixel-shuffled) and, with keep=True, every stage the reference dumps
    (input{k}, pe_out{k}, pe_add{k}, shortcut, input4_special) for N == 1 comparisons.
    """
    L = net.L
    st: Dict[str, np.ndarray] = {}
    q = quantize_input(x, net.scale[0], net.zero[0])
    short = None
    for k, lay in enumerate(net.layers):
        if keep:
            st[f"input{k}"] = q
        pe_out, acc, pe_raw = conv_pe(q, lay, net.zero[k], net.pe, net.acc_bits, net.add_bits, return_raw=True)
        if keep:
            st[f"pe_raw{k}"] = pe_raw                  # all frames, int64, un-saturated
            st[f"pe_out{k}"] = pe_out[0].astype(np.int32)
            st[f"pe_add{k}"] = (acc - lay.add_const.astype(np.int64)[None, :, None, None]).astype(np.int32)
        t = requant(acc, lay.M, lay.n) if lay.M_oc is None else requant(acc, lay.M_oc, lay.n_oc)
        if lay.relu:
            t = np.maximum(t, F32(0))
        if k == L - 1:
            # quan_func.py:584-594: requantise into the output domain, then dequantise for software tests
            qo = _q8(t + F32(net.zero[L])).astype(np.int8)
            y = (qo.astype(F32) - F32(net.zero[L])) * F32(net.scale[L])
            if keep:
                st[f"input{L}"] = qo
            st["q_out"] = pixel_shuffle(qo, net.pixel_shuffle)
            st["y"] = pixel_shuffle(y.astype(F32), net.pixel_shuffle)
            break
        if k == 0:
            short = t                                   # quan_func.py:530,549 (ReLU'd, un-rounded)
            if keep:
                st["shortcut"] = short.astype(F32)
        if k == L - 2:
            # long residual merged in the integer domain at the input of the last conv
            # quan_func.py:249-270: both operands re-quantised with offset -128, +256, requant
            rc = _q8(short - F32(128))
            ic = _q8(t - F32(128))
            if keep:
                st["input4_special"] = ic.astype(np.int8)
            u = rc + ic + F32(256)
            v = (u * F32(net.M_res)) * F32(2.0 ** (-net.n_res))
            q = _q8(v + F32(net.zero[k + 1])).astype(np.int8)
        else:
            q = _q8(t + F32(net.zero[k + 1])).astype(np.int8)      # quan_func.py:275-280
    return st


# --------------------------------------------------------------------------- synthetic nets
def synth_net(kind: str, seed: int = 0, n_blocks: int = 3, hard: bool = False) -> Net:
    """Deterministic random integer bundle of a reference topology (no checkpoint needed).

    kind: 'sesr_x4' (1->16, PS4), 'sesr_x2' (3->12, PS2), 'nrdm' (3->3).  `hard` draws wide
    weights / odd zero points so that the saturation and z<-128 branches fire.
    """
    rng = np.random.default_rng(seed)
    cin, cout, ps = {"sesr_x4": (1, 16, 4), "sesr_x2": (3, 12, 2), "nrdm": (3, 3, 1)}[kind]
    shapes = [(16, cin, 5)] + [(16, 16, 3)] * n_blocks + [(cout, 16, 5)]
    L = len(shapes)
    layers = []
    for k, (oc, ic, ks) in enumerate(shapes):
        if hard:
            w = rng.choice(np.array([-128, -100, 90, 127], dtype=np.int64), size=(oc, ic, ks, ks))
            w = np.where(rng.random(w.shape) < 0.15, rng.integers(-128, 128, w.shape), w)
        else:
            w = np.clip(np.rint(rng.standard_normal((oc, ic, ks, ks)) * 14.0), -128, 127)
            w[rng.integers(oc), rng.integers(ic), ks // 2, ks // 2] = 127
        fan = ic * ks * ks
        tgt = 60.0 / (np.sqrt(fan) * (110.0 if hard else 14.0) * 74.0)     # keeps activations spread over int8
        M, n = qconst(float(tgt * rng.uniform(0.6, 1.6)))
        ac = np.clip(rng.integers(-40000, 40000, oc), -32768, 32767).astype(np.int32) if hard else \
            rng.integers(-6000, 6000, oc).astype(np.int32)
        layers.append(Layer(wq=w.astype(np.int8), add_const=ac, M=M, n=n, relu=(k != L - 1)))
    if hard:
        zero = [int(z) for z in rng.integers(-150, -100, L + 1)]
    else:
        zero = [-128] * (L + 1)
    scale = [float(s) for s in rng.uniform(0.003, 0.04, L + 1)]
    scale[0] = 1.0 / 255.0
    M_res, n_res = qconst(float(rng.uniform(0.2, 0.9)))
    return Net(layers=layers, scale=scale, zero=zero, M_res=M_res, n_res=n_res, pixel_shuffle=ps,
               name=f"synth_{kind}_{seed}{'_hard' if hard else ''}")
