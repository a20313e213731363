"""Calibration pass on the GPU: the reference's exe_mode 0 (test.py:79-113,141-217).

The float net -- collapsed convs with fake-quantised weights, the float long skip, a fake-quantiser in
front of every conv (and of PixelShuffle) -- is run on calibration frames while the running min/max of
every quantiser input is observed; `finalize()` turns them into the activation domains (scale, zero) the
integer path consumes, with the reference's rule for the output domain (min := 0, so zero_L = -128).
The per-conv arithmetic lives in libsesrq (sesrq_calib_conv / _minmax / _fakequant); this module is the
host-side bookkeeping (python float arithmetic exactly as the reference's scripts do it).
Pinned to the reference within a tolerance (fp32 summation order differs), not bit for bit.

Entropy variant (`method="entropy"`; BASELINE's north star names KL-entropy activation ranges, the reference itself only
has min/max -- no oracle, PARITY UNPINNED): after the min/max pass a second pass over the same frames accumulates a
histogram of every quantiser input over its observed range (sesrq_calib_histogram), and `finalize()` replaces (min, max)
by the clipping range that minimises the KL divergence between the observed distribution and its 256-level quantisation
(`entropy_range`).  The output domain keeps the reference's rule (min := 0)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .bundle import Bundle, derive_bundle, quantize_weight


def _smooth(d: np.ndarray, eps: float = 1e-4) -> np.ndarray:
    """Move a little mass onto the empty bins of a normalised histogram so that KL(p || q) stays finite where the folded
    outliers of p meet an empty bin of q (the usual smoothing of entropy calibrators)."""
    zero = d == 0
    nz = int(zero.sum())
    if nz == 0 or nz == d.size:
        return d
    take = eps * nz / (d.size - nz)
    out = np.where(zero, eps, d - take)
    return np.maximum(out, eps * 1e-3)          # a non-empty bin lighter than its share of the smoothing stays positive


def _kl_upper_cut(hist: np.ndarray, levels: int, stride: int) -> int:
    """Number of leading bins to keep (levels..len(hist)): the cut i minimising KL(P_i || Q_i), P_i = hist[:i] with the
    mass beyond folded into its last bin, Q_i = hist[:i] merged into `levels` equal buckets and spread back evenly over the
    non-empty bins of each bucket (the usual entropy-calibration construction)."""
    B = len(hist)
    h = hist.astype(np.float64)
    total = h.sum()
    if total <= 0 or B <= levels:
        return B
    tail = np.concatenate([np.cumsum(h[::-1])[::-1][1:], [0.0]])          # tail[i-1] = mass of bins i..B-1
    best_i, best_kl = B, np.inf
    for i in list(range(levels, B, stride)) + [B]:
        p = h[:i].copy()
        p[i - 1] += tail[i - 1]
        bucket = (np.arange(i) * levels) // i
        mass = np.bincount(bucket, weights=h[:i], minlength=levels)
        nonzero = np.bincount(bucket, weights=(h[:i] > 0).astype(np.float64), minlength=levels)
        q = np.where(h[:i] > 0, mass[bucket] / np.maximum(nonzero[bucket], 1.0), 0.0)
        qs = q.sum()
        if qs <= 0:
            continue
        p, q = _smooth(p / p.sum()), _smooth(q / qs)
        kl = float(np.sum(p * np.log(p / q)))
        if kl < best_kl - 1e-12:
            best_kl, best_i = kl, i
    return best_i


def entropy_range(hist: np.ndarray, lo: float, hi: float, levels: int = 256, stride: int = 8):
    """Clipping range (min, max) inside [lo, hi] for a histogram of equal bins over [lo, hi): the upper cut by KL search,
    then -- for domains that reach below zero -- the lower cut by the same search on the mirrored histogram of what is
    left.  A domain that starts at or above zero (everything behind a ReLU, image inputs) keeps its lower end."""
    hist = np.asarray(hist)
    B = len(hist)
    if B < 2 or not hi > lo:
        raise ValueError("entropy_range: need at least two bins and lo < hi")
    w = (hi - lo) / B
    up = _kl_upper_cut(hist, levels, stride)
    kept = hist[:up].astype(np.float64).copy()
    kept[up - 1] += float(hist[up:].sum())
    down = 0
    if lo < 0 and up > levels:
        down = up - _kl_upper_cut(kept[::-1], levels, stride)
    return lo + down * w, lo + up * w


class Calibrator:
    def __init__(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], pixel_shuffle: int = 1,
                 device: Optional[torch.device] = None, pe_acc_bits: int = 18, pe_add_bits: int = 20, bias_bits: int = 16,
                 quantized=None, method: str = "minmax", bins: int = 2048):
        """weights: float collapsed convs (quantised here), or None with `quantized` = [(Wq int8, weight scale)]
        when quantize_model_weight already did it."""
        if not torch.cuda.is_available():
            raise RuntimeError("sesrq.Calibrator needs a HIP device (no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.L = len(biases)
        self.pixel_shuffle = int(pixel_shuffle)
        self.acc_bits, self.add_bits, self.bias_bits = pe_acc_bits, pe_add_bits, bias_bits
        self.weights_f = [np.asarray(w, np.float32) for w in weights] if weights is not None else None
        self.biases_f = [np.asarray(b, np.float32) for b in biases]
        self.wq, self.sw, self._wdev = [], [], []
        if quantized is None:
            quantized = [quantize_weight(w) for w in self.weights_f]
        for q, s in quantized:
            q = np.ascontiguousarray(q, dtype=np.int8)
            self.wq.append(q)
            self.sw.append(float(s))
            self._wdev.append(torch.from_numpy(q.astype(np.int32)).to(self.device).contiguous())
        self.run_min: List[Optional[float]] = [None] * (self.L + 1)
        self.run_max: List[Optional[float]] = [None] * (self.L + 1)
        self.last_scale: List[Optional[float]] = [None] * (self.L + 1)   # per-batch values, like input.K.scale.pt during test.py
        self.last_zero: List[Optional[int]] = [None] * (self.L + 1)
        self._mm = torch.empty(2, dtype=torch.float32, device=self.device)
        self._scratch = torch.empty(2, dtype=torch.int32, device=self.device)
        self.hist: List[Optional[torch.Tensor]] = [None] * (self.L + 1)     # entropy variant: second-pass histograms
        self._hist_pass = False
        self.set_method(method, bins)

    def set_method(self, method: str, bins: int = 2048):
        """'minmax' = the reference's rule; 'entropy' = KL-minimising clipping ranges from a second (histogram) pass."""
        if method not in ("minmax", "entropy"):
            raise ValueError("Calibrator: method must be 'minmax' (the reference's) or 'entropy'")
        if not 256 < bins <= 4096:
            raise ValueError("Calibrator: bins must be in 257..4096")
        if self._hist_pass:
            raise RuntimeError("Calibrator.set_method: a histogram pass is in progress (reset() first)")
        self.method, self.bins = method, int(bins)

    # ---- observers -----------------------------------------------------------------------
    def reset(self):
        """test.py:108-113: forget earlier ranges before a calibration run."""
        self.run_min = [None] * (self.L + 1)
        self.run_max = [None] * (self.L + 1)
        self.hist = [None] * (self.L + 1)
        self._hist_pass = False

    def begin_histogram_pass(self):
        """Entropy variant: call after the min/max pass; the following observe() calls (the same frames again) accumulate
        the histograms over the ranges found so far instead of widening them."""
        if self.method != "entropy":
            raise RuntimeError("begin_histogram_pass: the calibrator was built with method='minmax'")
        if any(v is None for v in self.run_min):
            raise RuntimeError("begin_histogram_pass: run the min/max pass (observe) first")
        self.hist = [torch.zeros(self.bins, dtype=torch.int32, device=self.device) for _ in range(self.L + 1)]
        self._hist_pass = True

    def _observe(self, k: int, t: torch.Tensor):
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().sesrq_calib_minmax(t.data_ptr(), t.numel(), self._mm.data_ptr(), self._scratch.data_ptr(), st))
        mn, mx = (float(v) for v in self._mm.cpu().numpy())
        if self._hist_pass:
            _lib.check(_lib.lib().sesrq_calib_histogram(t.data_ptr(), t.numel(), float(np.float32(self.run_min[k])),
                                                        float(np.float32(self.run_max[k])), self.bins, self.hist[k].data_ptr(), st))
        else:
            if self.run_max[k] is None or self.run_max[k] < mx:
                self.run_max[k] = mx
            if self.run_min[k] is None or self.run_min[k] > mn:
                self.run_min[k] = mn
        assert mx != mn, "Input tensor is all equal,{}".format(k)
        scale = (mx - mn) / 255
        zero = -128 - round(mn / scale)
        self.last_scale[k], self.last_zero[k] = scale, int(zero)
        return scale, int(zero)

    # ---- one calibration forward -----------------------------------------------------------
    def observe(self, x: torch.Tensor) -> torch.Tensor:
        """x: (N, Cin, H, W) fp32 on the device.  Returns what the reference's mode-0 model returns (the
        fake-quantised float output, pixel-shuffled)."""
        if x.dim() != 4 or x.dtype != torch.float32 or x.device != self.device:
            raise ValueError("Calibrator.observe: need a (N,C,H,W) float32 tensor on " + str(self.device))
        lib = _lib.lib()
        st = torch.cuda.current_stream(self.device).cuda_stream
        N, _, H, W = x.shape
        a = x.contiguous()
        first = None
        L = self.L
        for k in range(L):
            scale, zero = self._observe(k, a)
            sw = self.sw[k]
            oc, ic, ks, _ = self.wq[k].shape
            bias_scale = scale * sw
            lo16, hi16 = -(2 ** (self.bias_bits - 1)), 2 ** (self.bias_bits - 1) - 1
            bq = np.clip(np.rint(self.biases_f[k] / np.float32(bias_scale)), lo16, hi16).astype(np.float32)
            qb = torch.from_numpy((bq * np.float32(bias_scale)).astype(np.float32)).to(self.device)
            hi_a, lo_a = 2 ** (self.acc_bits - 1) - 1, -(2 ** (self.acc_bits - 1))
            hi_s, lo_s = 2 ** (self.add_bits - 1) - 1, -(2 ** (self.add_bits - 1))
            desc = _lib.CalibConvDesc(k=ks, ic=ic, oc=oc, w=self._wdev[k].data_ptr(), qbias=qb.data_ptr(),
                                      in_scale=float(np.float32(scale)), in_zero=zero, ss=float(np.float32(scale * sw)),
                                      acc_lo=float(np.float32((lo_a - zero) * scale * sw)), acc_hi=float(np.float32((hi_a - zero) * scale * sw)),
                                      add_lo=float(np.float32((lo_s - zero) * scale * sw)), add_hi=float(np.float32((hi_s - zero) * scale * sw)),
                                      relu=int(k != L - 1))
            out = torch.empty((N, oc, H, W), dtype=torch.float32, device=self.device)
            skip = first if k == L - 2 else None          # long skip: x_{L-1} = a_{L-2} + a_0 (float AddOp)
            _lib.check(lib.sesrq_calib_conv(C.byref(desc), a.data_ptr(), skip.data_ptr() if skip is not None else None,
                                            out.data_ptr(), N, H, W, st))
            if k == 0:
                first = out
            a = out
        if self.pixel_shuffle > 1:
            scale, zero = self._observe(L, a)                 # quantiser in front of PixelShuffle (test.py:90-91)
            fq = torch.empty_like(a)
            _lib.check(lib.sesrq_calib_fakequant(a.data_ptr(), fq.data_ptr(), a.numel(), float(np.float32(scale)), zero, st))
            return torch.nn.functional.pixel_shuffle(fq, self.pixel_shuffle)
        self._observe(L, a)                                   # nets without PixelShuffle: range of the last conv's output (quan_func.py:460-479)
        return a

    # ---- results ------------------------------------------------------------------------------
    def finalize(self):
        """running (min, max) -> (scale[0..L], zero[0..L]) as test.py:185-217 (output domain: min := 0)."""
        from .bundle import calib_scale_zero
        if self.method == "entropy" and not self._hist_pass:
            raise RuntimeError("Calibrator.finalize: method='entropy' needs the histogram pass (begin_histogram_pass, observe again)")
        scale, zero = [], []
        self.ranges = []
        for k in range(self.L + 1):
            if self.run_min[k] is None:
                raise RuntimeError("Calibrator.finalize: no frames observed")
            mn, mx = self.run_min[k], self.run_max[k]
            if self.method == "entropy":
                lo32, hi32 = float(np.float32(mn)), float(np.float32(mx))       # the edges the histogram kernel used
                mn, mx = entropy_range(self.hist[k].cpu().numpy().astype(np.int64), lo32, hi32)
            self.ranges.append((mn, mx))
            s, z = calib_scale_zero(0.0 if k == self.L else mn, mx)
            scale.append(s)
            zero.append(z)
        return scale, zero

    def bundle(self, name: str = "") -> Bundle:
        from .bundle import derive_bundle_from_quantized
        scale, zero = self.finalize()
        return derive_bundle_from_quantized(self.wq, self.sw, self.biases_f, scale, zero, self.pixel_shuffle, name=name,
                                            pe_acc_bits=self.acc_bits, pe_add_bits=self.add_bits, bias_bit=self.bias_bits)


def finish_calibration(store, L: int):
    """The tail of the reference's test.py (:185-217): running input.K.{min,max}_val -> final
    input.K.{scale,zero} for K = 0..L, output domain with min := 0."""
    from .bundle import calib_scale_zero
    scale, zero = [], []
    for k in range(L + 1):
        mn = 0.0 if k == L else float(store[f"input/input.{k}.min_val"])
        s, z = calib_scale_zero(mn, float(store[f"input/input.{k}.max_val"]))
        scale.append(s)
        zero.append(z)
    store.set_activation_domains(scale, zero)
    return scale, zero
