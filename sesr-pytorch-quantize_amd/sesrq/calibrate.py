"""Calibration pass on the GPU: the reference's exe_mode 0 (test.py:79-113,141-217).

The float net -- collapsed convs with fake-quantised weights, the float long skip, a fake-quantiser in
front of every conv (and of PixelShuffle) -- is run on calibration frames while the running min/max of
every quantiser input is observed; `finalize()` turns them into the activation domains (scale, zero) the
integer path consumes, with the reference's rule for the output domain (min := 0, so zero_L = -128).
The per-conv arithmetic lives in libsesrq (sesrq_calib_conv / _minmax / _fakequant); this module is the
host-side bookkeeping (python float arithmetic exactly as the reference's scripts do it).
Pinned to the reference within a tolerance (fp32 summation order differs), not bit for bit."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .bundle import Bundle, derive_bundle, quantize_weight


class Calibrator:
    def __init__(self, weights: Sequence[np.ndarray], biases: Sequence[np.ndarray], pixel_shuffle: int = 1,
                 device: Optional[torch.device] = None, pe_acc_bits: int = 18, pe_add_bits: int = 20, bias_bits: int = 16,
                 quantized=None):
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

    # ---- observers -----------------------------------------------------------------------
    def reset(self):
        """test.py:108-113: forget earlier ranges before a calibration run."""
        self.run_min = [None] * (self.L + 1)
        self.run_max = [None] * (self.L + 1)

    def _observe(self, k: int, t: torch.Tensor):
        st = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(_lib.lib().sesrq_calib_minmax(t.data_ptr(), t.numel(), self._mm.data_ptr(), self._scratch.data_ptr(), st))
        mn, mx = (float(v) for v in self._mm.cpu().numpy())
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
        scale, zero = [], []
        for k in range(self.L + 1):
            if self.run_min[k] is None:
                raise RuntimeError("Calibrator.finalize: no frames observed")
            s, z = calib_scale_zero(0.0 if k == self.L else self.run_min[k], self.run_max[k])
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
