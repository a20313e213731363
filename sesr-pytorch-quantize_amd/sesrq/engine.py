"""Device engine: a Bundle uploaded to one GPU (sesrq_create) + the fused integer forward
(sesrq_forward).  torch is plumbing only: device memory, the current HIP stream."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .bundle import Bundle


class GraphedForward:
    """A forward (or a chain of forwards: nrdm_6 -> SESR-x2) captured as a HIP graph (Engine.capture)."""

    def __init__(self, engine, x, want_q, want_f, slot, downstream):
        if downstream and not want_q:
            raise ValueError("a chain hands the int8 output on: want_q must be True")
        dt = engine._check_in(x)
        del dt
        if not x.is_contiguous():
            raise ValueError("capture needs a contiguous input buffer (it is baked into the graph)")
        self.engines = (engine,) + downstream
        self.x = x
        dev = engine.device
        N, _, H, W = x.shape
        self.outs = []
        cur_shape = (N, H, W)
        for j, e in enumerate(self.engines):
            shp = e.out_shape(*cur_shape)
            last = j == len(self.engines) - 1
            q = torch.empty(shp, dtype=torch.int8, device=dev) if (want_q or not last) else None
            y = torch.empty(shp, dtype=torch.float32, device=dev) if (want_f and last) else None
            self.outs.append((q, y))
            e.workspace(*cur_shape, slot)                 # allocate outside the capture
            cur_shape = (shp[0], shp[2], shp[3])
        self.q, self.y = self.outs[-1]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._issue(side, slot)                       # warm-up: one-time occupancy queries / attribute calls happen here
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self._issue(side, slot)

    def _issue(self, stream, slot):
        cur = self.x
        for e, (q, y) in zip(self.engines, self.outs):
            e.forward(cur, want_q=q is not None, want_f=y is not None, out_q=q, out_f=y, stream=stream, slot=slot,
                      assume_ordered=True)
            cur = q

    def replay(self):
        self.graph.replay()
        return self.q, self.y


class Submission:
    """Frames, buffers and streams of Engine.submission(), laid out once as the C arrays sesrq_forward_many takes."""

    def __init__(self, engine, frames, outs_q, outs_f, streams, group=1):
        if not frames or not streams or (outs_q is None and outs_f is None):
            raise ValueError("submission: frames / outputs / streams do not match")
        for o in (outs_q, outs_f):
            if o is not None and len(o) != len(frames):
                raise ValueError("submission: frames / outputs / streams do not match")
        if len(frames) % len(streams):
            raise ValueError("submission: the frame list must be a multiple of the stream count (frame k runs on stream k % S, cyclically)")
        self.engine, self.n = engine, len(frames)
        self.dt = engine._check_in(frames[0])
        N, _, H, W = frames[0].shape
        self.shape = (N, H, W)
        shp = engine.out_shape(N, H, W)
        for k, x in enumerate(frames):
            if engine._check_in(x) != self.dt or tuple(x.shape) != tuple(frames[0].shape) or not x.is_contiguous():
                raise ValueError("submission: every frame must be a contiguous tensor of the same shape and dtype")
            for o, dtp, what in ((outs_q, torch.int8, "int8"), (outs_f, torch.float32, "fp32")):
                if o is not None and (tuple(o[k].shape) != tuple(shp) or o[k].dtype != dtp or not o[k].is_contiguous()):
                    raise ValueError(f"submission: every {what} output must be a contiguous tensor of the forward's output shape")
        self._keep = (list(frames), list(outs_q) if outs_q is not None else None, list(outs_f) if outs_f is not None else None, list(streams))
        # two periods back to back: any window of up to n frames starting anywhere in the cycle is one contiguous slice
        self.io = (_lib.FrameIO * (2 * self.n))()
        for k in range(2 * self.n):
            j = k % self.n
            self.io[k] = _lib.FrameIO(frames[j].data_ptr(), outs_q[j].data_ptr() if outs_q is not None else None,
                                      outs_f[j].data_ptr() if outs_f is not None else None)
        S = len(streams)
        if group < 1 or (group > 1 and N != 1):
            raise ValueError("submission: group >= 1, and grouping needs single-image frames")
        self.group = int(group)
        if self.group > 1:      # the frames of one launch sequence are written concurrently (the library refuses a shared buffer too)
            for o in (outs_q, outs_f):
                if o is None:
                    continue
                for s_ in range(S):
                    seq = [t.data_ptr() for t in o[s_::S]]
                    g_eff = min(self.group, len(seq))      # one call hands over at most one period: a group never wraps onto its own frames
                    ring = seq + seq[:g_eff - 1]
                    for i in range(len(seq)):
                        win = ring[i:i + g_eff]
                        if len(set(win)) != len(win):
                            raise ValueError("submission: frames that may share a launch sequence (any `group` consecutive frames of a "
                                             "stream) share an output buffer")
        self.ws = [engine.workspace(N * self.group, H, W, s) for s in range(S)]       # room for `group` frames per launch sequence
        self.ws_bytes = self.ws[0].numel()
        # the library maps frame k of a call to streams[k % S]: a window that starts at frame f of the list gets the arrays rotated by f % S
        self.ws_rot = [(C.c_void_p * S)(*[self.ws[(r + i) % S].data_ptr() for i in range(S)]) for r in range(S)]
        self.st_rot = [(C.c_void_p * S)(*[streams[(r + i) % S].cuda_stream for i in range(S)]) for r in range(S)]
        self.S = S

    def enqueue(self, count: int, first: int = 0):
        """Frames first .. first+count-1 (indices taken modulo the list): sesrq_forward_many calls of at most one period each."""
        lib, N, (n, H, W) = _lib.lib(), self.n, self.shape
        k, base, sz = first, C.addressof(self.io), C.sizeof(_lib.FrameIO)
        while count > 0:
            c = min(count, N)
            off = k % N
            _lib.check(lib.sesrq_forward_many(self.engine._h, C.cast(base + off * sz, C.POINTER(_lib.FrameIO)), c, self.dt, n, H, W,
                                              self.ws_rot[off % self.S], self.ws_bytes, self.st_rot[off % self.S], self.S, self.group))
            k += c
            count -= c


class Engine:
    """One immutable net on one device.  forward() is stream-ordered and allocation-free once
    the workspace for a given (N, H, W) exists."""

    def __init__(self, bundle: Bundle, device: Optional[torch.device] = None, engine: int = _lib.ENGINE_AUTO,
                 force_general: bool = False, exact_division: bool = False, anchor_add: bool = False,
                 fuse_hidden=1, wg_budget: int = 0, upstream: Optional[Bundle] = None, reciprocal_division: bool = False,
                 reduced_forms: int = -1):
        """reciprocal_division: form x / s0 of the input quantiser as x * fl(1/s0) -- torch's tensor / scalar on a GPU --
        instead of the true quotient the CPU-run reference and the goldens define (sesrq_options.exact_div = 2).
        upstream: the net whose int8 OUTPUT frames this engine takes as int8 input (chained nets, e.g. nrdm_6 ->
        SESR-x2): they are re-quantised into this net's input domain while the first layer stages them.
        reduced_forms: sesrq_options.reduced_forms -- which of the proven reduced epilogue forms the kernels may select (-1 = all); same
        bits either way, it only picks the kernel instantiation (tests run every one of them on reference-made data)."""
        if not torch.cuda.is_available():
            raise RuntimeError("sesrq.Engine needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback in this package")
        self.bundle = bundle
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        L = bundle.L
        self._keep = []
        layers = (_lib.LayerDesc * L)()
        for k, l in enumerate(bundle.layers):
            w = np.ascontiguousarray(l.wq, dtype=np.int8)
            ac = np.ascontiguousarray(l.add_const, dtype=np.int32)
            self._keep += [w, ac]
            layers[k] = _lib.LayerDesc(k=w.shape[2], ic=w.shape[1], oc=w.shape[0],
                                       w=w.ctypes.data_as(C.POINTER(C.c_int8)),
                                       add_const=ac.ctypes.data_as(C.POINTER(C.c_int32)),
                                       M=l.M, n=l.n, relu=int(l.relu))
            if getattr(l, "M_oc", None) is not None:      # per-output-channel requant constants (that layer runs on the dot4 kernels)
                moc = np.ascontiguousarray(l.M_oc, dtype=np.uint32)
                noc = np.ascontiguousarray(l.n_oc, dtype=np.uint32)
                if moc.shape != (w.shape[0],) or noc.shape != (w.shape[0],):
                    raise ValueError("per-channel requant constants: one (M, n) per output channel")
                self._keep += [moc, noc]
                layers[k].M_oc = moc.ctypes.data_as(C.POINTER(C.c_uint32))
                layers[k].n_oc = noc.ctypes.data_as(C.POINTER(C.c_uint32))
        zero = (C.c_int32 * (L + 1))(*bundle.zero)
        desc = _lib.NetDesc(n_layers=L, layers=layers, zero=zero,
                            scale_in=float(np.float32(bundle.scale[0])), scale_out=float(np.float32(bundle.scale[L])),
                            M_res=bundle.M_res, n_res=bundle.n_res, pixel_shuffle=bundle.pixel_shuffle,
                            pe_num=bundle.pe_num, pe_acc_bits=bundle.pe_acc_bits, pe_add_bits=bundle.pe_add_bits)
        # options are part of sesrq_create: the device net is immutable afterwards
        opts = _lib.Options()
        _lib.lib().sesrq_default_options(C.byref(opts))
        opts.engine = int(engine)
        opts.force_general = int(bool(force_general))
        if exact_division and reciprocal_division:
            raise ValueError("exact_division and reciprocal_division exclude each other")
        opts.exact_div = 2 if reciprocal_division else int(bool(exact_division))
        opts.anchor_add = int(bool(anchor_add))
        opts.fuse_hidden = 1 if fuse_hidden is True else int(fuse_hidden)     # 0 per layer, 1 (default) fused hidden trios
        opts.wg_budget = int(wg_budget)
        opts.reduced_forms = int(reduced_forms)
        if upstream is not None:
            opts.i8_in_scale = float(np.float32(upstream.scale[upstream.L]))
            opts.i8_in_zero = int(upstream.zero[upstream.L])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().sesrq_create(C.byref(desc), C.byref(opts), C.byref(handle)), ValueError)
        self._h = handle
        self._ws: Dict[tuple, torch.Tensor] = {}

    def close(self):
        if getattr(self, "_op_id", None):      # registered as a torch.ops.sesrq.forward handle: the C++ side drops its pointer first
            from . import torch_op
            torch_op.unregister_engine(self)
        if getattr(self, "_h", None):
            _lib.lib().sesrq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def fast_division_proven(self) -> bool:
        return bool(_lib.lib().sesrq_fast_division_proven(self._h))

    def one_fma_layers(self):
        """Per layer, the form of its requant (sesrq_layer_one_fma, proven per (M, n) at create): 1 = ONE fused multiply-add,
        2 = one fma that also subtracts the 128 plus the add back (output layer only), 0 = the two-step form."""
        return [int(_lib.lib().sesrq_layer_one_fma(self._h, k)) for k in range(self.bundle.L)]

    def layer_engines(self):
        return [(_lib.lib().sesrq_layer_engine(self._h, k) or b"").decode() for k in range(self.bundle.L)]

    def launch_plan(self):
        """[(first_layer, n_layers)] per kernel launch of forward() (n_layers == 3: fused hidden trio)."""
        L = self.bundle.L
        first, count = (C.c_int * L)(), (C.c_int * L)()
        n = _lib.lib().sesrq_launch_plan(self._h, first, count)
        return [(first[i], count[i]) for i in range(n)]

    def out_shape(self, N, H, W):
        r = self.bundle.pixel_shuffle
        return (N, self.bundle.out_channels, H * r, W * r)

    def workspace(self, N, H, W, slot: int = 0) -> torch.Tensor:
        """Device workspace for (N,H,W); `slot` selects an independent copy so that several frames can be
        in flight on different streams (the library itself keeps no per-call state)."""
        key = (N, H, W, slot)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.lib().sesrq_workspace_bytes(self._h, N, H, W)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _enter_stream(self, stream, *tensors):
        """Resolve the stream to launch on.  A side stream is ordered after the work already queued on the
        current stream (which may still be producing `x` or own the memory of freshly allocated outputs),
        and every tensor the launch touches is recorded on it so the caching allocator does not recycle it
        while the kernels run."""
        cur = torch.cuda.current_stream(self.device)
        if stream is None or stream == cur:
            return cur
        stream.wait_stream(cur)
        for t in tensors:
            if t is not None:
                t.record_stream(stream)
        return stream

    def _check_in(self, x: torch.Tensor):
        if x.dim() != 4:
            raise ValueError("Expect input tensor dimension: 4, but get %d" % x.dim())
        if x.device != self.device:
            raise ValueError(f"input is on {x.device}, engine on {self.device}")
        if x.shape[0] < 1 or x.shape[2] < 1 or x.shape[3] < 1:
            raise ValueError("empty frame: N, H and W must be positive")
        if x.shape[1] != self.bundle.in_channels:
            raise ValueError(f"expected {self.bundle.in_channels} input channels, got {x.shape[1]}")
        if x.dtype == torch.float32:
            return _lib.F32
        if x.dtype == torch.int8:
            return _lib.I8
        raise ValueError("input must be float32 (frame) or int8 (already quantised q0)")

    def forward(self, x: torch.Tensor, want_q: bool = True, want_f: bool = True, out_q=None, out_f=None, stream=None,
                slot: int = 0, assume_ordered: bool = False):
        """x: (N, Cin, H, W) float32 | int8 on self.device -> (q int8 | None, y float32 | None).
        stream: torch.cuda.Stream to enqueue on (default: current); slot: workspace copy to use -- give
        concurrent in-flight frames different slots.  A side stream is ordered behind the current stream and the
        tensors are recorded on it (safe default); assume_ordered=True skips that (and the device switch) for a caller
        that owns persistent, contiguous buffers and fences its streams itself -- the throughput harness: the
        bookkeeping costs more host time per call than the three kernel launches."""
        dt = self._check_in(x)
        if assume_ordered:
            if not x.is_contiguous() or (want_q and out_q is None) or (want_f and out_f is None) or stream is None:
                raise ValueError("assume_ordered=True needs a contiguous input, caller-owned output buffers and a stream")
            N, _, H, W = x.shape
            ws = self.workspace(N, H, W, slot)
            _lib.check(_lib.lib().sesrq_forward(self._h, x.data_ptr(), dt, out_q.data_ptr() if want_q else None,
                                                out_f.data_ptr() if want_f else None, N, H, W, ws.data_ptr(), ws.numel(),
                                                stream.cuda_stream))
            return (out_q if want_q else None), (out_f if want_f else None)
        with torch.cuda.device(self.device):
            x = x.contiguous()
            N, _, H, W = x.shape
            shp = self.out_shape(N, H, W)
            if want_q and out_q is None:
                out_q = torch.empty(shp, dtype=torch.int8, device=self.device)
            if want_f and out_f is None:
                out_f = torch.empty(shp, dtype=torch.float32, device=self.device)
            ws = self.workspace(N, H, W, slot)
            st = self._enter_stream(stream, x, out_q, out_f, ws).cuda_stream
            rc = _lib.lib().sesrq_forward(self._h, x.data_ptr(), dt, out_q.data_ptr() if out_q is not None else None,
                                          out_f.data_ptr() if out_f is not None else None, N, H, W, ws.data_ptr(),
                                          ws.numel(), st)
        _lib.check(rc)
        return out_q, out_f

    __call__ = forward

    def submission(self, frames, outs_q, streams, outs_f=None, group: int = 1):
        """A prepared batch of independent forwards for sesrq_forward_many: frame k = frames[k] -> outs_q[k] (/ outs_f[k]; either may be None) on
        streams[k % len(streams)] with workspace slot k % len(streams).  Returns a Submission; .enqueue(count, first=0) issues frames
        first .. first+count-1 of the (cyclically repeated) list with ONE call into the library -- for callers whose frames are so small
        that the host's per-call cost bounds the rate.  group = G > 1 (single-image frames only): the per-stream workspaces are sized
        for G frames, and the library then runs up to G consecutive frames of a stream as the images of one launch sequence.
        Caller-owned persistent buffers; the caller fences the streams."""
        return Submission(self, frames, outs_q, outs_f, streams, group)

    def capture(self, x: torch.Tensor, want_q: bool = True, want_f: bool = False, slot: int = 0, downstream=()):
        """Capture one forward on `x` (and then, optionally, the chained `downstream` engines on its int8 output) as a HIP
        graph.  Returns a GraphedForward: `.replay()` re-issues the kernels on the current stream with ONE host call
        (x, q, y are fixed buffers: overwrite x in place -- stream-ordered -- to process another frame).  The kernels and
        their results are exactly those of forward(); what changes is the host cost per frame."""
        return GraphedForward(self, x, want_q, want_f, slot, tuple(downstream))

    def forward_timed(self, x: torch.Tensor, iters: int = 10):
        """Measurement hook (sesrq_forward_timed): average device ms per kernel launch (see launch_plan())
        and per forward, HIP events on the launch stream."""
        dt = self._check_in(x)
        with torch.cuda.device(self.device):
            x = x.contiguous()
            N, _, H, W = x.shape
            shp = self.out_shape(N, H, W)
            out_q = torch.empty(shp, dtype=torch.int8, device=self.device)
            ws = self.workspace(N, H, W)
            st = torch.cuda.current_stream(self.device).cuda_stream
            nl = len(self.launch_plan())
            launch_ms = (C.c_float * nl)()
            fwd = C.c_float()
            _lib.check(_lib.lib().sesrq_forward_timed(self._h, x.data_ptr(), dt, out_q.data_ptr(), None, N, H, W,
                                                      ws.data_ptr(), ws.numel(), st, iters, launch_ms, C.byref(fwd)))
        return list(launch_ms), float(fwd.value)

    def forward_debug(self, x: torch.Tensor, pe: bool = True, overflow: bool = False, acts: bool = True, special: bool = False):
        """Forward with the reference's dump taps (define.py *_W_FLG): returns a dict with
        q_out, y, input{k} (int8 NCHW), pe_out{k} (N,4,OC,H,W int32), pe_add{k} (N,OC,H,W int32) and, with
        overflow=True, `overflow` (L,2) int32: PE sums above / below the accumulator range before saturation --
        the events the reference prints as max_overflow / min_overflow (quan_func.py:358-361).
        On the MFMA engine the PE taps are written by the per-PE MFMA kernels themselves; the quantised-input tap
        (input0), the overflow counters and the pe-split last layer (OC <= 4) run their layer on the dot4 kernels.
        acts=False: no input{k} taps (layer 0 then stays on its MFMA kernel too).
        special=True: also `shortcut` (N, OC_0, H, W) fp32 = the reference's residual/shortcut_tensor.pt (layer 0's un-rounded ReLU'd requant
        output, quan_func.py:529-549) and `input4_special` (N, OC_{L-2}, H, W) int8 = input.4.spcial.pt (ic, quan_func.py:250-254); both are
        taps of the dot4 kernels (layers 0 and L-2 then run there)."""
        dt = self._check_in(x)
        with torch.cuda.device(self.device):
            x = x.contiguous()
            N, _, H, W = x.shape
            L = self.bundle.L
            shp = self.out_shape(N, H, W)
            res = {"q_out": torch.empty(shp, dtype=torch.int8, device=self.device),
                   "y": torch.empty(shp, dtype=torch.float32, device=self.device)}
            taps = _lib.Taps()
            for k, l in enumerate(self.bundle.layers):
                oc, ic = l.wq.shape[0], l.wq.shape[1]
                if acts:
                    res[f"input{k}"] = torch.empty((N, ic, H, W), dtype=torch.int8, device=self.device)
                    taps.act[k] = res[f"input{k}"].data_ptr()
                if pe:
                    res[f"pe_out{k}"] = torch.empty((N, 4, oc, H, W), dtype=torch.int32, device=self.device)
                    res[f"pe_add{k}"] = torch.empty((N, oc, H, W), dtype=torch.int32, device=self.device)
                    taps.pe_out[k] = res[f"pe_out{k}"].data_ptr()
                    taps.pe_add[k] = res[f"pe_add{k}"].data_ptr()
            if special:
                l0, lm = self.bundle.layers[0], self.bundle.layers[L - 2]
                res["shortcut"] = torch.empty((N, l0.wq.shape[0], H, W), dtype=torch.float32, device=self.device)
                res["input4_special"] = torch.empty((N, lm.wq.shape[0], H, W), dtype=torch.int8, device=self.device)
                taps.shortcut = res["shortcut"].data_ptr()
                taps.ic = res["input4_special"].data_ptr()
            if overflow:
                res["overflow"] = torch.zeros((L, 2), dtype=torch.int32, device=self.device)
                taps.overflow = res["overflow"].data_ptr()
            ws = self.workspace(N, H, W)
            st = torch.cuda.current_stream(self.device).cuda_stream
            rc = _lib.lib().sesrq_forward_debug(self._h, x.data_ptr(), dt, res["q_out"].data_ptr(), res["y"].data_ptr(),
                                                N, H, W, ws.data_ptr(), ws.numel(), st, C.byref(taps))
        _lib.check(rc)
        return res
