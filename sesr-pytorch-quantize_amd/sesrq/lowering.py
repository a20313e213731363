"""Lowering of a spliced fx graph to the fused device op.

A graph produced by myQL.graph_modify in exe_mode 1 is a straight chain
    [quantize_asymmetrical_by_tensor -> reshape_input_for_hardware_pe -> Conv2d -> PEs_and_bias_adder
     -> requan_conv2d_output -> ReLU|Identity] x L  (-> PixelShuffle)
(reference: sim.py:82-114 builds it, sim.py:205 runs it).  lower() checks that shape, collects the
integer weights, the float biases carried by the bias-bypass nodes, the calibrated activation
domains from the parameter store, derives the integer bundle through libsesrq's host entry points
and returns it; SesrqGraphModule.forward then runs sesrq_forward on the frame."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .bundle import Bundle, derive_bundle_from_quantized
from .store import STORE

_STAGE_ORDER_IN = ("quantize_asymmetrical_by_tensor", "reshape_input_for_hardware_pe")
_STAGE_ORDER_OUT = ("PEs_and_bias_adder", "requan_conv2d_output")


def _fname(node):
    return getattr(node.target, "__name__", None) if node.op == "call_function" else None


def _single_user(node):
    users = list(node.users)
    if len(users) != 1:
        raise RuntimeError(f"sesrq lowering: node '{node.name}' must have exactly one user, has {len(users)}")
    return users[0]


def lower(gm: torch.fx.GraphModule, store=STORE) -> Bundle:
    modules = dict(gm.named_modules())
    convs = [n for n in gm.graph.nodes if n.op == "call_module" and type(modules[n.target]) is nn.Conv2d]
    if len(convs) < 3:
        raise RuntimeError("sesrq lowering: need at least 3 convolutions")
    L = len(convs)
    wq, wsc, biases, relu = [], [], [], []
    widths = None
    prev_tail = None
    for k, c in enumerate(convs):
        # ---- stages in front of the conv
        r = c.args[0]
        if _fname(r) != _STAGE_ORDER_IN[1]:
            raise RuntimeError(f"sesrq lowering: conv {k} is not fed by reshape_input_for_hardware_pe "
                               "(splice order: quantize-before, then reshape-before; sim.py:89-100)")
        q = r.args[0]
        if _fname(q) != _STAGE_ORDER_IN[0]:
            raise RuntimeError(f"sesrq lowering: conv {k} lacks quantize_asymmetrical_by_tensor in front of the PE split")
        if q.kwargs.get("exe_mode") != 1 or q.kwargs.get("func_id") != k or q.kwargs.get("width") != 8:
            raise RuntimeError(f"sesrq lowering: conv {k}: quantiser must be exe_mode=1, width=8, func_id={k} "
                               f"(got {dict(q.kwargs)}); exe_mode 0 (calibration) is not lowered")
        if int(r.kwargs.get("pe_num", 4)) != 4:
            raise RuntimeError("sesrq lowering: pe_num must be 4")
        src = q.args[0]
        if k == 0:
            if src.op != "placeholder":
                raise RuntimeError("sesrq lowering: first conv chain must start at the graph input")
        elif src is not prev_tail:
            raise RuntimeError(f"sesrq lowering: conv {k} is not fed by the previous chain (non-sequential graph)")
        # ---- stages behind the conv
        pb = _single_user(c)
        if _fname(pb) != _STAGE_ORDER_OUT[0]:
            raise RuntimeError(f"sesrq lowering: conv {k} is not followed by PEs_and_bias_adder (insert_bias_bypass)")
        rq = _single_user(pb)
        if _fname(rq) != _STAGE_ORDER_OUT[1]:
            raise RuntimeError(f"sesrq lowering: conv {k}: PEs_and_bias_adder is not followed by requan_conv2d_output")
        if pb.kwargs.get("exe_mode") != 1 or rq.kwargs.get("exe_mode") != 1:
            raise RuntimeError("sesrq lowering: PEs_and_bias_adder / requan_conv2d_output must be exe_mode=1")
        if pb.kwargs.get("func_id") != k or rq.kwargs.get("func_id") != k:
            raise RuntimeError(f"sesrq lowering: conv {k}: stage func_id mismatch")
        w = (int(pb.kwargs["pe_add_width"]), int(pb.kwargs["pe_acc_width"]), int(pb.kwargs["bias_width"]), int(pb.kwargs["pe_num"]))
        if widths is None:
            widths = w
        elif w != widths:
            raise RuntimeError("sesrq lowering: bit widths differ between layers")
        tail = rq
        act = list(rq.users)
        is_relu = False
        if len(act) == 1 and act[0].op == "call_module" and isinstance(modules[act[0].target], (nn.ReLU, nn.Identity)):
            is_relu = isinstance(modules[act[0].target], nn.ReLU)
            tail = act[0]
        relu.append(is_relu)
        prev_tail = tail
        # ---- parameters
        mod = modules[c.target]
        if mod.stride != (1, 1) or mod.dilation != (1, 1) or mod.groups != 1 or mod.kernel_size[0] != mod.kernel_size[1] \
                or mod.padding != (mod.kernel_size[0] // 2,) * 2:
            raise RuntimeError(f"sesrq lowering: conv {k} must be stride-1 'same' k x k")
        wt = mod.weight.detach().cpu().numpy()
        if not np.array_equal(wt, np.rint(wt)) or wt.min() < -128 or wt.max() > 127:
            raise RuntimeError(f"sesrq lowering: conv {k} weights are not INT8-valued; run "
                               "quantize_model_weight(model, 8, 1) before splicing (sim.py:85)")
        wq.append(wt.astype(np.int8))
        sc_k = store[f"weight/conv.weight.{k}.scale"]
        # a scalar (the reference) or, with define.WEIGHT_PER_CHANNEL, an [OC] tensor: per-channel requant constants for this layer
        wsc.append(np.asarray(sc_k, dtype=np.float64).reshape(-1) if hasattr(sc_k, "shape") and len(getattr(sc_k, "shape", ())) == 1 else float(sc_k))
        biases.append(np.asarray(pb.kwargs["bias"], dtype=np.float32))
    ps = 1
    users = list(prev_tail.users)
    if len(users) == 1 and users[0].op == "call_module" and isinstance(modules[users[0].target], nn.PixelShuffle):
        ps = int(modules[users[0].target].upscale_factor)
        users = list(users[0].users)
    if len(users) != 1 or users[0].op != "output":
        raise RuntimeError("sesrq lowering: unexpected operations after the last conv chain")
    if relu[-1] or not all(relu[:-1]):
        raise RuntimeError("sesrq lowering: expected ReLU after every conv but the last (reference topologies)")
    scale, zero = store.activation_domains(L)
    pe_add, pe_acc, bias_w, pe_num = widths
    import define
    return derive_bundle_from_quantized(wq, wsc, biases, scale, zero, ps, name=type(gm).__name__, bias_bit=bias_w,
                                        requan_bit=define.REQUAN_BIT, requan_n_max=define.REQUAN_N_MAX, pe_num=pe_num,
                                        pe_acc_bits=pe_acc, pe_add_bits=pe_add)


def graph_mode(gm: torch.fx.GraphModule) -> int:
    """exe_mode of the spliced quantiser nodes: 1 = integer inference, 0 = calibration, -1 = none spliced."""
    for n in gm.graph.nodes:
        if _fname(n) == _STAGE_ORDER_IN[0]:
            return int(n.kwargs.get("exe_mode", -1))
    return -1


def lower_calibration(gm: torch.fx.GraphModule, device, store=STORE):
    """A mode-0 graph (reference test.py:79-106: quantiser before every conv and before PixelShuffle, PE
    split, bias bypass; float long skip) -> sesrq.calibrate.Calibrator."""
    from .calibrate import Calibrator
    modules = dict(gm.named_modules())
    convs = [n for n in gm.graph.nodes if n.op == "call_module" and type(modules[n.target]) is nn.Conv2d]
    quantized, biases, widths = [], [], None
    for k, c in enumerate(convs):
        r = c.args[0]
        q = r.args[0] if _fname(r) == _STAGE_ORDER_IN[1] else None
        if q is None or _fname(q) != _STAGE_ORDER_IN[0] or q.kwargs.get("exe_mode") != 0 or q.kwargs.get("func_id") != k:
            raise RuntimeError(f"sesrq calibration: conv {k} must be fed by quantize_asymmetrical_by_tensor(exe_mode=0, "
                               f"func_id={k}) -> reshape_input_for_hardware_pe (test.py:86-100)")
        pb = _single_user(c)
        if _fname(pb) != _STAGE_ORDER_OUT[0] or pb.kwargs.get("exe_mode") != 0:
            raise RuntimeError(f"sesrq calibration: conv {k} must be followed by PEs_and_bias_adder(exe_mode=0)")
        widths = (int(pb.kwargs["pe_add_width"]), int(pb.kwargs["pe_acc_width"]), int(pb.kwargs["bias_width"]))
        wq = np.asarray(store[f"weight/conv.weight.{k}"].cpu().numpy() if hasattr(store[f"weight/conv.weight.{k}"], "cpu")
                        else store[f"weight/conv.weight.{k}"])
        sc_k = store[f"weight/conv.weight.{k}.scale"]
        if hasattr(sc_k, "shape") and len(getattr(sc_k, "shape", ())) == 1:
            raise RuntimeError("sesrq calibration: define.WEIGHT_PER_CHANNEL is an inference-time option of this package (no reference counterpart); "
                               "calibrate the activation domains with the reference's per-tensor weights (test.py), then switch it on")
        quantized.append((wq.astype(np.int8), float(sc_k)))
        biases.append(np.asarray(pb.kwargs["bias"], dtype=np.float32))
    ps = 1
    for n in gm.graph.nodes:
        if n.op == "call_module" and isinstance(modules[n.target], nn.PixelShuffle):
            ps = int(modules[n.target].upscale_factor)
    return Calibrator(None, biases, ps, device, pe_acc_bits=widths[1], pe_add_bits=widths[0], bias_bits=widths[2],
                      quantized=quantized)


class SesrqGraphModule(torch.fx.GraphModule):
    """GraphModule whose forward is the fused device op.  `last_q` keeps the int8 result (input.L.pt
    after PixelShuffle) of the most recent call; the return value is the reference's float tensor."""

    def _sesrq_engine(self, device):
        cache = self.__dict__.setdefault("_sesrq_cache", {})
        key = str(device)
        if key not in cache:
            from .engine import Engine
            bundle = lower(self)
            cache[key] = Engine(bundle, device)
        return cache[key]

    def sesrq_bundle(self) -> Bundle:
        return lower(self)

    def sesrq_lowered(self, device) -> torch.fx.GraphModule:
        """The spliced graph collapsed to ONE operator node: input -> torch.ops.sesrq.forward -> float output."""
        from .torch_op import lowered_module
        return lowered_module(self._sesrq_engine(device))

    def recompile(self):
        # GraphModule installs the generated python forward on the per-instance class; put the fused
        # device forward back on top of it (the generated code stays available as `self.code`).
        out = super().recompile()
        type(self).forward = SesrqGraphModule._fused_forward
        return out

    def _fused_forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise ValueError('Expect input tensor dimension: 4, but get %d' % (x.dim() if isinstance(x, torch.Tensor) else -1))
        if not x.is_cuda:
            raise RuntimeError("sesrq: the integer path runs on the GPU only; move the frame to a HIP device "
                               "(model(inps.cuda())) -- there is no CPU fallback")
        if graph_mode(self) == 0:
            # calibration forward (reference test.py:146): observe ranges, keep them in the store like the
            # reference keeps input.K.{min,max}_val / per-batch scale / zero files
            cal = self.__dict__.get("_sesrq_cal")
            if cal is None or cal.device != x.device:
                cal = lower_calibration(self, x.device)
                self.__dict__["_sesrq_cal"] = cal
            y = cal.observe(x.float())
            for k in range(cal.L + 1):
                STORE[f"input/input.{k}.min_val"] = cal.run_min[k]
                STORE[f"input/input.{k}.max_val"] = cal.run_max[k]
                STORE[f"input/input.{k}.scale"] = cal.last_scale[k]
                STORE[f"input/input.{k}.zero"] = cal.last_zero[k]
            return y
        from . import torch_op
        eng = self._sesrq_engine(x.device)
        eid = getattr(eng, "_op_id", None) or torch_op.register_engine(eng)
        q, y = torch.ops.sesrq.forward(x.float() if x.dtype != torch.int8 else x, eid)     # the registered operator
        self.__dict__["last_q"] = q
        self._dump_taps(eng, x)
        return y

    def _dump_taps(self, eng, x):
        """The reference's dump switches (define.py:23-31): with a *_W_FLG on, the forward also leaves the tensors the
        reference would have written under ./output_pt/ in the parameter store, under the reference's file names
        (myQL/quan_func.py:284, 378, 443, 489, 536-609) -- `STORE.save_output_pt(dir)` then writes the
        tree.  They come from the device engine's debug forward (sesrq_forward_debug), not from a second code path."""
        import define
        flags = {n: bool(getattr(define, n, False)) for n in ("INPUT_W_FLG", "OUTPUT_PE_W_FLG", "OUTPUT_PE_ADD_W_FLG", "BIAS_W_FLG",
                                                              "BIAS_QUAN_W_FLG", "REQUAN_FACTOR_W_FLG")}
        if not any(flags.values()):
            return
        b = eng.bundle
        L = b.L
        if flags["INPUT_W_FLG"] or flags["OUTPUT_PE_W_FLG"] or flags["OUTPUT_PE_ADD_W_FLG"]:
            if (flags["OUTPUT_PE_W_FLG"] or flags["OUTPUT_PE_ADD_W_FLG"]) and x.shape[0] != 1:
                # the reference asserts batch == 1 where it dumps the PE tensors (quan_func.py:373: pe_outputK_P.pt is ONE frame's (OC, H, W));
                # a tree with frame 0's PE sums next to all frames' activations would be silently inconsistent
                raise ValueError("the PE dump flags (OUTPUT_PE_W_FLG / OUTPUT_PE_ADD_W_FLG) need batch 1 like the reference's "
                                 "(quan_func.py:373); got a batch of %d frames" % x.shape[0])
            # (with a dump flag on, every forward runs a second, debug forward on the per-PE kernels: sesrq_forward_debug)
            res = eng.forward_debug(x.float() if x.dtype != torch.int8 else x,
                                    pe=flags["OUTPUT_PE_W_FLG"] or flags["OUTPUT_PE_ADD_W_FLG"], special=flags["INPUT_W_FLG"])
            r = b.pixel_shuffle
            for k in range(L):
                if flags["INPUT_W_FLG"]:
                    STORE[f"input/input.{k}"] = res[f"input{k}"].float().cpu()
                if flags["OUTPUT_PE_W_FLG"]:
                    for p in range(b.pe_num):
                        STORE[f"pe_out/pe_output{k}_{p}"] = res[f"pe_out{k}"][0, p].float().cpu()
                if flags["OUTPUT_PE_ADD_W_FLG"]:
                    STORE[f"pe_add/pe_add_output{k}"] = res[f"pe_add{k}"].float().cpu()
            if flags["INPUT_W_FLG"]:
                # the two tensors the reference writes unconditionally beside the activations (round 5): layer 0's un-rounded ReLU'd requant
                # output (quan_func.py:549) and the merging layer's re-quantised operand ic (quan_func.py:254, "spcial" is the reference's spelling)
                STORE["residual/shortcut_tensor"] = res["shortcut"].cpu()
                STORE[f"input/input.{L - 1}.spcial"] = res["input4_special"].float().cpu()
            if flags["INPUT_W_FLG"]:        # input.L.pt: the int8 result before PixelShuffle (quan_func.py:592)
                qL = res["q_out"].float()
                STORE[f"input/input.{L}"] = (torch.nn.functional.pixel_unshuffle(qL, r) if r > 1 else qL).cpu()
        if flags["BIAS_QUAN_W_FLG"]:
            for k, l in enumerate(b.layers):
                STORE[f"bias/conv.bias.quan{k}"] = torch.from_numpy(np.asarray(l.add_const, dtype=np.float32)).reshape(1, -1, 1, 1)
        if flags["REQUAN_FACTOR_W_FLG"]:
            for k, l in enumerate(b.layers):    # the reference's names: layer k -> "k_k+1" (layer L-2 too, although it targets domain 1)
                STORE[f"requan_factor/requan_{k}_{k + 1}"] = int(l.M)
                STORE[f"requan_factor/n_{k}_{k + 1}"] = int(l.n)
            STORE["requan_factor/requan_res"] = int(b.M_res)
            STORE["requan_factor/n_res"] = int(b.n_res)
