"""The reference's quantisation callables (myQL/quan_func.py), re-hosted on libsesrq.so.

Same names, argument meaning and error behaviour as the reference functions they replace:

    quantize_model_weight            quan_func.py:128   load time, host
    quantize_symmetrical_by_tensor   quan_func.py:44    load time, host  (-> sesrq_quantize_weight)
    quan_layer_between_const         quan_func.py:495   load time, host  (-> sesrq_requant_const)
    quantize_asymmetrical_by_tensor  quan_func.py:161   per-conv stage
    reshape_input_for_hardware_pe    quan_func.py:298   per-conv stage
    PEs_and_bias_adder               quan_func.py:418   per-conv stage
    requan_conv2d_output             quan_func.py:517   per-conv stage

Both execution modes of the reference are covered: exe_mode 1 (integer inference) lowers to
sesrq_forward, exe_mode 0 (calibration: fake-quant float forward + running min/max, test.py) lowers to
sesrq.calibrate.Calibrator (sesrq_calib_conv / _minmax / _fakequant).

In the reference every stage is a float32 torch op that hands its parameters to the next stage
through files under ./output_pt/.  Here the four per-conv stage callables are *markers*: the graph
splicers (myQL/graph_modify.py) insert them exactly like the reference does, and the spliced
module lowers every [quantize -> PE split -> conv -> PE adder -> requant -> activation] chain --
the whole net -- to ONE call of the fused device op (sesrq_forward).  Calling a stage marker
eagerly on a tensor is an error: there is deliberately no per-stage CPU/PyTorch path.
Parameters live in sesrq.store.STORE (keys = the reference's output_pt file names).
"""
import copy

import numpy as np
import torch
from torch import nn

from define import QUAN_BIT, REQUAN_BIT, REQUAN_N_MAX  # noqa: F401  (same import surface as the reference)
import sesrq
from sesrq.store import STORE


def remove_suffix(string: str) -> str:
    return string.rsplit(".", 1)[0]


def float_to_hex(item, bit_width: int) -> str:
    """signed integer value -> two's-complement hex string of ceil(bit_width/4) digits (min 2)."""
    digits = max(2, -(-bit_width // 4))
    return format(int(item) & ((1 << bit_width) - 1), "0{}x".format(digits))


def quan_layer_between_const(input, data_bit=16, shift_max=32):
    assert data_bit < shift_max, "requan data bit must be less than shift_max"
    return sesrq.requant_const(float(input), data_bit, shift_max)


def quantize_symmetrical_by_tensor(tensor_input: torch.Tensor, width: int, exe_mode: int, func_id: int = None,
                                   filename: str = None) -> torch.Tensor:
    """exe mode 0: fake-quantised float weights; exe mode 1: integer-valued weights."""
    w = tensor_input.detach().cpu().numpy().astype(np.float32)
    assert float(np.abs(w).max()) > 0, "Conv2d weight tensor is all zero"
    import define
    if getattr(define, "WEIGHT_PER_CHANNEL", False):
        # this package's extension (define.WEIGHT_PER_CHANNEL; the reference is per tensor): one scale per output channel, stored as an [OC]
        # tensor under the same key -- the lowering then derives per-channel requant constants (sesrq_layer_desc.M_oc)
        wq, scales = sesrq.quantize_weight_per_channel(w, width)
        STORE[f"weight/conv.weight.{func_id}.scale"] = torch.from_numpy(scales)
        STORE[f"weight/conv.weight.{func_id}"] = torch.from_numpy(wq.astype(np.float32))
        q = torch.from_numpy(wq.astype(np.float32)).to(tensor_input.device)
        return q * torch.from_numpy(scales.astype(np.float32)).to(q.device).reshape(-1, 1, 1, 1) if exe_mode == 0 else q
    wq, scale = sesrq.quantize_weight(w, width)
    STORE[f"weight/conv.weight.{func_id}.scale"] = scale
    STORE[f"weight/conv.weight.{func_id}"] = torch.from_numpy(wq.astype(np.float32))
    q = torch.from_numpy(wq.astype(np.float32)).to(tensor_input.device)
    return q * scale if exe_mode == 0 else q


def quantize_model_weight(model_input: nn.Module, weight_width: int, exe_mode: int):
    """Quantise every nn.Conv2d weight of a deep copy of the model; conv ordinal = func_id."""
    model = copy.deepcopy(model_input)
    params = model.state_dict()
    modules = dict(model.named_modules())
    conv_id = 0
    for name in params:
        owner = modules[remove_suffix(name)]
        if type(owner) is not nn.Conv2d:
            continue
        if "weight" in name:
            params[name] = quantize_symmetrical_by_tensor(params[name], weight_width, exe_mode, func_id=conv_id)
            conv_id += 1
        elif "bias" not in name:
            raise KeyError("Unsupported state dict type found. (%s)" % name)
    model.load_state_dict(params)
    return model


class _StageMarker:
    """Callable inserted into the fx graph; consumed by sesrq.lowering, never executed eagerly."""

    def __init__(self, name, doc):
        self.__name__ = self.__qualname__ = name
        self.__doc__ = doc
        self.__module__ = __name__

    def __call__(self, *args, **kwargs):
        raise RuntimeError(
            f"{self.__name__} is a stage marker: splice it with myQL.graph_modify and call the spliced model; the "
            "chain runs as one fused device op (there is no per-stage fallback)")


quantize_asymmetrical_by_tensor = _StageMarker(
    "quantize_asymmetrical_by_tensor",
    "(tensor_input, width, exe_mode, func_id=None): per-tensor asymmetric INT8 activation quantiser / requant finish.")
PEs_and_bias_adder = _StageMarker(
    "PEs_and_bias_adder",
    "(input_tensor, bias, pe_add_width, pe_acc_width, bias_width, func_id, pe_num, exe_mode): 18-bit PE clamp, "
    "PE sum, 20-bit clamp, 16-bit bias constant.")
requan_conv2d_output = _StageMarker(
    "requan_conv2d_output", "(input_tensor, func_id, exe_mode): acc * M * 2^-n requantisation, role by func_id.")


def reshape_input_for_hardware_pe(input_tensor, pe_num: int = 4):
    """(B,C,H,W) -> (B*pe_num,C,H,W), copy p keeps channels c = p (mod pe_num).  In the device path the
    PE split is a channel ORDER inside the NHWC tile, so this marker lowers to nothing; called eagerly it
    only validates its argument like the reference does."""
    input_dimension = len(input_tensor.shape)
    assert input_dimension == 4, 'Expect input tensor dimension: 4, but get %d' % input_dimension
    raise RuntimeError("reshape_input_for_hardware_pe is a stage marker (see module docstring)")
