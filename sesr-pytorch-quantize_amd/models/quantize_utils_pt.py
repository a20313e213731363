"""``quantize.prepare()`` for the integer path: what a ``*_qat_G.pth`` checkpoint needs to be folded.

The reference consumes its QAT checkpoints by wrapping every conv in a ``QuantConv2d`` (reference
models/quantize_utils_pt.py:331-512, ``prepare`` :801-835, called at sim.py:64-66 with ``a_bits=8, w_bits=8,
q_type=0, q_level="C"``) and then calling ``collapse()`` while the net is in training mode (sim.py:63,76).  The
fold therefore runs the block's two convs on a delta image THROUGH their fake-quantisers, and because a fresh
observer takes its first range from the tensor it sees (MinMaxObserver / MovingAverageMinMaxObserver
``num_flag == 0``, :68-81, :107-120) the observer state stored in the checkpoint plays no part: the collapsed
weights depend on the conv weights alone.  Per conv, for its input x and weight w (Quantizer.forward :220-246,
SymmetricQuantizer.update_qparams :299-312, Round :150-166; ``q_level="C"`` is not the integer 0 the layer
constructor tests for, so both quantisers are per-TENSOR, :376-404):

    fq(t, lo, hi):  s = max(max(|min t|, |max t|) / ((hi - lo) / 2), eps_f32)
                    q = clamp(sign(t/s) * floor(|t/s| + 0.5), lo, hi)          # round half away from zero
                    return q * s
    y = conv2d(fq(x, -128, 127), fq(w, -127, 127), bias)

Only this inference-time fold is implemented -- observers' running averages, the straight-through backward, the
asymmetric / histogram / per-channel variants, QuantReLU / QuantAdd and BN fusing are training machinery outside the
hot path ("out of scope").  The QuantAdd state of the two long-skip adds (``add_residual``,
``add_upsampled_input``) that a QAT checkpoint also carries is accepted and ignored by the loader (sim.py): the
integer path merges the skip in the integer domain and never evaluates it.
"""
import copy

import torch
import torch.nn.functional as F
from torch import nn

def fake_quantize(t, lo, hi):
    """Symmetric per-tensor fake quantisation with the range taken from ``t`` itself (first call of a fresh observer)."""
    eps = torch.tensor(torch.finfo(torch.float32).eps, dtype=torch.float32)
    span = torch.max(torch.abs(torch.min(t)), torch.abs(torch.max(t)))
    s = torch.max(span / (float(hi - lo) / 2), eps)
    u = t / s
    q = torch.clamp(torch.sign(u) * torch.floor(torch.abs(u) + 0.5), lo, hi)
    return q * s


class _QuantizerState(nn.Module):
    """Holds the buffers a reference Quantizer saves, so a QAT state_dict loads key for key.  Values are not used."""

    def __init__(self, lo, hi):
        super().__init__()
        self.register_buffer("scale", torch.ones(1))
        self.register_buffer("zero_point", torch.zeros(1))
        self.register_buffer("eps", torch.tensor(torch.finfo(torch.float32).eps))
        self.register_buffer("quant_min_val", torch.tensor(float(lo)))
        self.register_buffer("quant_max_val", torch.tensor(float(hi)))
        self.observer = nn.Module()
        self.observer.register_buffer("min_val", torch.zeros(1))
        self.observer.register_buffer("max_val", torch.zeros(1))


class QuantConv2d(nn.Conv2d):
    """A conv whose forward fake-quantises its input (8-bit activations: [-128, 127]) and its weight ([-127, 127])
    per tensor, ranges observed on the spot -- the training-mode first call of the reference's layer."""

    def __init__(self, conv: nn.Conv2d, a_bits=8, w_bits=8):
        super().__init__(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation,
                         conv.groups, conv.bias is not None, conv.padding_mode)
        self.weight = conv.weight
        self.bias = conv.bias
        self.a_range = (-(1 << (a_bits - 1)), (1 << (a_bits - 1)) - 1)
        self.w_range = (-((1 << (w_bits - 1)) - 1), (1 << (w_bits - 1)) - 1)
        self.activation_quantizer = _QuantizerState(*self.a_range)
        self.weight_quantizer = _QuantizerState(*self.w_range)

    def forward(self, x):
        return F.conv2d(fake_quantize(x, *self.a_range), fake_quantize(self.weight, *self.w_range), self.bias, self.stride,
                        self.padding, self.dilation, self.groups)


def prepare(model, inplace=False, a_bits=8, w_bits=8, q_type=0, q_level=0, **unsupported):
    """Wrap every ``nn.Conv2d`` of ``model`` for the QAT fold (reference prepare(), :801-835).  Only the
    configuration the reference's integer path uses is accepted: symmetric (``q_type=0``), 8/8 bits, per-tensor
    weight ranges (``q_level`` anything but the integer 0)."""
    if unsupported:
        raise ValueError(f"prepare(): options {sorted(unsupported)} belong to QAT training, which this package does not implement")
    if q_type != 0:
        raise ValueError("prepare(): only symmetric quantisers (q_type=0, reference sim.py:66) are implemented")
    if isinstance(q_level, int) and q_level == 0:
        raise ValueError("prepare(): per-channel weight ranges (q_level=0) are not implemented; the reference's integer "
                         "path passes q_level=\"C\", which selects per-tensor ranges (quantize_utils_pt.py:376-404)")
    if a_bits == 32 or w_bits == 32 or a_bits < 2 or w_bits < 2:
        raise ValueError("prepare(): a_bits / w_bits must be integer widths below 32")
    if not inplace:
        model = copy.deepcopy(model)

    def wrap(module):
        for name, child in list(module.named_children()):
            if isinstance(child, QuantConv2d):
                continue
            if isinstance(child, nn.Conv2d):
                setattr(module, name, QuantConv2d(child, a_bits, w_bits))
            else:
                wrap(child)
    wrap(model)
    return model


def is_qat_state_dict(sd):
    return any("_quantizer." in k for k in sd)
