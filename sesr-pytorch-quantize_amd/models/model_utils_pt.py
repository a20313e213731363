"""Collapsible linear blocks (SESR): a k x k conv into a wide hidden width followed by a 1 x 1
conv back down, trained over-parameterised and folded analytically into ONE k x k conv for
inference.  Module and parameter names follow the reference (models/model_utils_pt.py:5-92) so its
checkpoints load unchanged; the fold itself is a closed-form contraction instead of the
reference's "push a delta image through the block":

    W[o, i, ky, kx] = sum_t  S[o, t] * E[t, i, ky, kx]        bias = squeeze bias
    residual block:  W[c, c, mid, mid] += 1   (the short skip becomes part of the weights)

(Float summation order differs from the reference's conv-based fold, so collapsed weights agree to
rounding; parity of the integer path is pinned on the INT8 bundle, SURVEY 7 "hard parts".)  A block prepared for a QAT
checkpoint (models/quantize_utils_pt.py) is not linear and is folded by its impulse response instead."""
import torch
from torch import nn

_ACTIVATIONS = {"prelu": nn.PReLU, "relu": nn.ReLU, "identity": nn.Identity}


class CollapsibleLinearBlock(nn.Module):
    def __init__(self, in_channels, out_channels, tmp_channels, kernel_size, activation="prelu"):
        super().__init__()
        if activation not in _ACTIVATIONS:
            raise Exception(f"Activation not supported: {activation}")
        self.conv_expand = nn.Conv2d(in_channels, tmp_channels, (kernel_size, kernel_size),
                                     padding=(kernel_size - 1) // 2, bias=False)
        self.conv_squeeze = nn.Conv2d(tmp_channels, out_channels, (1, 1))
        self.activation = _ACTIVATIONS[activation]()
        self.collapsed = False

    def forward(self, x):
        y = self.conv_expand(x)
        if not self.collapsed:
            y = self.conv_squeeze(y)
        return self.activation(y)

    def _folded(self):
        if type(self.conv_expand) is not nn.Conv2d:
            return self._folded_by_probe()
        expand, squeeze = self.conv_expand.weight, self.conv_squeeze.weight[:, :, 0, 0]
        return torch.einsum("ot,tikl->oikl", squeeze, expand), self.conv_squeeze.bias

    def _folded_by_probe(self):
        """Fold of a block whose convs are not plain linear maps (models/quantize_utils_pt.py: fake-quantising convs of
        a QAT checkpoint): the block's response to one unit impulse per input channel IS the reference's definition of
        the collapsed kernel there (models/model_utils_pt.py:40-56 pushes a delta image through conv_squeeze(conv_expand(.))
        and subtracts the bias), so the same ops run here, on the same tensor shapes, to get the same float32 bits."""
        cin, k = self.conv_expand.in_channels, self.conv_expand.kernel_size[0]
        probe = torch.zeros(cin, cin, k, k, device=self.conv_expand.weight.device)
        probe[torch.arange(cin), torch.arange(cin), k // 2, k // 2] = 1.0
        bias = self.conv_squeeze.bias
        response = self.conv_squeeze(self.conv_expand(probe)) - bias[None, :, None, None]
        return response.flip(2, 3).transpose(0, 1), bias

    def collapse(self):
        if self.collapsed:
            print("Already collapsed")
            return
        with torch.no_grad():
            weight, bias = self._folded()
            k = self.conv_expand.kernel_size[0]
            conv = nn.Conv2d(self.conv_expand.in_channels, weight.shape[0], (k, k), padding=k // 2)
            conv.weight.copy_(weight)
            conv.bias.copy_(bias)
        self.conv_expand = conv.to(weight.device)
        self.conv_squeeze = nn.Identity()
        self.collapsed = True


class ResidualCollapsibleLinearBlock(CollapsibleLinearBlock):
    """y = act(x + squeeze(expand(x))); after collapse the identity lives on the centre tap."""

    def forward(self, x):
        if self.collapsed:
            return self.activation(self.conv_expand(x))
        return self.activation(x + self.conv_squeeze(self.conv_expand(x)))

    def collapse(self):
        if self.collapsed:
            print("Already collapsed")
            return
        super().collapse()
        mid = self.conv_expand.kernel_size[0] // 2
        with torch.no_grad():
            idx = torch.arange(self.conv_expand.in_channels)
            self.conv_expand.weight[idx, idx, mid, mid] += 1.0


class AddOp(nn.Module):
    def forward(self, x1, x2):
        return x1 + x2


class CollapsibleNet(nn.Module):
    """conv_first (5x5) -> num_lblocks residual 3x3 blocks -> conv_last (5x5) [-> PixelShuffle].
    The long skip (conv_first output added before conv_last) is NOT part of this module: the integer
    path merges it in the integer domain at the input of conv_last (reference models/*_sim.py)."""

    def __init__(self, in_channels, out_channels, num_channels=16, num_lblocks=3, scaling_factor=1, tmp_channels=256):
        super().__init__()
        self.conv_first = CollapsibleLinearBlock(in_channels, num_channels, tmp_channels, 5, activation="relu")
        self.residual_block = nn.Sequential(*[
            ResidualCollapsibleLinearBlock(num_channels, num_channels, tmp_channels, 3, activation="relu")
            for _ in range(num_lblocks)])
        self.conv_last = CollapsibleLinearBlock(num_channels, out_channels * scaling_factor ** 2, tmp_channels, 5,
                                                activation="identity")
        if scaling_factor > 1:
            self.depth_to_space = nn.PixelShuffle(scaling_factor)
        self.scaling_factor = scaling_factor

    def collapse(self):
        self.conv_first.collapse()
        for block in self.residual_block:
            block.collapse()
        self.conv_last.collapse()

    def before_quantization(self):
        self.collapse()

    def forward(self, input):
        y = self.conv_last(self.residual_block(self.conv_first(input)))
        return self.depth_to_space(y) if self.scaling_factor > 1 else y
