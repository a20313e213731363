"""SESR x2, RGB: 3 -> 16 -> ... -> 12 channels, PixelShuffle(2)  (reference models/sesr_arch_sim.py:207)."""
from models.model_utils_pt import CollapsibleNet


class sesr(CollapsibleNet):
    def __init__(self, in_channels=3, out_channels=3, num_channels=16, num_lblocks=3, scaling_factor=2):
        super().__init__(in_channels, out_channels, num_channels, num_lblocks, scaling_factor)
