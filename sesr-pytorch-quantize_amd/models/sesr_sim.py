"""SESR x4, single (Y) channel: 1 -> 16 -> ... -> 16 channels, PixelShuffle(4)  (reference models/sesr_sim.py)."""
from models.model_utils_pt import CollapsibleNet


class sesr(CollapsibleNet):
    def __init__(self, in_channels=1, out_channels=1, num_channels=16, num_lblocks=3, scaling_factor=4):
        super().__init__(in_channels, out_channels, num_channels, num_lblocks, scaling_factor)
