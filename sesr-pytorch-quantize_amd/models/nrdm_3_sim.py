"""NR / DM / NRDM denoise+demosaic net: 3 -> 16 -> ... -> 3 channels, no upscaling (reference models/nrdm_3_sim.py)."""
from models.model_utils_pt import CollapsibleNet


class nr(CollapsibleNet):
    def __init__(self, in_channels=3, out_channels=3, num_channels=16, num_lblocks=3, scaling_factor=1):
        super().__init__(in_channels, out_channels, num_channels, num_lblocks, scaling_factor)
