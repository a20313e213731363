"""Deeper NRDM (6 residual blocks, 8 convs).  The reference cannot int-simulate this depth (its roles
are hard-coded for 5 convs); here roles generalise by position -- parity UNPINNED."""
from models.model_utils_pt import CollapsibleNet


class nr(CollapsibleNet):
    def __init__(self, in_channels=3, out_channels=3, num_channels=16, num_lblocks=6, scaling_factor=1):
        super().__init__(in_channels, out_channels, num_channels, num_lblocks, scaling_factor)
