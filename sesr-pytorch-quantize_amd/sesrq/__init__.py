"""sesrq -- MI355X-native INT8 SESR / NRDM integer inference (host side of libsesrq.so)."""
from . import _lib
from .bundle import Bundle, LayerParams, derive_bundle, requant_const, requant_form, quantize_weight, quantize_weight_per_channel, add_const, calib_scale_zero

__all__ = ["Bundle", "LayerParams", "derive_bundle", "requant_const", "requant_form", "quantize_weight", "quantize_weight_per_channel", "add_const",
           "calib_scale_zero", "Engine"]


def __getattr__(name):
    if name == "Engine":                 # torch is imported only when the device path is used
        from .engine import Engine
        return Engine
    raise AttributeError(name)
