"""ctypes binding of libsesrq.so (C ABI declared in include/sesrq.h).

The library is the product: if it is missing or a symbol cannot be resolved this module
raises immediately -- there is no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SESRQ_LIB selects another build of the same ABI (A/B kernel experiments); never a fallback.
LIB_PATH = os.environ.get("SESRQ_LIB") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libsesrq.so"))

MAX_LAYERS = 16
MAX_CH = 16
F32, I8 = 0, 1
ENGINE_AUTO, ENGINE_DOT4, ENGINE_MFMA = 0, 1, 2
ABI_VERSION = 4


class LayerDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("ic", C.c_int32), ("oc", C.c_int32),
                ("w", C.POINTER(C.c_int8)), ("add_const", C.POINTER(C.c_int32)),
                ("M", C.c_uint32), ("n", C.c_uint32), ("relu", C.c_int32),
                ("M_oc", C.POINTER(C.c_uint32)), ("n_oc", C.POINTER(C.c_uint32))]


class NetDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("layers", C.POINTER(LayerDesc)), ("zero", C.POINTER(C.c_int32)),
                ("scale_in", C.c_float), ("scale_out", C.c_float), ("M_res", C.c_uint32), ("n_res", C.c_uint32),
                ("pixel_shuffle", C.c_int32), ("pe_num", C.c_int32), ("pe_acc_bits", C.c_int32),
                ("pe_add_bits", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("engine", C.c_int32), ("force_general", C.c_int32), ("exact_div", C.c_int32),
                ("anchor_add", C.c_int32), ("fuse_hidden", C.c_int32), ("wg_budget", C.c_int32),
                ("i8_in_scale", C.c_float), ("i8_in_zero", C.c_int32), ("reduced_forms", C.c_int32)]


class CalibConvDesc(C.Structure):
    _fields_ = [("k", C.c_int32), ("ic", C.c_int32), ("oc", C.c_int32), ("w", C.c_void_p), ("qbias", C.c_void_p),
                ("in_scale", C.c_float), ("in_zero", C.c_int32), ("ss", C.c_float), ("acc_lo", C.c_float),
                ("acc_hi", C.c_float), ("add_lo", C.c_float), ("add_hi", C.c_float), ("relu", C.c_int32)]


class FrameIO(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out_q", C.c_void_p), ("out_f", C.c_void_p)]


class Taps(C.Structure):
    _fields_ = [("act", C.c_void_p * MAX_LAYERS), ("pe_out", C.c_void_p * MAX_LAYERS),
                ("pe_add", C.c_void_p * MAX_LAYERS), ("overflow", C.c_void_p), ("shortcut", C.c_void_p), ("ic", C.c_void_p)]


# every symbol include/sesrq.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sesrq_default_options": (None, [C.POINTER(Options)]),
    "sesrq_create": (C.c_int, [C.POINTER(NetDesc), C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "sesrq_destroy": (None, [C.c_void_p]),
    "sesrq_launch_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sesrq_fast_division_proven": (C.c_int, [C.c_void_p]),
    "sesrq_layer_one_fma": (C.c_int, [C.c_void_p, C.c_int]),
    "sesrq_requant_form": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int]),
    "sesrq_net_shape": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sesrq_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "sesrq_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                C.c_void_p, C.c_size_t, C.c_void_p]),
    "sesrq_forward_many": (C.c_int, [C.c_void_p, C.POINTER(FrameIO), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "sesrq_submit_selftest": (C.c_int, [C.c_int, C.c_int]),
    "sesrq_instance_count": (C.c_int, []),
    "sesrq_instance_name": (C.c_char_p, [C.c_int]),
    "sesrq_instance_launches": (C.c_longlong, [C.c_int]),
    "sesrq_forward_debug": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Taps)]),
    "sesrq_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.POINTER(C.c_float),
                                      C.POINTER(C.c_float)]),
    "sesrq_layer_engine": (C.c_char_p, [C.c_void_p, C.c_int]),
    "sesrq_calib_conv": (C.c_int, [C.POINTER(CalibConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]),
    "sesrq_calib_minmax": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sesrq_calib_fakequant": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_void_p]),
    "sesrq_calib_histogram": (C.c_int, [C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "sesrq_requant_const": (C.c_int, [C.c_double, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "sesrq_quantize_weight": (C.c_int, [C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_int8),
                                        C.POINTER(C.c_double)]),
    "sesrq_quantize_weight_per_channel": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_int8),
                                                    C.POINTER(C.c_double)]),
    "sesrq_add_const": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_double, C.c_int,
                                  C.c_double, C.c_int, C.POINTER(C.c_int32)]),
    "sesrq_calib_scale_zero": (C.c_int, [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "sesrq_last_error": (C.c_char_p, []),
    "sesrq_version": (C.c_int, []),
}

_lib = None


def lib() -> C.CDLL:
    """Load libsesrq.so once and bind every declared symbol; raise loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"sesrq: native library not found at {LIB_PATH}. Build it with "
                "`make -C sesr-pytorch-quantize_amd/csrc` (or __graft_entry__.build()); there is no fallback path.")
        # PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, needed by torch as
        # "libamdhip64.so").  Loading libsesrq.so first would pull /opt/rocm's copy in and torch would then
        # load a SECOND runtime; importing torch first makes the loader satisfy our NEEDED entry by SONAME
        # with the runtime torch already mapped -> one HIP runtime per process, shared streams/pointers.
        try:
            import torch  # noqa: F401
        except ImportError:  # standalone C/C++ use of the library: system runtime only
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def instances():
    """{name: launches so far} of every kernel instantiation the library can select (sesrq_instance_*)."""
    l = lib()
    return {l.sesrq_instance_name(i).decode(): int(l.sesrq_instance_launches(i)) for i in range(l.sesrq_instance_count())}


def last_error() -> str:
    return (lib().sesrq_last_error() or b"").decode()


def check(rc: int, exc=RuntimeError) -> None:
    if rc != 0:
        raise exc("sesrq: " + last_error())
