"""`torch.ops.sesrq.forward` -- the whole integer forward as ONE registered torch operator ("the mode-1 graph
collapses ... into one op torch.ops.sesrq.forward(x, bundle)").

    q, y = torch.ops.sesrq.forward(x, engine_id)
    torch.ops.sesrq.forward_into(x, engine_id, out_q, out_f, workspace)      # caller-owned buffers, no allocation

x: (N, Cin, H, W) float32 frame (or int8 q0) on a HIP device; engine_id: handle of a sesrq.Engine registered with
register_engine() (an operator schema cannot carry a Python object, so the immutable device net travels as an int);
q: (N, Cout, H*r, W*r) int8 = input.L.pt after PixelShuffle; y: same shape, float32 -- what the reference's model(inps)
returns (sim.py:205).

Round 5: the operator is registered in C++ (csrc/torch_op/sesrq_torch_op.cpp: TORCH_LIBRARY(sesrq) + CUDA / Meta kernels), built with
torch.utils.cpp_extension (build_extension(), called by __graft_entry__.build()) into lib/torch_op/sesrq_torch_op.so, which links
libsesrq.so, reads the current HIP stream and calls sesrq_forward -- no Python between the dispatcher and the C ABI (rounds 3-4: a Python
custom_op over ctypes).  ctypes stays for sesrq_create and introspection.  A missing extension is an error at import of this module's
users (lowered_module / register_engine), never a fallback to another implementation."""
from __future__ import annotations

import ctypes as C
import itertools
import os
import weakref

import torch

from . import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.normpath(os.path.join(_HERE, "..", "csrc", "torch_op", "sesrq_torch_op.cpp"))
BUILD_DIR = os.path.normpath(os.path.join(_HERE, "..", "lib", "torch_op"))
EXT_PATH = os.path.join(BUILD_DIR, "sesrq_torch_op.so")

_ENGINES = weakref.WeakValueDictionary()
_ids = itertools.count(1)
_ext = None


def build_extension(verbose: bool = False) -> str:
    """Compile csrc/torch_op/sesrq_torch_op.cpp in-tree with torch.utils.cpp_extension (ninja + the host compiler; no device code) and link
    it against libsesrq.so.  Returns the path of the built library.  Build container only: the GPU box loads the prebuilt file."""
    from torch.utils import cpp_extension
    lib_dir = os.path.dirname(_lib.LIB_PATH)
    if not os.path.isfile(_lib.LIB_PATH):
        raise RuntimeError(f"sesrq: build libsesrq.so first ({_lib.LIB_PATH} is missing)")
    os.makedirs(BUILD_DIR, exist_ok=True)
    _lib.lib()      # cpp_extension loads what it built: libsesrq.so is in the process first, as at run time
    cpp_extension.load(name="sesrq_torch_op", sources=[SRC],
                       extra_include_paths=[os.path.normpath(os.path.join(_HERE, "..", "..", "include"))],
                       extra_cflags=["-O2", "-std=c++17"], with_cuda=True,
                       # $ORIGIN/..: libsesrq.so lives one directory up; a build already loaded under the same SONAME (SESRQ_LIB) wins
                       extra_ldflags=[f"-L{lib_dir}", "-lsesrq", "-Wl,-rpath,'$$ORIGIN/..'"],      # $$: ninja's escape; quotes: the shell's
                       build_directory=BUILD_DIR, is_python_module=False, verbose=verbose)
    if not os.path.isfile(EXT_PATH):
        raise RuntimeError(f"sesrq: cpp_extension did not produce {EXT_PATH}")
    return EXT_PATH


def extension() -> C.CDLL:
    """Load the C++ operator library once (after libsesrq.so, whose SONAME its NEEDED entry resolves to)."""
    global _ext
    if _ext is None:
        _lib.lib()
        if not os.path.isfile(EXT_PATH):
            raise RuntimeError(f"sesrq: the C++ operator library is missing ({EXT_PATH}); build it with __graft_entry__.build() "
                               "(sesrq.torch_op.build_extension()); there is no Python fallback for torch.ops.sesrq.forward")
        torch.ops.load_library(EXT_PATH)
        h = C.CDLL(EXT_PATH)
        h.sesrq_torch_register.restype, h.sesrq_torch_register.argtypes = C.c_int, [C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int]
        h.sesrq_torch_unregister.restype, h.sesrq_torch_unregister.argtypes = C.c_int, [C.c_int64]
        _ext = h
    return _ext


def register_engine(engine) -> int:
    """engine: a sesrq.Engine (device net), or any object with a `bundle` and no `_h` -- a shape-only handle: the operator can then be
    traced (Meta kernel, lowered graph) on a host without a device, and refuses to run."""
    eid = next(_ids)
    b = engine.bundle
    if extension().sesrq_torch_register(eid, getattr(engine, "_h", None), int(b.in_channels), int(b.out_channels) * int(b.pixel_shuffle) ** 2,
                                        int(b.pixel_shuffle)) != 0:
        raise RuntimeError("sesrq::forward: could not register the engine")
    _ENGINES[eid] = engine
    engine._op_id = eid
    return eid


def unregister_engine(engine) -> None:
    """Called by Engine.close(): the C++ side must not keep a pointer to a destroyed net."""
    eid = getattr(engine, "_op_id", None)
    if eid is not None and _ext is not None:
        _ext.sesrq_torch_unregister(eid)
        engine._op_id = None


def lowered_module(engine) -> torch.fx.GraphModule:
    """An fx.GraphModule whose graph is  x -> sesrq::forward(x, id) -> [1] -> output : one op node."""
    eid = getattr(engine, "_op_id", None) or register_engine(engine)
    g = torch.fx.Graph()
    x = g.placeholder("input")
    node = g.call_function(torch.ops.sesrq.forward.default, (x, eid))
    import operator
    y = g.call_function(operator.getitem, (node, 1))
    g.output(y)
    gm = torch.fx.GraphModule(torch.nn.Module(), g)
    gm._sesrq_keepalive = engine
    return gm
