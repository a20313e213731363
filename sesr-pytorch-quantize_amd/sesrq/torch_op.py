"""`torch.ops.sesrq.forward` -- the whole integer forward as ONE registered torch operator ("the mode-1 graph
collapses ... into one op torch.ops.sesrq.forward(x, bundle)").

    q, y = torch.ops.sesrq.forward(x, engine_id)

x: (N, Cin, H, W) float32 frame (or int8 q0) on a HIP device; engine_id: handle of a sesrq.Engine registered with
register_engine() (an operator schema cannot carry a Python object, so the immutable device net travels as an int);
q: (N, Cout, H*r, W*r) int8 = input.L.pt after PixelShuffle; y: same shape, float32 -- what the reference's model(inps)
returns (sim.py:205).  The implementation is the C ABI call (sesrq_forward through ctypes): torch is plumbing.
A fake (meta) kernel gives shapes/dtypes for tracing; there is no CPU kernel -- calling the op on a CPU tensor fails loudly."""
from __future__ import annotations

import itertools
import weakref
from typing import Tuple

import torch

_ENGINES = weakref.WeakValueDictionary()
_ids = itertools.count(1)


def register_engine(engine) -> int:
    eid = next(_ids)
    _ENGINES[eid] = engine
    engine._op_id = eid
    return eid


def _engine(eid: int):
    e = _ENGINES.get(int(eid))
    if e is None:
        raise RuntimeError(f"sesrq::forward: engine handle {eid} is not registered (or its Engine was destroyed)")
    return e


@torch.library.custom_op("sesrq::forward", mutates_args=(), device_types="cuda")
def forward(x: torch.Tensor, engine_id: int) -> Tuple[torch.Tensor, torch.Tensor]:
    q, y = _engine(engine_id).forward(x)
    return q, y


@forward.register_fake
def _(x, engine_id):
    e = _engine(engine_id)
    if x.dim() != 4:
        raise ValueError("Expect input tensor dimension: 4, but get %d" % x.dim())
    shp = e.out_shape(x.shape[0], x.shape[2], x.shape[3])
    return x.new_empty(shp, dtype=torch.int8), x.new_empty(shp, dtype=torch.float32)


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
