"""ctypes wrapper of the C oracle (oracle/sesrq_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class _Layer(C.Structure):
    _fields_ = [("k", C.c_int32), ("ic", C.c_int32), ("oc", C.c_int32), ("w", C.c_void_p), ("add_const", C.c_void_p),
                ("M", C.c_uint32), ("n", C.c_uint32), ("relu", C.c_int32)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libsesrq_oracle.so"))
        _lib.orc_forward.restype = C.c_int
        _lib.orc_max_threads.restype = C.c_int
    return _lib


def max_threads():
    return int(_load().orc_max_threads())


def forward(net, x, threads=0, keep=False, want_f=True):
    """net: oracle.sesrq_oracle.Net ; x (N,Cin,H,W) fp32 -> dict(q_out, y[, input{k}, pe_out{k}, pe_add{k}])."""
    lib = _load()
    x = np.ascontiguousarray(x, np.float32)
    N, cin, H, W = x.shape
    L = net.L
    keepalive = []
    layers = (_Layer * L)()
    for k, l in enumerate(net.layers):
        if getattr(l, "M_oc", None) is not None:
            raise NotImplementedError("the C oracle restates the reference's per-tensor requant only; per-channel nets (unpinned) use the numpy oracle")
        w = np.ascontiguousarray(l.wq, np.int8)
        a = np.ascontiguousarray(l.add_const, np.int32)
        keepalive += [w, a]
        layers[k] = _Layer(w.shape[2], w.shape[1], w.shape[0], w.ctypes.data, a.ctypes.data, l.M, l.n, int(l.relu))
    zero = (C.c_int32 * (L + 1))(*net.zero)
    r = net.pixel_shuffle
    cout = net.layers[-1].wq.shape[0] // (r * r)
    out_q = np.empty((N, cout, H * r, W * r), np.int8)
    out_f = np.empty((N, cout, H * r, W * r), np.float32) if want_f else None
    res = {}
    sq = (C.c_void_p * L)()
    sp = (C.c_void_p * L)()
    sa = (C.c_void_p * L)()
    if keep:
        for k, l in enumerate(net.layers):
            oc, ic = l.wq.shape[0], l.wq.shape[1]
            res[f"input{k}"] = np.empty((1, ic, H, W), np.int8)
            res[f"pe_out{k}"] = np.empty((4, oc, H, W), np.int32)
            res[f"pe_add{k}"] = np.empty((1, oc, H, W), np.int32)
            sq[k], sp[k], sa[k] = res[f"input{k}"].ctypes.data, res[f"pe_out{k}"].ctypes.data, res[f"pe_add{k}"].ctypes.data
    rc = lib.orc_forward(C.c_int(L), layers, zero, C.c_float(np.float32(net.scale[0])), C.c_float(np.float32(net.scale[L])),
                         C.c_uint32(net.M_res), C.c_uint32(net.n_res), C.c_int(r), C.c_int(net.acc_bits),
                         C.c_int(net.add_bits), C.c_void_p(x.ctypes.data), C.c_int(N), C.c_int(H), C.c_int(W),
                         C.c_void_p(out_q.ctypes.data), C.c_void_p(out_f.ctypes.data if want_f else None),
                         C.c_int(threads), sq if keep else None, sp if keep else None, sa if keep else None)
    if rc != 0:
        raise RuntimeError(f"orc_forward failed: {rc}")
    res["q_out"] = out_q
    if want_f:
        res["y"] = out_f
    return res
