"""Diagnostic: per-phase timeline of the persistent MFMA kernels (needs a -DSESRQ_STAMPS build:
make -C sesr-pytorch-quantize_amd/csrc stamps; run with SESRQ_LIB=.../lib/stamps/libsesrq.so)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
import sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
b = Bundle.load(os.path.join(ROOT, "tests/golden/sesr_x2_rand.crop.npz"))
e = sesrq.Engine(b, torch.device("cuda:0"), engine=_lib.ENGINE_MFMA, fuse_hidden=False)   # stamps live in the per-layer kernels
x = torch.rand(1, 3, 1080, 1920, device="cuda")
for _ in range(5): e.forward(x)
torch.cuda.synchronize()
lib = _lib.lib()
buf = np.zeros(1 << 20, np.int32)
lib.sesrq_debug_fetch_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.sesrq_debug_fetch_stamps(buf.ctypes.data, buf.nbytes) == 0
nwg = int(np.count_nonzero(buf.reshape(-1, 32)[:, 0]))
st = buf[:nwg * 32].reshape(nwg, 16, 2).astype(np.int64)
rt, ct = st[:, :, 0], st[:, :, 1]
rel = (rt - rt[:, 0].min()) & 0xffffffff
names = ["start", "loads issued", "lds+barrier"] + [f"{p}{t}" for t in range(4) for p in ("compute", "stage", "barrier")]
print(nwg, "workgroups; realtime us (median / min / max):")
for k, nm in enumerate(names[:16]):
    if rt[:, k].any():
        v = rel[rt[:, k] != 0, k] / 100.0
        print(f"  {nm:14s} {np.median(v):7.2f} {v.min():7.2f} {v.max():7.2f}")
dc = (ct[:, 1:] - ct[:, :-1]) & 0xffffffff
print("cycle deltas (median):", [int(np.median(dc[:, k])) for k in range(11)])
