"""Diagnostic: where do the cycles of a fused-trio step go?  (needs the -DSESRQ_STAMPS build: make -C sesr-pytorch-quantize_amd/csrc stamps;
run with SESRQ_LIB=.../lib/stamps/libsesrq.so).  Every wave stamps s_memtime at: 0 step start, 1 end of phase a, 2 after barrier 1,
3 after the shift writes, 4 end of phase b + staging store, 5 after barrier 2, 6 end of phase c, 7 end of step.
(profiles/r04_trio_stamps_before.txt was taken with round 3's step: staging store at 3, a third barrier at 7.)"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
import sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
budget = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = Bundle.load(os.path.join(ROOT, "tests/golden/sesr_x2_rand.crop.npz"))
e = sesrq.Engine(b, torch.device("cuda:0"), wg_budget=budget)
x = torch.rand(1, 3, 1080, 1920, device="cuda")
for _ in range(5): e.forward(x)
torch.cuda.synchronize()
lib = _lib.lib()
buf = np.zeros(1024 * 4 * 12 * 8, np.uint64)
lib.sesrq_debug_fetch_trio_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert lib.sesrq_debug_fetch_trio_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(1024, 4, 12, 8).astype(np.int64)
ok = (st[..., 0] != 0) & (st[..., 7] != 0)
d = st[..., 1:] - st[..., :-1]
names = ["phase a (+ loads issued, rc window)", "wait barrier 1", "shift writes (bufB, bufI)", "phase b + staging store", "wait barrier 2", "phase c (+ shift A write, B read)", "(no third barrier since round 4)"]
print(f"wg_budget {budget}: {int(ok[:, 0].any(axis=1).sum())} workgroups, {int(ok.sum())} wave-steps; shader cycles per step (median / mean / p90):")
tot = (st[..., 7] - st[..., 0])[ok]
for k, nm in enumerate(names):
    v = d[..., k][ok]
    print(f"  {nm:36s} {np.median(v):8.0f} {v.mean():8.0f} {np.percentile(v, 90):8.0f}   {100 * v.mean() / tot.mean():5.1f} %")
print(f"  {'whole step':32s} {np.median(tot):8.0f} {tot.mean():8.0f} {np.percentile(tot, 90):8.0f}")
# per wave: the barrier waits of the fastest and the slowest wave of a workgroup
w = d[..., 1] + d[..., 4] + d[..., 6]
print("  barrier wait per step by wave (mean):", [int(w[:, k][ok[:, k]].mean()) for k in range(4)])
