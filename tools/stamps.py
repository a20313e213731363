"""Diagnostic: per-phase timeline of the persistent MFMA kernels (needs a -DSESRQ_STAMPS build:
make -C sesr-pytorch-quantize_amd/csrc stamps; run with SESRQ_LIB=.../lib/stamps/libsesrq.so)."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
import sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
b = Bundle.load(os.path.join(ROOT, "tests/golden/sesr_x2_rand.crop.npz"))
e = sesrq.Engine(b, torch.device("cuda:0"), engine=_lib.ENGINE_MFMA, fuse_hidden=False, wg_budget=int(os.environ.get('WG_BUDGET', '512')))   # stamps live in the per-layer kernels; the LAST layer's launch is what is read back
x = torch.rand(1, 3, 1080, 1920, device="cuda")
STREAMS = int(os.environ.get("STREAMS", "1"))
if STREAMS > 1:      # the bench plan: FUSED engine, STREAMS streams, sesrq_forward_many for ~2 s -- the last-layer launch read back ran under that load
    e = sesrq.Engine(b, torch.device("cuda:0"), engine=_lib.ENGINE_MFMA, wg_budget=int(os.environ.get('WG_BUDGET', '512')))
    xs = [torch.rand(1, 3, 1080, 1920, device="cuda") for _ in range(8 * STREAMS)]
    outs = [torch.empty(e.out_shape(1, 1080, 1920), dtype=torch.int8, device="cuda") for _ in xs]
    sub = e.submission(xs, outs, [torch.cuda.Stream() for _ in range(STREAMS)])
    for _ in range(int(os.environ.get("ROUNDS", "1500"))): sub.enqueue(len(xs))
else:
    for _ in range(50): e.forward(x, want_f=False)      # int8 frame only: the last-layer instance bench.py times
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
if rt[:, 15].any():      # slot 15: the clocks at kernel entry (xcd_block)
    ent = (rt[:, 15] - rt[:, 0].min()) & 0xffffffff
    ent = np.where(ent > 1 << 31, ent - (1 << 32), ent) / 100.0
    print(f"  {'entry':14s} {np.median(ent):7.2f} {ent.min():7.2f} {ent.max():7.2f}   (relative to the first workgroup's `start`)")
    print("  entry -> start (prologue before the first loads), cycles median:", int(np.median((ct[:, 0] - ct[:, 15]) & 0xffffffff)))
for k, nm in enumerate(names[:16]):
    if rt[:, k].any():
        v = rel[rt[:, k] != 0, k] / 100.0
        print(f"  {nm:14s} {np.median(v):7.2f} {v.min():7.2f} {v.max():7.2f}")
dc = (ct[:, 1:] - ct[:, :-1]) & 0xffffffff
# in-kernel shader clock (MI355X_MICROARCH.md, DVFS give-back item 6): cycles / realtime over the stamped tiles of every workgroup
k0, k1 = 2, max(k for k in range(15) if rt[:, k].all())
dcy = ((ct[:, k1] - ct[:, k0]) & 0xffffffff).astype(np.float64); drt = ((rt[:, k1] - rt[:, k0]) & 0xffffffff).astype(np.float64)
print(f"in-kernel clock over slots {k0}..{k1}: median {np.median(dcy / drt * 100):.0f} MHz (p10 {np.percentile(dcy / drt * 100, 10):.0f}, p90 {np.percentile(dcy / drt * 100, 90):.0f}), streams = {STREAMS}")
print("cycle deltas (median):", [int(np.median(dc[:, k])) for k in range(11)])
# who is late: the fourth tile's end (slot 12) by strip (x) and by vertical run (y) -- edge strips pad, runs differ in length by one tile
if os.environ.get("STAMP_GRID"):
    gx_, gy_ = (int(v) for v in os.environ["STAMP_GRID"].split("x"))
    k_last = max(k for k in range(15) if rt[:, k].all())
    t_last = rel[:gx_ * gy_, k_last].reshape(gy_, gx_) / 100.0
    t_first = rel[:gx_ * gy_, 2].reshape(gy_, gx_) / 100.0
    print(f"slot {k_last} ({names[k_last]}) by strip x: ", " ".join(f"{v:.1f}" for v in np.median(t_last, axis=0)))
    print(f"slot {k_last} by run y:   ", " ".join(f"{v:.1f}" for v in np.median(t_last, axis=1)))
    print("first barrier by strip x:", " ".join(f"{v:.1f}" for v in np.median(t_first, axis=0)))
    print("first barrier by run y:  ", " ".join(f"{v:.1f}" for v in np.median(t_first, axis=1)))
    d = t_last - t_first
    print(f"tiles 0..3 duration: median {np.median(d):.2f} us, p10 {np.percentile(d, 10):.2f}, p90 {np.percentile(d, 90):.2f}, max {d.max():.2f}")
    slow = np.argwhere(d > np.percentile(d, 95))
    print("slowest 5 % (y, x):", [tuple(int(v) for v in p) for p in slow][:40])
