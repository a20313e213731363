#!/usr/bin/env python3
"""Where does the PCIe leg of bench.py (e2e) lose its time?  Pieces timed alone and together, pinned buffers, 96 steps, 3 passes each:
H2D only (fp32 / int8 frame), D2H only (the 24.9 MB int8 4K frame), both directions on two streams, kernels only, and the full loop.
   python tools/e2e_probe.py            (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "sesr-pytorch-quantize_amd")):
    sys.path.insert(0, p)
import torch
import sesrq
from sesrq.bundle import Bundle
dev = torch.device("cuda:0")
b = Bundle.load(os.path.join(ROOT, "tests", "golden", "sesr_x2_rand.crop.npz"))
e = sesrq.Engine(b, dev, wg_budget=512)
depth, frames = 3, 96
x = torch.rand((1, 3, 1080, 1920))
hin32 = [x.clone().pin_memory() for _ in range(depth)]
hin8 = [(x * 100).to(torch.int8).pin_memory() for _ in range(depth)]
din32 = [torch.empty_like(x, device=dev) for _ in range(depth)]
din8 = [torch.empty((1, 3, 1080, 1920), dtype=torch.int8, device=dev) for _ in range(depth)]
dout = [torch.empty(e.out_shape(1, 1080, 1920), dtype=torch.int8, device=dev) for _ in range(depth)]
hout = [torch.empty(e.out_shape(1, 1080, 1920), dtype=torch.int8).pin_memory() for _ in range(depth)]
s_in, s_k, s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
torch.cuda.synchronize()


def timeit(name, fn, nbytes):
    fn(2 * depth)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(frames); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%-44s" % name, " ".join("%7.1f steps/s" % (frames / t) for t in ts), "  %6.1f GB/s" % (nbytes * frames / min(ts) / 1e9) if nbytes else "")


def h2d(hin, din):
    def f(n):
        with torch.cuda.stream(s_in):
            for i in range(n): din[i % depth].copy_(hin[i % depth], non_blocking=True)
    return f
def d2h(n):
    with torch.cuda.stream(s_out):
        for i in range(n): hout[i % depth].copy_(dout[i % depth], non_blocking=True)
def both(hin, din):
    def f(n):
        for i in range(n):
            with torch.cuda.stream(s_in): din[i % depth].copy_(hin[i % depth], non_blocking=True)
            with torch.cuda.stream(s_out): hout[i % depth].copy_(dout[i % depth], non_blocking=True)
    return f
def kern(din):
    def f(n):
        for i in range(n): e.forward(din[i % depth], want_q=True, want_f=False, out_q=dout[i % depth], stream=s_k, slot=i % depth, assume_ordered=True)
    return f
def full(hin, din, d2h_stream, same_stream_d2h=False):
    ev_in = [torch.cuda.Event() for _ in range(depth)]; ev_k = [torch.cuda.Event() for _ in range(depth)]; ev_out = [torch.cuda.Event() for _ in range(depth)]
    def f(n):
        for i in range(n):
            bb = i % depth
            with torch.cuda.stream(s_in):
                s_in.wait_event(ev_k[bb]); din[bb].copy_(hin[bb], non_blocking=True); ev_in[bb].record(s_in)
            s_k.wait_event(ev_in[bb]); s_k.wait_event(ev_out[bb])
            e.forward(din[bb], want_q=True, want_f=False, out_q=dout[bb], stream=s_k, slot=bb, assume_ordered=True)
            ev_k[bb].record(s_k)
            with torch.cuda.stream(d2h_stream):
                d2h_stream.wait_event(ev_k[bb]); hout[bb].copy_(dout[bb], non_blocking=True); ev_out[bb].record(d2h_stream)
    return f
def per_slot(hin, din):
    """one stream per frame slot: H2D -> kernels -> D2H in stream order, no events at all; overlap comes from the three slots"""
    ss = [s_in, s_k, s_out]
    def f(n):
        for i in range(n):
            bb = i % depth
            st = ss[bb]
            with torch.cuda.stream(st):
                din[bb].copy_(hin[bb], non_blocking=True)
                e.forward(din[bb], want_q=True, want_f=False, out_q=dout[bb], stream=st, slot=bb, assume_ordered=True)
                hout[bb].copy_(dout[bb], non_blocking=True)
    return f
MB32, MB8, MBO = x.numel() * 4, x.numel(), hout[0].numel()
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
timeit("H2D fp32 24.9 MB alone", h2d(hin32, din32), MB32)
timeit("H2D int8 6.2 MB alone", h2d(hin8, din8), MB8)
timeit("D2H int8 24.9 MB alone", d2h, MBO)
timeit("H2D fp32 + D2H on two streams", both(hin32, din32), MB32 + MBO)
timeit("H2D int8 + D2H on two streams", both(hin8, din8), MB8 + MBO)
timeit("kernels alone (fp32 frames resident)", kern(din32), 0)
timeit("full loop fp32 (bench e2e)", full(hin32, din32, s_out), MB32 + MBO)
timeit("full loop int8 (bench e2e int8)", full(hin8, din8, s_out), MB8 + MBO)
timeit("full loop int8, D2H on the H2D stream", full(hin8, din8, s_in), MB8 + MBO)
timeit("full loop fp32 again", full(hin32, din32, s_out), MB32 + MBO)
timeit("one stream per slot, fp32", per_slot(hin32, din32), MB32 + MBO)
timeit("one stream per slot, int8", per_slot(hin8, din8), MB8 + MBO)
timeit("one stream per slot, fp32 again", per_slot(hin32, din32), MB32 + MBO)
timeit("full loop fp32 (events) after that", full(hin32, din32, s_out), MB32 + MBO)
