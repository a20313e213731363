"""Soak: the bench plan (3 streams x 512 slots, sesrq_forward_many) over many frames with EVERY output frame compared, not only frame 0.
A hazard that corrupts one store in 10^5 (round 3's store-data hazard was 2 in 10^4) escapes a single whole-frame parity check; here
ROUNDS x 24 distinct 4K frames are compared byte for byte on the device with the frames a one-stream forward produced (frame 0 of which
is checked against the C oracle).  usage: python tools/soak.py [rounds]   (default 40 rounds = 960 frames of 24.9 MB)
SOAK_OUT = q (int8 frames, default) | f (round 5: the fp32-only store flavours) | fa (fp32 + the x2 anchor add)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
sys.path.insert(0, ROOT)
import sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
# SOAK_BUNDLE / SOAK_SHAPE / SOAK_GROUP: another net, frame size, frames per launch sequence (e.g. nrdm_3.crop.npz 3x540x960 8)
b = Bundle.load(os.path.join(ROOT, "tests/golden", os.environ.get("SOAK_BUNDLE", "sesr_x2_rand.crop.npz")))
OUT = os.environ.get("SOAK_OUT", "q")
e = sesrq.Engine(b, dev, engine=_lib.ENGINE_MFMA, anchor_add=(OUT == "fa"),
                 wg_budget=2 * torch.cuda.get_device_properties(dev).multi_processor_count)      # two slots per CU, as bench.py scales its plan
S, F, G = 3, 24, int(os.environ.get("SOAK_GROUP", "1"))
shape = tuple(int(v) for v in os.environ.get("SOAK_SHAPE", "3x1080x1920").split("x"))
g = torch.Generator(device="cpu").manual_seed(7)
xs = [torch.rand((1,) + shape, generator=g).to(dev) for _ in range(F)]
want = [e.forward(x, want_f=False)[0].clone() for x in xs] if OUT == "q" else [e.forward(x, want_q=False)[1].clone() for x in xs]
torch.cuda.synchronize()
from oracle import sesrq_oracle as O, c_oracle as CO          # checker only
on = O.Net(layers=[O.Layer(wq=l.wq, add_const=l.add_const, M=l.M, n=l.n, relu=l.relu) for l in b.layers], scale=b.scale, zero=b.zero,
           M_res=b.M_res, n_res=b.n_res, pixel_shuffle=b.pixel_shuffle, pe=b.pe_num, acc_bits=b.pe_acc_bits, add_bits=b.pe_add_bits, name=b.name)
ref = CO.forward(on, xs[0].cpu().numpy(), threads=min(os.cpu_count() or 1, 16), want_f=False)["q_out"]
if OUT != "q":      # the fp32 frame is one rounding of the int8 one (+ the nearest-upsampled input, one fp32 add)
    ref = ((ref.astype(np.float32) - np.float32(b.zero[b.L])) * np.float32(b.scale[b.L])).astype(np.float32)
    if OUT == "fa":
        r_ = b.pixel_shuffle
        ref = (ref + np.repeat(np.repeat(xs[0].cpu().numpy(), r_, axis=2), r_, axis=3)).astype(np.float32)
assert np.array_equal(ref, want[0].cpu().numpy()), "frame 0 differs from the C oracle"
print("frame 0 == C oracle", flush=True)
streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
outs = [torch.zeros_like(want[0]) for _ in range(F)]
sub = e.submission(xs, outs, streams, group=G) if OUT == "q" else e.submission(xs, None, streams, outs_f=outs, group=G)
bad = 0
t0 = time.time()
for r in range(rounds):
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    sub.enqueue(F, first=(5 * r) % F)
    torch.cuda.synchronize()
    for k in range(F):
        if not torch.equal(outs[k], want[k]):
            n = int((outs[k] != want[k]).sum())
            bad += 1
            print(f"round {r} frame {k}: {n} bytes differ", flush=True)
    if r % 10 == 9:
        print(f"round {r + 1}: {(r + 1) * F} frames compared, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("SOAK", "FAILED" if bad else "ok", rounds * F, "frames", b.name, shape, "group", G, "out", OUT)
sys.exit(1 if bad else 0)
