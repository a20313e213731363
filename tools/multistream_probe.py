import sys, time, os
sys.path.insert(0, "sesr-pytorch-quantize_amd")
import torch, sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
b = Bundle.load("tests/golden/sesr_x2_rand.crop.npz")
dev = torch.device("cuda:0")
for engname, kw in (("trio", dict()), ("per-layer", dict(fuse_hidden=False))):
    e = sesrq.Engine(b, dev, engine=_lib.ENGINE_MFMA, **kw)
    x = torch.rand(1, 3, 1080, 1920, device=dev)
    for NS in (1, 2, 3, 4, 6):
        streams = [torch.cuda.Stream() for _ in range(NS)]
        outs = [torch.empty(e.out_shape(1, 1080, 1920), dtype=torch.int8, device=dev) for _ in range(NS)]
        for i in range(20): e.forward(x, want_f=False, out_q=outs[i % NS], stream=streams[i % NS], slot=i % NS)
        torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        for i in range(K): e.forward(x, want_f=False, out_q=outs[i % NS], stream=streams[i % NS], slot=i % NS)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(engname, "streams", NS, "fps", round(K / dt, 1), "us/frame", round(dt / K * 1e6, 1))
