import sys, time, os
sys.path.insert(0, "sesr-pytorch-quantize_amd"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, sesrq, numpy as np
from sesrq import _lib
from oracle import sesrq_oracle as O
from helpers import bundle_from_oracle
dev = torch.device("cuda:0")
net = O.synth_net("sesr_x2", 0)
e = sesrq.Engine(bundle_from_oracle(net), dev)
print(e.layer_engines())
x = torch.rand(1, 3, 1080, 1920, device=dev)
print("layer_ms", e.forward_timed(x, 20))
for NS in (1, 3):
    streams = [torch.cuda.Stream() for _ in range(NS)]
    outs = [torch.empty(e.out_shape(1, 1080, 1920), dtype=torch.int8, device=dev) for _ in range(NS)]
    for i in range(20): e.forward(x, want_f=False, out_q=outs[i % NS], stream=streams[i % NS], slot=i % NS)
    torch.cuda.synchronize()
    K = 300; t0 = time.perf_counter()
    for i in range(K): e.forward(x, want_f=False, out_q=outs[i % NS], stream=streams[i % NS], slot=i % NS)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("merged x2 1080p streams", NS, "fps", round(K / dt, 1), "us/frame", round(dt / K * 1e6, 1))
