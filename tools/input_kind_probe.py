import os, sys, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
import sesrq
from sesrq import _lib
from sesrq.bundle import Bundle
dev = torch.device("cuda:0")
b = Bundle.load("tests/golden/sesr_x2_rand.crop.npz")
e = sesrq.Engine(b, dev, engine=_lib.ENGINE_MFMA, wg_budget=512)
g = torch.Generator().manual_seed(1)
H, W = 1080, 1920
yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
smooth = torch.stack([yy, xx, (yy + xx) / 2]).unsqueeze(0).contiguous()
inputs = {"random": torch.rand((1, 3, H, W), generator=g), "constant 0.5": torch.full((1, 3, H, W), 0.5), "smooth ramp": smooth,
          "ramp + 2 % noise": (smooth + 0.02 * torch.randn((1, 3, H, W), generator=g)).clamp(0, 1)}
for _ in range(2):
    for name, x in inputs.items():
        x = x.to(dev)
        for _ in range(20): e.forward(x, want_f=False)
        torch.cuda.synchronize()
        r = e.forward_timed(x, iters=100)
        print(f"{name:18s}", [round(v * 1e3, 2) for v in r[0]], flush=True)
