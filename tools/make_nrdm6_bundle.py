#!/usr/bin/env python3
"""GPU box: calibrate the collapsed nrdm_6 convs (tests/golden/unpinned/nrdm_6.collapsed.npz, step 1:
tools/make_nrdm6_params.py) with this package's own Calibrator (the reference's exe_mode 0 on the device) on the
reference's committed random frame rand_DM_Input_80x960, and write the integer bundle.
    gpurun -- python tools/make_nrdm6_bundle.py        -> gpurun_out/nrdm_6.bundle.npz (copy to tests/golden/unpinned/)
Parity UNPINNED: the reference cannot int-simulate 8 convs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
from sesrq.calibrate import Calibrator  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "unpinned", "nrdm_6.collapsed.npz"), allow_pickle=False)
L = 8
dev = torch.device("cuda:0")
cal = Calibrator([z[f"Wf{k}"] for k in range(L)], [z[f"bf{k}"] for k in range(L)], 1, dev)
x = np.load(os.path.join(ROOT, "tests", "golden", "rand_DM_Input_80x960.npy"))
cal.observe(torch.from_numpy(x).to(dev))
b = cal.bundle(name="nrdm_6 (nrdm_6_G.pth, calibrated on rand_DM_Input_80x960 by sesrq.Calibrator; parity unpinned)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = os.path.join(ROOT, "gpurun_out", "nrdm_6.bundle.npz")
b.save(out)
print("scale", b.scale, "zero", b.zero, "M", [l.M for l in b.layers], "n", [l.n for l in b.layers], "res", b.M_res, b.n_res)
print("wrote", out)
