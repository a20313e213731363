#!/usr/bin/env python3
"""Build container only: collapse the reference's nrdm_6 float checkpoint (weights-only load) with THIS package's
closed-form fold and store the 8 collapsed convs as tests/golden/unpinned/nrdm_6.collapsed.npz (data, like the other
*.params.npz).  The reference has no integer path at this depth: everything derived from this file is
"parity unpinned".  Step 2 (tools/make_nrdm6_bundle.py) calibrates it on the GPU box."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sesr-pytorch-quantize_amd"))
import sim  # noqa: E402

CKPT = "/root/reference/model_params/nrdm_6_G.pth"
m = sim.float_model(4, ckpt=CKPT)          # strict load + collapse
convs = [m.conv_first.conv_expand] + [b.conv_expand for b in m.residual_block] + [m.conv_last.conv_expand]
arrs = {}
for k, c in enumerate(convs):
    arrs[f"Wf{k}"] = c.weight.detach().numpy().astype(np.float32)
    arrs[f"bf{k}"] = c.bias.detach().numpy().astype(np.float32)
meta = dict(case="nrdm_6", mflag=4, ckpt=os.path.basename(CKPT), pixel_shuffle=1,
            note="collapsed by sesr-pytorch-quantize_amd/models/model_utils_pt.py; parity unpinned (no reference integer path for 8 convs)")
out = os.path.join(ROOT, "tests", "golden", "unpinned", "nrdm_6.collapsed.npz")
np.savez_compressed(out, meta=np.array(json.dumps(meta)), **arrs)
print("wrote", out, [a.shape for k, a in arrs.items() if k.startswith("Wf")])
