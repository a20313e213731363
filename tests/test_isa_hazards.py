"""Build-time check of the generated ISA (no GPU): no VALU write of a wide store's data registers inside the window gfx950 needs.

hipcc 7.2 pads this hazard by LLVM's model, which says a dwordx3/x4 store with an SGPR soffset needs no wait state; the hardware
needs one (tools/store_hazard_probe.hip, measured), and nothing is padded around inline asm.  The kernels therefore keep soffset = 0
on their 16-byte stores (sesrq_mfma_common.h: store_rows4); this test keeps it that way for every kernel of the hot path."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sesr-pytorch-quantize_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("stem", ["sesrq_mfma", "sesrq_trio", "sesrq_quad"])
def test_no_valu_write_behind_a_wide_store(stem, tmp_path):
    asm = str(tmp_path / (stem + ".s"))
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-mfma-vgpr-form",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S"]
    subprocess.run([HIPCC] + flags + [os.path.join(CSRC, stem + ".hip"), "-o", asm], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_hazard_scan.py"), asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
