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
@pytest.mark.parametrize("stem", ["sesrq_mfma", "sesrq_trio"])
def test_no_valu_write_behind_a_wide_store(stem, tmp_path):
    asm = str(tmp_path / (stem + ".s"))
    # the library's own flags (one source of truth: the Makefile), device side only, assembly out
    flags = subprocess.run(["make", "-s", "-C", CSRC, "print-cxxflags"], check=True, capture_output=True, text=True).stdout.split()
    assert "-ffp-contract=off" in flags and "-fPIC" in flags
    flags += subprocess.run(["make", "-s", "-C", CSRC, "print-fileflags-" + stem], check=True, capture_output=True, text=True).stdout.split()
    flags += ["-I" + CSRC, "--cuda-device-only", "-S"]
    subprocess.run([HIPCC] + flags + [os.path.join(CSRC, stem + ".hip"), "-o", asm], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_hazard_scan.py"), asm], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    # kernarg_warm() (sesrq_mfma_common.h) touches one dword per 64-byte line of the argument segment with the kernel's first instructions:
    # every such read must lie inside the segment the code object declares (explicit struct + the implicit arguments behind it)
    import re
    text = open(asm).read()
    kernels = 0
    for m in re.finditer(r"^(_ZN5sesrq\w+):[^\n]*\n(.*?)\.amdhsa_kernarg_size (\d+)", text, re.S | re.M):
        body, size = m.group(2), int(m.group(3))
        warm = []
        for line in body.split("\n"):
            t = line.strip()
            if not t or t.startswith((";", ".")):
                continue
            mm = re.match(r"s_load_dword s\d+, s\[0:1\], (0x[0-9a-f]+|\d+)$", t)
            if not mm:
                if re.match(r"s_load_dwordx\d+ s\[\d+:\d+\], s\[0:1\], ", t):
                    continue                              # an argument load of the kernel proper, scheduled into the batch
                break
            warm.append(int(mm.group(1), 0))
        if warm:
            kernels += 1
            warm.sort()                                   # the scheduler may permute the batch
            assert len(warm) >= 4 and warm[0] == 0 and max(warm) + 4 <= size, (m.group(1), warm, size)
            # ... and inside the EXPLICIT argument struct: the implicit arguments begin 256 bytes before the end of the segment
            assert max(warm) + 4 <= size - 256 + 4, (m.group(1), warm, size)
            assert all(b - a <= 64 for a, b in zip(warm, warm[1:])), (m.group(1), warm)     # no line of the segment skipped
    assert kernels >= 5, kernels


def test_scanner_follows_branches_and_sees_asm_defined_operands(tmp_path):
    """The scanner itself, on hand-made assembly: a write of the store's data register reached only through a taken branch, the same
    window padded, and an MFMA reading a VGPR that an inline-asm statement has just defined."""
    def run(body):
        f = tmp_path / "k.s"
        f.write_text("_Z1kv:\n" + body + "\ts_endpgm\n")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "store_hazard_scan.py"), str(f)], capture_output=True, text=True)
        return r.returncode, r.stdout
    store = "\tbuffer_store_dwordx4 v[4:7], v1, s[0:3], 0 offen\n"
    rc, out = run(store + "\ts_cbranch_scc1 .LBB0_2\n\tv_mov_b32_e32 v9, v0\n\ts_nop 0\n.LBB0_2:\n\tv_mov_b32_e32 v5, v0\n")
    assert rc == 1 and "writes [5]" in out, out                       # only the TAKEN path overwrites v5, one wait state behind the store
    rc, out = run(store + "\ts_nop 1\n\ts_cbranch_scc1 .LBB0_2\n\tv_mov_b32_e32 v9, v0\n.LBB0_2:\n\tv_mov_b32_e32 v5, v0\n")
    assert rc == 0, out
    rc, out = run(store + "\tv_mov_b32_e32 v4, v0\n")
    assert rc == 1, out
    rc, out = run("\t;;#ASMSTART\n\tv_mov_b32 v8, s2\n\t;;#ASMEND\n\tv_mfma_i32_16x16x64_i8 v[0:3], v[8:11], v[12:15], v[0:3]\n")
    assert rc == 1 and "inline asm" in out, out
    rc, out = run("\t;;#ASMSTART\n\ts_nop 1\n\tv_mov_b32 v8, s2\n\t;;#ASMEND\n\tv_add_f32_e32 v9, v8, v8\n")
    assert rc == 0, out
    # round 4's bug: the asm v_mov lands in a register the MFMA in flight still writes (its result there is dead)
    mf = "\tv_mfma_i32_16x16x64_i8 v[104:107], v[30:33], v[108:111], v[104:107]\n"
    rc, out = run(mf + "\tv_add3_u32 v58, v58, v89, v61\n\t;;#ASMSTART\n\ts_nop 1\n\tv_mov_b32 v107, s59\n\t;;#ASMEND\n")
    assert rc == 1 and "WAW" in out, out
    rc, out = run(mf + "\ts_nop 15\n\ts_nop 3\n\t;;#ASMSTART\n\tv_mov_b32 v107, s59\n\t;;#ASMEND\n")
    assert rc == 0, out
