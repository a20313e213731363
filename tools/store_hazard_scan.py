#!/usr/bin/env python3
"""Scan a hipcc .s file for the "VMEM store data" hazard: a buffer/global/flat store of more than 64 bits whose data VGPRs are
written again too early (an instruction = 1 wait state, s_nop N = N + 1).  Measured on gfx950 (tools/store_hazard_probe.hip):
soffset = 0 needs 2 wait states (what hipcc pads), an SGPR soffset needs 1 -- but LLVM's model (and the ISA manual's table) say
an SGPR soffset needs NONE, so hipcc 7.2 may put a VALU write of the data registers directly behind such a store, and pads
nothing at all around inline asm: 0.02 % of the stores then carry the overwritten register (wrong from run to run).
usage: store_hazard_scan.py file.s ['demangled substring']   (exit status 1 if a site is found)"""
import re, subprocess, sys
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
lines = open(src).read().split("\n")
kern = None
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
def written(op, args):
    """VGPRs an instruction writes (first operand; both operands for the swaps)"""
    if not op.startswith("v_"): return set()       # VALU writers only: LDS / memory loads return long after the store has read its data
    w = regs(args[0]) if args else set()
    if op.startswith("v_permlane") and "swap" in op and len(args) > 1: w |= regs(args[1])
    if op.startswith(("v_cmp", "v_readfirstlane", "v_readlane")): return set()
    return w
hits = 0
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    t = l.strip()
    if kern and pat in kern and re.match(r"(buffer|global|flat)_store_dwordx[34]", t):
        data = regs(t.split()[1].rstrip(","))
        toks = [a.rstrip(",") for a in t.split()]
        soff = next((a for a in toks[3:] if re.match(r"(s\d+|m0|0|off)$", a)), "0") if t.startswith("buffer") else "0"
        need = 1 if re.match(r"s\d+$|m0$", soff) else 2
        states, j, asm = 0, i + 1, False
        while j < len(lines) and states < need:
            u = lines[j].strip()
            j += 1
            if u.startswith(";;#ASMSTART"): asm = True; continue
            if u.startswith(";;#ASMEND"): asm = False; continue
            if not u or u.startswith((";", ".")): continue
            op = u.split()[0]
            args = [a.strip().rstrip(",") for a in u.split()[1:]]
            if op == "s_nop":
                states += int(args[0]) + 1; continue
            w = written(op, args)
            if w & data:
                hits += 1
                print(f"{kern}\n  line {i + 1}: {t}\n  line {j}: {u}   <- writes {sorted(w & data)} after {states} wait state(s){' [inline asm]' if asm else ''}")
                break
            states += 1
    i += 1
print(f"{hits} hazard site(s) in {src}" + (f" (kernels matching '{pat}')" if pat else ""))
sys.exit(1 if hits else 0)
