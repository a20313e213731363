#!/usr/bin/env python3
"""Scan a hipcc .s file for the gfx9 "VMEM store data" hazard: a buffer/global/flat store of more than 64 bits whose data
VGPRs are written again before 2 wait states have passed (an instruction = 1 state, s_nop N = N + 1).  hipcc pads this for the
instructions it models; inline asm and (as found here) the SECOND operand of v_permlane*_swap are written without the pad.
usage: store_hazard_scan.py file.s ['demangled substring']"""
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
    if not op.startswith(("v_", "ds_read", "buffer_load", "global_load")): return set()
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
        states, j, asm = 0, i + 1, False
        while j < len(lines) and states < 2:
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
