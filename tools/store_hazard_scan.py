#!/usr/bin/env python3
"""Scan a hipcc .s file for the "VMEM store data" hazard: a buffer/global/flat store of more than 64 bits whose data VGPRs are
written again too early (an instruction = 1 wait state, s_nop N = N + 1).  Measured on gfx950 (tools/store_hazard_probe.hip):
soffset = 0 needs 2 wait states (what hipcc pads), an SGPR soffset needs 1 -- but LLVM's model (and the ISA manual's table) say
an SGPR soffset needs NONE, so hipcc 7.2 may put a VALU write of the data registers directly behind such a store, and pads
nothing at all around inline asm: 0.02 % of the stores then carry the overwritten register (wrong from run to run).
A second rule covers what hipcc cannot see at all: a VGPR written INSIDE an inline-asm statement (;;#ASMSTART .. ;;#ASMEND) and read by an
MFMA or a vector-memory instruction within the next few instructions -- hipcc pads hazards only for instructions it models, so such a
consumer would need the padding inside the asm string (today every asm-defined VGPR of the kernels feeds VALU / LDS instructions only).
A third rule: an inline-asm VALU write of a VGPR inside the destination of an MFMA issued at most 18 wait states earlier (the MFMA writes
its result late: a register whose MFMA result is dead can be handed out again at once; round 4's in_vgpr() bug, sesrq_mfma_common.h).
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
# ---- rule 2: inline-asm VGPR definitions feeding an MFMA or VMEM instruction right behind the statement
# ---- rule 3: an inline-asm VALU write of a VGPR that an MFMA issued shortly before still has to write (WAW: hipcc pads it only for
# instructions it models; it happens when the MFMA's result in that register is dead, so the allocator hands the register out again)
kern2, in_asm, fresh = None, False, {}        # fresh: register -> instructions left in which a modelled consumer would be a hazard
pending = {}                                  # register -> wait states left until a recent MFMA has surely written it
for n, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern2 = m.group(1); fresh = {}; pending = {}
    u0 = l.strip()
    if u0 and not u0.startswith((";", ".")) and not re.match(r"^[.\w]+:", u0):
        op0 = u0.split()[0]; args0 = [a.strip().rstrip(",") for a in u0.split()[1:]]
        if in_asm and op0.startswith("v_") and args0:
            bad = regs(args0[0]) & set(pending)
            if bad and (not pat or pat in subprocess.run(["c++filt", kern2 or ""], capture_output=True, text=True).stdout):
                hits += 1
                print(f"{kern2}\n  line {n + 1}: {u0}   <- inline asm writes {sorted(bad)} while an MFMA issued <= 18 wait states earlier still writes it (unmodelled WAW)")
        if not in_asm and pending and len(args0) > 1:
            # a MODELLED instruction that reads such a register was padded by hipcc until the MFMA's result had landed: the register is
            # settled from here on (not so an MFMA that accumulates in place, vDst == SrcC: that dependency is forwarded, not waited for)
            mf = op0.startswith(("v_mfma", "v_smfmac"))
            for k, a in enumerate(args0[1:], 1):
                if mf and k == 3 and regs(a) == regs(args0[0]): continue
                for r in regs(a): pending.pop(r, None)
        step = int(args0[0]) + 1 if op0 == "s_nop" and args0 else 1
        pending = {r: c - step for r, c in pending.items() if c > step}
        if op0.startswith(("v_mfma", "v_smfmac")) and args0:
            for r in regs(args0[0]): pending[r] = 18
    u = l.strip()
    if u.startswith(";;#ASMSTART"): in_asm = True; continue
    if u.startswith(";;#ASMEND"): in_asm = False; continue
    if not u or u.startswith((";", ".")) or re.match(r"^[.\w]+:", u): continue
    op = u.split()[0]
    args = [a.strip().rstrip(",") for a in u.split()[1:]]
    if in_asm:
        if op.startswith("v_") and args:
            for r in regs(args[0]): fresh[r] = 4
        continue
    if fresh:
        if op.startswith(("v_mfma", "v_smfmac", "buffer_", "global_", "flat_")):
            used = set()
            for a in args[(1 if op.startswith(("v_mfma", "v_smfmac")) else 0):]: used |= regs(a)
            bad = used & set(fresh)
            if bad and (not pat or pat in subprocess.run(["c++filt", kern2 or ""], capture_output=True, text=True).stdout):
                hits += 1
                print(f"{kern2}\n  line {n + 1}: {u}   <- reads {sorted(bad)} defined by inline asm {4 - min(fresh[r] for r in bad)} instruction(s) earlier (unmodelled hazard)")
        if op == "s_nop": fresh = {}
        else: fresh = {r: c - 1 for r, c in fresh.items() if c > 1}
# ---- rule 1: VALU write of a wide store's data registers inside the window, FOLLOWING control flow (both successors of a conditional
# branch, the target of an unconditional one; a branch instruction is itself one wait state)
labels = {}
for n, l in enumerate(lines):
    m = re.match(r"^(\.L\w+):", l)
    if m: labels[m.group(1)] = n
def scan(j, states, need, data, seen):
    """-> (line number, text, states, in inline asm) of the first offending write on any path, else None"""
    asm = False
    while j < len(lines) and states < need:
        u = lines[j].strip()
        j += 1
        if u.startswith(";;#ASMSTART"): asm = True; continue
        if u.startswith(";;#ASMEND"): asm = False; continue
        if not u or u.startswith((";", ".")) or re.match(r"^[.\w]+:", u): continue
        op = u.split()[0]
        args = [a.strip().rstrip(",") for a in u.split()[1:]]
        if op == "s_endpgm": return None
        if op.startswith(("s_setpc", "s_swappc")): return (j, u + "   (indirect jump: successor unknown)", states, asm)
        if op.startswith(("s_branch", "s_cbranch")):
            tgt = labels.get(args[0]) if args else None
            if tgt is None: return (j, u + "   (branch target not found)", states, asm)
            if (tgt, states) not in seen:
                seen.add((tgt, states))
                r = scan(tgt + 1, states + 1, need, data, seen)
                if r: return r
            if op == "s_branch": return None
            states += 1; continue
        if op == "s_nop":
            states += int(args[0]) + 1; continue
        w = written(op, args)
        if w & data: return (j, u + f"   <- writes {sorted(w & data)}", states, asm)
        states += 1
    return None
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    t = l.strip()
    if kern and pat in kern and re.match(r"(buffer|global|flat)_store_dwordx[34]", t):
        data = regs(t.split()[1].rstrip(","))
        toks = [a.rstrip(",") for a in t.split()]
        soff = next((a for a in toks[3:] if re.match(r"(s\d+|m0|0|off)$", a)), "0") if t.startswith("buffer") else "0"
        need = 1 if re.match(r"s\d+$|m0$", soff) else 2
        r = scan(i + 1, 0, need, data, set())
        if r:
            hits += 1
            print(f"{kern}\n  line {i + 1}: {t}\n  line {r[0]}: {r[1]} after {r[2]} wait state(s){' [inline asm]' if r[3] else ''}")
print(f"{hits} hazard site(s) in {src}" + (f" (kernels matching '{pat}')" if pat else ""))
sys.exit(1 if hits else 0)
