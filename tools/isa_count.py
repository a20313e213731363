#!/usr/bin/env python3
"""Count instructions per basic block of one kernel in a hipcc -save-temps .s file.
usage: isa_count.py file.s 'demangled-substring' [--blocks]"""
import re, subprocess, sys, collections
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
names = {}
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        names[i] = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
start = [i for i, n in names.items() if pat in n]
if not start:
    sys.exit("no kernel matches; have:\n" + "\n".join(names.values()))
s = start[0]
e = next(i for i in range(s, len(lines)) if ".amdhsa_kernel" in lines[i])
print(names[s])
def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith("v_"): return "VALU"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "WAIT"
    if op.startswith("s_barrier"): return "BAR"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "BR"
    if op.startswith("s_"): return "SALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
    return "OTHER"
blk = "entry"; order = [blk]; cnt = collections.defaultdict(collections.Counter); ops = collections.defaultdict(collections.Counter)
for l in lines[s + 1:e]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blk = m.group(1); order.append(blk); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."): continue
    op = t.split()[0]
    cnt[blk][cls(op)] += 1
    ops[blk][op] += 1
tot = collections.Counter()
for b in order:
    tot.update(cnt[b])
    if "--blocks" in sys.argv and sum(cnt[b].values()) > 12:
        print(f"{b:12s}", dict(cnt[b]))
        if "--ops" in sys.argv:
            print("     ", dict(ops[b].most_common(14)))
print("TOTAL", dict(tot))
for l in lines[e:e + 40]:
    if "next_free_vgpr" in l or "group_segment_fixed_size" in l or "next_free_sgpr" in l: print(l.strip())
