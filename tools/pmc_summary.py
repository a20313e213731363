#!/usr/bin/env python3
"""Merge the rocprofv3 PMC passes of tools/profile_round.sh into <dir>/pmc_summary.json (bench.py reads the committed copy,
profiles/pmc_summary.json) and a markdown table <dir>/sq_counters.md.  Per-launch averages over the bench kernels.
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes): FETCH_SIZE reports half of the bytes of coalesced streaming
reads on gfx950 (4 and 16 B/lane calibrated), WRITE_SIZE is exact for 16-B/lane stores (/opt/skills/guides/MI355X_MICROARCH.md, HBM); narrower stores
(the 2-byte PixelShuffle runs of the last layer) are uncalibrated and flagged."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
N_SIMD, N_SE = 1024, 32
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sesrq" not in k or "verify" in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
order = ["mfma_f5", "mfma_trio", "mfma_h3", "mfma_h5"]


def launch_key(k):
    for i, o in enumerate(order):
        if o in k:
            return i
    return 99


kernels = sorted(agg, key=launch_key)
rows, summary = [], {}
for li, k in enumerate(kernels):
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    cyc = c.get("SQ_BUSY_CYCLES", 0) / N_SE                      # kernel length in cycles (SQ busy, per SE)
    ent = {"kernel": k.split("(")[0]}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # FETCH_SIZE reports half of the bytes of coalesced streaming reads on gfx950 (MI355X_MICROARCH.md: 16 B/lane); round 4 calibrated the
        # first layer's width too (tools/fetch_calib_probe.hip: 1 GiB read once with 4 B per lane -> FETCH_SIZE 0.5 GiB, like 16 B per lane).
        # Rounds 1-3 took the first layer's counter as reported ("x1, uncalibrated") and under-stated its traffic by its input bytes.
        ent["fetch_bytes"] = 2 * c["FETCH_SIZE"] * 1024
        ent["fetch_correction"] = "x2 (calibrated for 4 B/lane and 16 B/lane loads: tools/fetch_calib_probe.hip)"
        ent["write_bytes"] = c["WRITE_SIZE"] * 1024
        ent["hbm_bytes_per_launch"] = round(ent["fetch_bytes"] + ent["write_bytes"])
    if cyc:
        ent["kernel_cycles"] = round(cyc)
        ent["valu_util"] = round(c.get("SQ_ACTIVE_INST_VALU", 0) * 4 / N_SIMD / cyc, 3)
        ent["mfma_util"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / N_SIMD / cyc, 3)
        ent["lds_util"] = round(c.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc, 3)
        # cycles of instruction activity per SIMD (all of its waves, overlap counted twice): with the clock the chip holds under load
        # (1.75-1.95 GHz in-kernel, tools/stamps.py; NOT 2.4) this is the kernel's instruction-bound time
        ent["inst_active_cycles_per_simd"] = round(c.get("SQ_ACTIVE_INST_ANY", 0) * 4 / N_SIMD)
        if c.get("SQ_WAVE_CYCLES"):
            ent["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)      # issue stalls / wave-cycles
            ent["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)            # s_waitcnt + barriers
        if c.get("SQ_WAVES"):
            ent["wave_lifetime_cycles"] = round(c.get("SQ_WAVE_CYCLES", 0) * 4 / c["SQ_WAVES"])
    if "GRBM_GUI_ACTIVE" in c:
        ent["grbm_gui_active"] = round(c["GRBM_GUI_ACTIVE"])
    ent["counters"] = {n: round(v) for n, v in sorted(c.items())}
    summary[k] = ent
    rows.append((k, c))
json.dump(summary, open(os.path.join(root, "pmc_raw.json"), "w"), indent=1)
# the committed form bench.py reads (profiles/pmc_summary.json): keyed by workload, then "launch{j}:{engine name}" of the default
# launch plan (first layer / trio / last layer of the profiled SESR-x2 bundle)
ENGINE_NAMES = {"mfma_f5": "mfma-f5-hybrid", "mfma_trio": "mfma-trio-merged", "mfma_h5": "mfma-h5-general"}
committed = {}
for li, k in enumerate(kernels):
    tag = next((v for o, v in ENGINE_NAMES.items() if o in k), None)
    if tag is None:
        continue
    committed[f"launch{li}:{tag}"] = {x: v for x, v in summary[k].items() if x != "counters"}
json.dump({"sesr_x2_1080p": committed,
           "_source": "tools/profile_round.sh (rocprofv3 --kernel-trace --pmc, one pass per counter group, --streams 1), tools/pmc_summary.py"},
          open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)
names = sorted({n for _, c in rows for n in c})
with open(os.path.join(root, "sq_counters.md"), "w") as f:
    f.write("| counter (per launch) | " + " | ".join(k.split("(")[0].replace("void sesrq::", "") for k, _ in rows) + " |\n")
    f.write("|---|" + "---|" * len(rows) + "\n")
    for n in names:
        f.write(f"| {n} | " + " | ".join(f"{c.get(n, float('nan')):.4g}" for _, c in rows) + " |\n")
    for key in ("kernel_cycles", "valu_util", "mfma_util", "lds_util", "wait_inst_frac", "wait_any_frac", "wave_lifetime_cycles", "hbm_bytes_per_launch"):
        f.write(f"| **{key}** | " + " | ".join(str(summary[k].get(key, "")) for k, _ in rows) + " |\n")
for k in kernels:
    e = summary[k]
    print(e["kernel"], {x: e.get(x) for x in ("hbm_bytes_per_launch", "kernel_cycles", "valu_util", "mfma_util", "lds_util", "wave_lifetime_cycles")})
