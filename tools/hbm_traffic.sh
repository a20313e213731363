#!/bin/bash
# HBM traffic of every kernel of the bench workload: two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE
# cannot share a pass: TCC slots), as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Writes
# gpurun_out/hbm_traffic_raw.json; copy the summary into profiles/hbm_traffic.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/hbm_$c -- python bench.py --steps 6 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/hbm_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/hbm_{c}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "sesrq" in r["Kernel_Name"] and "verify" not in r["Kernel_Name"]:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        res[k][c] = sum(v) / len(v)
print(json.dumps(res, indent=1))
json.dump(res, open("gpurun_out/hbm_traffic_raw.json", "w"), indent=1)
PY
