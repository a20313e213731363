#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...>   -- one rocprofv3 PMC pass over a short bench run (kernels run alone: 1 stream)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/$out -- python bench.py --steps 5 --warmup 2 --repeats 1 --streams 1 --no-cpu-baseline --no-e2e $BENCH_ARGS > gpurun_out/$out.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/$out/**/*counter_collection.csv",recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    if "sesrq" not in k or "verify" in k: continue
    print(k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
