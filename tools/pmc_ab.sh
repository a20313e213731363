#!/bin/bash
# SQ counters of the bench kernels for several builds of libsesrq.so in one GPU session: tools/pmc_ab.sh <out dir> base prev ...
# (names = directories under sesr-pytorch-quantize_amd/lib/, "base" = the default build); one rocprofv3 --pmc pass per counter group.
O=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-e2e --steps 5 --warmup 2 --repeats 1 --streams 1 --submit step"
for v in "$@"; do
  if [ "$v" = "base" ]; then unset SESRQ_LIB; else export SESRQ_LIB=$PWD/sesr-pytorch-quantize_amd/lib/$v/libsesrq.so; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/${v}_sq1 -- python3 bench.py $B > $O/${v}_sq1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/${v}_sq2 -- python3 bench.py $B > $O/${v}_sq2.log 2>&1
done
python3 - "$O" "$@" <<'P'
import sys, glob, csv, collections, os
O, names = sys.argv[1], sys.argv[2:]
tab = {}
for v in names:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(O, v + "_sq*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sesrq::mfma" in k: agg[k.split("(")[0].replace("void sesrq::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tab[v] = {k: {n: sum(x) / len(x) for n, x in c.items()} for k, c in agg.items()}
kernels = sorted({k for v in tab.values() for k in v})
for k in kernels:
    print("\n" + k)
    cs = sorted({n for v in names for n in tab[v].get(k, {})})
    print("  %-28s" % "counter" + "".join("%14s" % v for v in names))
    for n in cs:
        print("  %-28s" % n + "".join("%14.4g" % tab[v].get(k, {}).get(n, float("nan")) for v in names))
P
