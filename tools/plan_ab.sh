#!/bin/bash
# Same-box A/B of launch plans / stream counts / workgroup budgets: tools/plan_ab.sh [rounds]
R=${1:-1}
CFG=("--fuse 1 --streams 2" "--fuse 1 --streams 3" "--fuse 1 --streams 3 --wg-budget 640" "--fuse 1 --streams 3 --wg-budget 512" "--fuse 1 --streams 3 --wg-budget 768" "--fuse 1 --streams 2 --wg-budget 640")
if [ -n "$PLAN_CFGS" ]; then IFS=";" read -ra CFG <<< "$PLAN_CFGS"; fi
for r in $(seq $R); do for c in "${CFG[@]}"; do
  timeout -k 10 120 python bench.py $c --steps 150 --warmup 30 --repeats 4 --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read()); print("%-48s" % sys.argv[1], d["value"], d["spread"]["max"], [round(l["ms"], 4) for l in d["roofline"]["launches"]], d["parity"]["mismatches"])' "$c"
done; done
