#!/bin/bash
# Same-box A/B of the launch plans: tools/fuse_ab.sh [rounds]  (prints fps and per-launch ms for --fuse 1 and --fuse 2)
R=${1:-2}
for r in $(seq $R); do for f in 1 2; do
  timeout -k 10 120 python bench.py --fuse $f --no-e2e --no-cpu-baseline 2>/dev/null | tail -1 | python -c '
import sys, json
d = json.loads(sys.stdin.read()); print("fuse", sys.argv[1], d["value"], [round(l["ms"], 4) for l in d["roofline"]["launches"]])' $f
done; done
