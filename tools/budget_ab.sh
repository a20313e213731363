#!/bin/bash
# Per-kernel workgroup budgets (experiment knobs SESRQ_WG_F5 / SESRQ_WG_TRIO / SESRQ_WG_H5) in one GPU session:
#   tools/budget_ab.sh "512 512 512" "256 512 512" ...      (first layer, trio, last layer)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do for c in "$@"; do
  read f t h <<< "$c"
  env SESRQ_WG_F5=$f SESRQ_WG_TRIO=$t SESRQ_WG_H5=$h python bench.py --steps 150 --warmup 30 --repeats 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c', d['value'], d['spread']['max'], [l['ms'] for l in d['roofline']['launches']], d['parity']['mismatches'])"
done; done
