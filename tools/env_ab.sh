#!/bin/bash
# A/B of an environment knob in one GPU session: tools/env_ab.sh SESRQ_DIRECT 0 1   (interleaved ROUNDS times)
ROUNDS=${ROUNDS:-2}
K=$1; shift
for r in $(seq $ROUNDS); do for v in "$@"; do
  env $K=$v python bench.py --steps 150 --warmup 30 --repeats 3 --no-cpu-baseline --no-e2e $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$K=$v', d['value'], d['spread']['max'], [l['ms'] for l in d['roofline']['launches']], d['parity']['mismatches'])"
done; done
