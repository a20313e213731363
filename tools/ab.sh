#!/bin/bash
# A/B: run bench with several builds of libsesrq.so in one GPU session (same device), interleaved ROUNDS times.
#   tools/ab.sh base pipe serial     (names = directories under sesr-pytorch-quantize_amd/lib/, "base" = the default build)
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset SESRQ_LIB; else export SESRQ_LIB=$PWD/sesr-pytorch-quantize_amd/lib/$v/libsesrq.so; fi
  python bench.py --steps 150 --warmup 30 --repeats 3 --no-cpu-baseline --no-e2e $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['spread']['max'], [l['ms'] for l in d['roofline']['launches']], d['parity']['mismatches'])"
done; done
