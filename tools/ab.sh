#!/bin/bash
# A/B: run bench with several builds of libsesrq.so in one GPU session (same device).
for v in "$@"; do
  if [ "$v" = "base" ]; then unset SESRQ_LIB; else export SESRQ_LIB=$PWD/sesr-pytorch-quantize_amd/lib/$v/libsesrq.so; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['roofline']['layer_ms'])"
done
