#!/bin/bash
# Per-basic-block instruction histogram of one kernel: tools/isa.sh <file stem, e.g. sesrq_mfma> '<demangled substring>' [--blocks] [--ops]
# (hipcc -save-temps into .scratch/isa, then tools/isa_count.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/sesrq_isa
cd /tmp/sesrq_isa
STEM=$1; shift
if [ ! -f $STEM-hip-amdgcn-amd-amdhsa-gfx950.s ] || [ $ROOT/sesr-pytorch-quantize_amd/csrc/$STEM.hip -nt $STEM-hip-amdgcn-amd-amdhsa-gfx950.s ] || [ $ROOT/sesr-pytorch-quantize_amd/csrc/sesrq_mfma_common.h -nt $STEM-hip-amdgcn-amd-amdhsa-gfx950.s ] || [ $ROOT/sesr-pytorch-quantize_amd/csrc/sesrq_common.h -nt $STEM-hip-amdgcn-amd-amdhsa-gfx950.s ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form $(make -s -C $ROOT/sesr-pytorch-quantize_amd/csrc print-fileflags-$STEM) $ISA_FLAGS \
    -I$ROOT/include -I$ROOT/sesr-pytorch-quantize_amd/csrc -save-temps -c $ROOT/sesr-pytorch-quantize_amd/csrc/$STEM.hip -o $STEM.o 2>&1 | grep -E "error|warning" || true
fi
python3 $ROOT/tools/isa_count.py $STEM-hip-amdgcn-amd-amdhsa-gfx950.s "$@"
