#!/bin/bash
# Build a variant of libsesrq.so for same-box A/Bs (tools/ab.sh): tools/build_variant.sh <name> "<extra hipcc flags>" [files...]
#   -> sesr-pytorch-quantize_amd/lib/<name>/libsesrq.so (objects of files not listed are taken from the default build)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=$2; shift 2
SRC=$ROOT/sesr-pytorch-quantize_amd/csrc; LIB=$ROOT/sesr-pytorch-quantize_amd/lib; OUT=$LIB/$NAME
FILES=${@:-sesrq_mfma sesrq_trio}
mkdir -p $OUT
FLAGS=$(make -s -C $SRC print-cxxflags)      # the library's own flags (a hand copy drifts)
pids=()
for f in $FILES; do /opt/rocm/bin/hipcc $FLAGS $(make -s -C $SRC print-fileflags-$f) $EXTRA -c $SRC/$f.hip -o $OUT/$f.o & pids+=($!); done      # the library's per-file flags first: EXTRA can override them
for p in "${pids[@]}"; do wait $p; done
OBJS=""
for s in $(make -s -C $SRC print-srcs); do
  f=${s%.hip}
  if [ -f $OUT/$f.o ]; then OBJS="$OBJS $OUT/$f.o"; else OBJS="$OBJS $LIB/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-soname,libsesrq.so -o $OUT/libsesrq.so $OBJS
echo "built $OUT/libsesrq.so"
