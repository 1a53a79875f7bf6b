#!/bin/bash
# Build an alternate libptb_hip.so from the current tree into pytorch_toolbelt_amd/lib/alt/<name>/ (travels with gpurun,
# git-ignored) for same-box A/B runs:  PTB_HIP_LIB=pytorch_toolbelt_amd/lib/alt/<name>/libptb_hip.so python bench.py ...
# EXTRA="-DPTB_NT_OUT=0" adds compiler flags (here: plain instead of non-temporal result stores).
set -eu
NAME=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/pytorch_toolbelt_amd/lib/alt/$NAME
mkdir -p $OUT/obj
for f in $ROOT/pytorch_toolbelt_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result ${EXTRA:-} -c $f -o $OUT/obj/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libptb_hip.so
rm -rf $OUT/obj
echo $OUT/libptb_hip.so
