#!/bin/bash
# Builds a VARIANT of libguidegen_hip.so for same-box A/B runs and diagnostic builds:
#   tools/experiments/build_variant.sh NAME "-DFLAG ..." file1 [file2 ...]   ->  tools/experiments/ab/libNAME.so
# The named .hip files are recompiled with the extra flags, every other object is taken from the product build in csrc/.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
SRC="$ROOT/jointimagegeneration_amd/csrc"
OUT="$ROOT/tools/experiments/ab"
NAME=$1; FLAGS=$2; shift 2
mkdir -p "$OUT/$NAME"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
BASE="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -fno-gpu-rdc"
OBJS=()
for f in gg_conv gg_conv_halo gg_conv_halo3 gg_conv_box gg_conv_tiny gg_norm gg_attn gg_sampler gg_f32 gg_ubench; do
  if [[ " $* " == *" $f "* ]]; then
    EXTRA=""; if [ $f = gg_attn ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
    (cd "$SRC" && rm -f "$OUT/$NAME/$f.o" && $HIPCC $BASE $EXTRA $FLAGS -c $f.hip -o "$OUT/$NAME/$f.o" 2>"$OUT/$NAME/$f.err" || { grep -A5 error "$OUT/$NAME/$f.err"; }) &
    OBJS+=("$OUT/$NAME/$f.o")
  else
    OBJS+=("$SRC/$f.o")
  fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib$NAME.so" "${OBJS[@]}"
echo "built $OUT/lib$NAME.so"
