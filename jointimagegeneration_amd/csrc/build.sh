#!/bin/bash
# Builds libguidegen_hip.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-gpu-rdc"
# GG_CLEAN=1: drop every object first, so that the build is a real one (the driver's build() on a tree that carries old .o files)
if [ "${GG_CLEAN:-0}" = "1" ]; then rm -f ./*.o libguidegen_hip.so; fi
OBJS=()
PIDS=()
for f in gg_conv gg_conv_halo gg_conv_halo3 gg_conv_box gg_conv_tiny gg_norm gg_attn gg_sampler gg_f32 gg_ubench; do
  if [ ! -f $f.o ] || [ $f.hip -nt $f.o ] || [ gg_common.h -nt $f.o ] || [ gg_conv.h -nt $f.o ] || [ gg_conv_halo3_asm.inc -nt $f.o ] || [ ../../include/guidegen_hip.h -nt $f.o ]; then
    echo "hipcc $f.hip"
    EXTRA=""
    # attention: MFMA results feed VALU softmax code directly; without this the compiler parks the score tiles in AGPRs and pays
    # ~250 v_accvgpr moves per 256-key tile
    if [ $f = gg_attn ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form"; fi
    rm -f $f.o             # a failed compile must not leave the previous object to be linked
    $HIPCC $FLAGS $EXTRA -c $f.hip -o $f.o &
    PIDS+=($!)
  fi
  OBJS+=($f.o)
done
for pid in "${PIDS[@]:-}"; do
  if [ -n "$pid" ]; then wait "$pid" || { echo "build.sh: a compile failed" >&2; exit 1; }; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o libguidegen_hip.so "${OBJS[@]}"
echo "built $(pwd)/libguidegen_hip.so"
