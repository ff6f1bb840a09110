#!/bin/bash
# tools/variant_lib.sh <name> <source.hip> "<extra hipcc flags>"  — builds pointreggpt_amd/libprg_<name>.so = the product library with ONE
# source recompiled with extra flags (timing experiments: -DPRG_C64_EXP=512, -DPRG_C64W_EXP=6, ...).  Run `make` in csrc first; select
# the library with PRG_HIP_LIB=... (tools/gpu_c64_exp.sh, tools/gpu_c64w_exp.sh expect libprg_c64exp{1,512}.so / libprg_c64wexp{1,2,4,6}.so).
set -e
cd "$(dirname "$0")/../pointreggpt_amd/csrc"
NAME=$1; SRC=$2; shift 2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function"
[ "$SRC" != conv_split512.hip ] && FLAGS="$FLAGS -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS=""
for f in geometry conv conv_split conv_split512 conv_ws conv_c64 conv_c64w conv_w256 blocks attn_fused attn_split sampler unet hostpool; do
  [ "$f.hip" = "$SRC" ] && OBJS="$OBJS /tmp/variant_$NAME.o" || OBJS="$OBJS $f.o"
done
/opt/rocm/bin/hipcc $FLAGS "$@" -c $SRC -o /tmp/variant_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libprg_$NAME.so $OBJS -lz -lpthread
echo built ../libprg_$NAME.so
