#!/bin/bash
# tools/split_variant.sh <name> "<extra hipcc flags>"  — builds pointreggpt_amd/libprg_<name>.so = the product library with conv_split.hip
# compiled with the extra flags (e.g. -DPRG_SPLIT_ABLATE=21, -DPRG_SPLIT_ILV=1).  Run on the build host; select with PRG_HIP_LIB=...
set -e
cd "$(dirname "$0")/../pointreggpt_amd/csrc"
NAME=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS="geometry.o conv.o conv_ws.o conv_c64.o conv_w256.o blocks.o attn_fused.o attn_split.o sampler.o unet.o hostpool.o"
/opt/rocm/bin/hipcc $FLAGS "$@" -c conv_split.hip -o /tmp/conv_split_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libprg_$NAME.so $OBJS /tmp/conv_split_$NAME.o -lz -lpthread
echo built ../libprg_$NAME.so
