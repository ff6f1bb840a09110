#!/bin/bash
# builds libprg_exp{1,2,3}.so (conv_split.hip with -DPRG_SPLIT_EXP=n) next to libprg_hip.so; run on the build host
set -e
cd "$(dirname "$0")/../pointreggpt_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1"
OBJS="geometry.o conv.o conv_ws.o conv_c64.o conv_w256.o blocks.o attn_fused.o attn_split.o sampler.o unet.o hostpool.o"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DPRG_SPLIT_EXP=$n -c conv_split.hip -o /tmp/conv_split_exp$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libprg_exp$n.so $OBJS /tmp/conv_split_exp$n.o -lz -lpthread
done
