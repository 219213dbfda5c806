#!/bin/bash
# Diagnostic build of libvaeseg with -DVS_STAMPS (s_memtime phase stamps in k3b_kernel); never shipped.
set -e
cd "$(dirname "$0")/../vae_segmentation_amd/csrc"
mkdir -p ../../tools/_dbg/obj
for f in igemm_k3_f32 igemm_k3x igemm_k3_bf16 igemm_k3_f16 igemm_k2s2 igemm_pw igemm_g1_f16 conv_api wgrad pack norm misc gs data up_compose k2s2_scatter8 chain_api config; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -D${VS_STAMPS_DEF:-VS_STAMPS} -Wno-unused-variable -c $f.hip -o ../../tools/_dbg/obj/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_dbg/${VS_STAMPS_OUT:-libvaeseg_stamps.so} ../../tools/_dbg/obj/*.o
