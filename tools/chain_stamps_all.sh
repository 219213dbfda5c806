#!/bin/bash
# GPU box: phase stamps of the chain kernels' layers (forward / backward, 2 x 6^3 x 128 and 2 x 3^3 x 256) on a -DVS_CHAIN_STAMPS build of the current sources
# (tools/_dbg/ does not travel: the diagnostic library is built on the box).  -> stdout; profiles/r06_chain_phase_stamps_final.txt is its output.
#   gpurun --timeout 900 -- 'bash tools/chain_stamps_all.sh > gpurun_out/chain_stamps.txt 2>&1'
set -e
cd "$(dirname "$0")/.."
VS_STAMPS_DEF="VS_CHAIN_STAMPS" VS_STAMPS_OUT=libvaeseg_chainstamps.so bash tools/build_stamps.sh > /dev/null 2>&1
for args in "2 128 128 6" "2 128 128 6 bwd" "2 256 256 3" "2 256 256 3 bwd"; do python tools/chain_stamps.py $args 2>&1 | grep -v amdgpu.ids; done
