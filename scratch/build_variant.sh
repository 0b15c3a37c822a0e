#!/bin/bash
# scratch/build_variant.sh NAME "-DFOO=1 -DBAR=2"  -> scratch/bin/libcpx_hip_NAME.so (experiment builds; never shipped)
set -e
cd "$(dirname "$0")/../classifier-pipeline_amd/csrc"
mkdir -p ../../scratch/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I. -Wall -Wno-unused-function -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form $2 \
  -x hip -shared -o ../../scratch/bin/libcpx_hip_$1.so cpx_api.cpp cpx_host.cpp cpx_track.hip cpx_assoc.hip cpx_classify.hip cpx_cnn.hip cpx_cptv.hip cpx_thumb.hip cpx_ir.hip cpx_cnn_bf3.hip cpx_cnn_rw.hip cpx_mog2.hip cpx_inflate.hip -lz -lpthread
