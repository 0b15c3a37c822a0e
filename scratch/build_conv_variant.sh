#!/bin/bash
# scratch/build_conv_variant.sh NAME "-DFLAG ..." : a tuning build of libcpx_hip.so with cpx_cnn_bf3.hip recompiled under
# extra flags -> scratch/bin/libcpx_hip_NAME.so (select with CPX_LIB=...)
set -e
cd "$(dirname "$0")/../classifier-pipeline_amd/csrc"
make >/dev/null
mkdir -p ../../scratch/bin /tmp/cpxvar
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -I. -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize $2 -x hip -c cpx_cnn_bf3.hip -o /tmp/cpxvar/bf3_$1.o
OBJS=$(ls build/*.o | grep -v cpx_cnn_bf3.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/bin/libcpx_hip_$1.so $OBJS /tmp/cpxvar/bf3_$1.o -lz -lpthread
echo built scratch/bin/libcpx_hip_$1.so
