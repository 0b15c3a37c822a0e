#!/bin/bash
# runs scratch/cnn_probe.py with every library under cpx/variants (on the GPU box)
cd "$(dirname "$0")/.."
cp classifier-pipeline_amd/cpx/libcpx_hip.so /tmp/libcpx_hip_orig.so
for v in classifier-pipeline_amd/cpx/variants/libcpx_hip_*.so; do
  echo "=== $v"
  cp "$v" classifier-pipeline_amd/cpx/libcpx_hip.so
  python scratch/cnn_probe.py ${1:-2048} 2>&1 | grep -v amdgpu.ids | head -6
done
cp /tmp/libcpx_hip_orig.so classifier-pipeline_amd/cpx/libcpx_hip.so
