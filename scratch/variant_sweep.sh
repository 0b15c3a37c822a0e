#!/bin/bash
# runs scratch/cnn_probe.py with every library under scratch/bin (on the GPU box); CPX_LIB selects the library
cd "$(dirname "$0")/.."
for v in scratch/bin/libcpx_hip_*.so; do
  echo "=== $v"
  CPX_LIB=$PWD/$v python scratch/cnn_probe.py ${1:-2048} 2>&1 | grep -v amdgpu.ids | head -${2:-6}
done
