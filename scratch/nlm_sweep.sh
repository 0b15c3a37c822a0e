#!/bin/bash
# runs scratch/nlm_probe.py with the shipped library and every library under scratch/bin (on the GPU box)
cd "$(dirname "$0")/.."
python scratch/nlm_probe.py "$@" 2>&1 | grep "us per"
for v in scratch/bin/libcpx_hip_*.so; do
  [ -f "$v" ] || continue
  CPX_LIB=$PWD/$v python scratch/nlm_probe.py "$@" 2>&1 | grep "us per"
done
