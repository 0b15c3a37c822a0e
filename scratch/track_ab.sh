#!/bin/bash
# the per-clip walk (default) against one launch per frame step (CPX_TRACK_PER_STEP=1), shipped library and every
# experiment build under scratch/bin, on the GPU box
cd "$(dirname "$0")/.."
python scratch/track_probe.py "$@" 2>&1 | grep "us per"
CPX_TRACK_PER_STEP=1 python scratch/track_probe.py "$@" 2>&1 | grep "us per"
for v in scratch/bin/libcpx_hip_*.so; do
  [ -f "$v" ] || continue
  CPX_LIB=$PWD/$v python scratch/track_probe.py "$@" 2>&1 | grep "us per"
  CPX_LIB=$PWD/$v CPX_TRACK_PER_STEP=1 python scratch/track_probe.py "$@" 2>&1 | grep "us per"
done
