#!/bin/bash
# same-box A/B of the conv_rw_kernel launches (stage 3 and the strided first convolutions): shipped library against every
# experiment build under scratch/bin; two interleaved rounds
cd "$(dirname "$0")/.."
for r in 1 2; do
  echo "round $r"
  echo "shipped"; python scratch/cnn_probe.py ${1:-1536} 2>&1 | grep -E "64 cout_g  64|stride [23]"
  for v in scratch/bin/libcpx_hip_*.so; do
    [ -f "$v" ] || continue
    echo $(basename $v .so | sed 's/libcpx_hip_//'); CPX_LIB=$PWD/$v python scratch/cnn_probe.py ${1:-1536} 2>&1 | grep -E "64 cout_g  64|stride [23]"
  done
done
