#!/bin/bash
# same-box A/B of the stage-2 block kernels (CPX_BLOCK32_SPLIT: 0 = conv_block32_kernel, 1 = conv_block32s_kernel, 2 = conv_block32p_kernel)
# and of every experiment build under scratch/bin; two rounds, interleaved
cd "$(dirname "$0")/.."
for r in 1 2; do
  echo "round $r"
  for f in 0 1 2; do
    echo -n "SPLIT=$f           "; CPX_BLOCK32_SPLIT=$f python scratch/cnn_probe.py ${1:-1536} 2>&1 | grep "32 cout_g  32"
  done
  for v in scratch/bin/libcpx_hip_*.so; do
    [ -f "$v" ] || continue
    printf "%-18s" $(basename $v .so | sed 's/libcpx_hip_//'); CPX_LIB=$PWD/$v python scratch/cnn_probe.py ${1:-1536} 2>&1 | grep "32 cout_g  32"
  done
done
