#!/bin/bash
# scratch/build_variant.sh NAME "-DFOO=1 -DBAR=2"  -> scratch/bin/libcpx_hip_NAME.so (experiment builds; never shipped).
# The shipped Makefile with extra flags, its objects under /tmp: per-file flags (cpx_cnn_rw / cpx_cnn_blk without the VGPR
# form, ...) are the shipped ones.  Select with CPX_LIB=$PWD/scratch/bin/libcpx_hip_NAME.so
set -e
cd "$(dirname "$0")/../classifier-pipeline_amd/csrc"
mkdir -p ../../scratch/bin /tmp/cpxvar/$1
make OBJDIR=/tmp/cpxvar/$1 OUT=../../scratch/bin/libcpx_hip_$1.so EXTRA="$2" 2>&1 | grep -E "error|Error" || true
ls -la ../../scratch/bin/libcpx_hip_$1.so
