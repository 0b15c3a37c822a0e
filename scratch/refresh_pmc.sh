#!/bin/bash
# the PMC part of refresh_evidence.sh alone (FETCH_SIZE / WRITE_SIZE in separate passes, no trace domains)
R=${CPX_ROUND:-r06}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/ev
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_e2e_$d -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0 > $ROOT/gpurun_out/ev/pmc_e2e_$d.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_trk_$d -- python3 $ROOT/bench.py --stage track --steps 1 --warmup 0 --cpu-clips 0 > $ROOT/gpurun_out/ev/pmc_trk_$d.log 2>&1
done
cd $ROOT
CPX_ROUND=$R python3 scratch/make_pmc_profile.py > gpurun_out/ev/pmc_summary.txt 2>&1
cp profiles/${R}_e2e_pmc.json gpurun_out/ev/
rm -rf gpurun_out/pmc_e2e_* gpurun_out/pmc_trk_*
cat gpurun_out/ev/pmc_summary.txt
