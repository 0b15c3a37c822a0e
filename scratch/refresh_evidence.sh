#!/bin/bash
# Regenerates the judged evidence under gpurun_out/ev on the GPU box (copy into profiles/ afterwards):
#   bench lines (default e2e incl. default_config + fs64, f32 math, track stage, config4, ir), rocprofv3 kernel stats
#   of the default command, PMC passes (FETCH_SIZE / WRITE_SIZE separately, no trace domains) and the PMC summary.
R=${CPX_ROUND:-r06}
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/ev
# PMC passes first: bench.py reads profiles/${R}_e2e_pmc.json for every `traffic` field of the lines taken below
(cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_e2e_$d -- python3 $ROOT/bench.py --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0 > $ROOT/gpurun_out/ev/pmc_e2e_$d.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_trk_$d -- python3 $ROOT/bench.py --stage track --steps 1 --warmup 0 --cpu-clips 0 > $ROOT/gpurun_out/ev/pmc_trk_$d.log 2>&1
done
cd $ROOT
CPX_ROUND=$R python3 scratch/make_pmc_profile.py > gpurun_out/ev/pmc_summary.txt 2>&1
cp profiles/${R}_e2e_pmc.json gpurun_out/ev/
)
python3 bench.py 2> gpurun_out/ev/bench_e2e.err | grep '^{' > gpurun_out/ev/${R}_bench_e2e.json
python3 bench.py --cnn-math f32 --cpu-clips 0 --no-extras 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_bench_e2e_f32math.json
python3 bench.py --cnn-math bf16x3 --cpu-clips 0 --no-extras 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_bench_e2e_bf16x3.json
python3 bench.py --stage track 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_bench_track.json
python3 bench.py --config4 --steps 1 --warmup 1 --cpu-clips 8 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_bench_config4.json
python3 bench.py --stage ir 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_bench_ir.json
python3 scratch/dir_bench_bulk.py 4096 auto 2>/dev/null | grep '^{' > gpurun_out/ev/${R}_directory_bulk.json
python3 scratch/dir_bench_bulk.py 64 1024 --denoise 2>/dev/null | grep '^{' >> gpurun_out/ev/${R}_directory_bulk.json
python3 scratch/dir_bench_bulk.py 1024 1024 --denoise 2>/dev/null | grep '^{' >> gpurun_out/ev/${R}_directory_bulk.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_ff -- python3 $ROOT/scratch/from_files_profile.py 8192 270 > $ROOT/gpurun_out/ev/prof_from_files.log 2>&1
f=$(ls -t $ROOT/gpurun_out/prof_ff/*/*kernel_stats.csv | head -1); cp "$f" $ROOT/gpurun_out/ev/${R}_from_files_kernel_stats.csv; rm -rf $ROOT/gpurun_out/prof_ff
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_ff -- python3 $ROOT/scratch/from_files_profile.py 16384 270 --fixtures > $ROOT/gpurun_out/ev/prof_from_files_fixture.log 2>&1
f=$(ls -t $ROOT/gpurun_out/prof_ff/*/*kernel_stats.csv | head -1); cp "$f" $ROOT/gpurun_out/ev/${R}_from_files_fixture_kernel_stats.csv; rm -rf $ROOT/gpurun_out/prof_ff
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_e2e -- python3 $ROOT/bench.py --cpu-clips 0 --no-extras --from-files 0 > $ROOT/gpurun_out/ev/prof_e2e.log 2>&1
cd $ROOT
f=$(ls -t gpurun_out/prof_e2e/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/ev/${R}_e2e_kernel_stats.csv
# the counter CSVs are large: keep only the summary
rm -rf gpurun_out/pmc_e2e_* gpurun_out/pmc_trk_* gpurun_out/prof_e2e
cut -c1-300 gpurun_out/ev/${R}_bench_e2e.json; cat gpurun_out/ev/pmc_summary.txt
