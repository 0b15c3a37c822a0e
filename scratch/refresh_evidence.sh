#!/bin/bash
# Regenerates the judged evidence under gpurun_out/ on the GPU box (copy into profiles/ afterwards):
#   bench lines (default e2e, f32 math, track stage, denoise, fs64), rocprofv3 kernel stats of the default command,
#   PMC passes (FETCH_SIZE / WRITE_SIZE separately, no trace domains) and the PMC summary.
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/ev
python3 bench.py > gpurun_out/ev/r01_bench_e2e.json 2> gpurun_out/ev/bench_e2e.err
python3 bench.py --cnn-math f32 --cpu-clips 0 > gpurun_out/ev/r01_bench_e2e_f32math.json 2>/dev/null
python3 bench.py --stage track > gpurun_out/ev/r01_bench_track.json 2>/dev/null
python3 bench.py --stage track --denoise --cpu-clips 0 > gpurun_out/ev/r01_bench_track_denoise.json 2>/dev/null
python3 bench.py --frame-size 64 --cpu-clips 0 > gpurun_out/ev/r01_bench_e2e_fs64.json 2>/dev/null
python3 bench.py --stage ir > gpurun_out/ev/r01_bench_ir.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_e2e -- python3 $ROOT/bench.py --cpu-clips 0 > $ROOT/gpurun_out/ev/prof_e2e.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_e2e_$d -- python3 $ROOT/bench.py --clips 1024 --steps 1 --warmup 0 --cpu-clips 0 > $ROOT/gpurun_out/ev/pmc_e2e_$d.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $ROOT/gpurun_out/pmc_trk_$d -- python3 $ROOT/bench.py --stage track --clips 1024 --steps 1 --warmup 0 --cpu-clips 0 > $ROOT/gpurun_out/ev/pmc_trk_$d.log 2>&1
done
cd $ROOT
mkdir -p profiles_tmp
python3 scratch/make_pmc_profile.py > gpurun_out/ev/pmc_summary.txt 2>&1
cp profiles/r01_e2e_pmc.json gpurun_out/ev/
f=$(ls -t gpurun_out/prof_e2e/*/*kernel_stats.csv | head -1); cp "$f" gpurun_out/ev/r01_e2e_kernel_stats.csv
# the counter CSVs are large: keep only the summary
rm -rf gpurun_out/pmc_e2e_* gpurun_out/pmc_trk_* gpurun_out/prof_e2e profiles_tmp
tail -1 gpurun_out/ev/r01_bench_e2e.json | cut -c1-400; cat gpurun_out/ev/pmc_summary.txt
