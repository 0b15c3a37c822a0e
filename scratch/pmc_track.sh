#!/bin/bash
# SQ counters of cpx_frame_kernel (separate passes, counters only): bench.py --stage track --clips 1024 under rocprofv3 --pmc
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc_trk_$i -- python3 $ROOT/bench.py --stage track --clips 1024 --steps 1 --warmup 0 --cpu-clips 0 > $ROOT/gpurun_out/pmc_trk_$i.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/pmc_trk_[0-9]*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "cpx_frame_kernel" not in k and "cpx_median_kernel" not in k: continue
            acc[k.split("(")[0] + " grid " + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, cs in acc.items():
    out[name] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[name]["launches"] = len(next(iter(cs.values())))
json.dump(out, open("gpurun_out/pmc_trk_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_trk_[0-9]*/
