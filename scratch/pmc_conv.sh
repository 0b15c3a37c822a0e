#!/bin/bash
# SQ counters of the convolution kernels (separate passes, counters only): scratch/cnn_probe.py 512 under rocprofv3 --pmc
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/pmc_conv_$i -- python3 $ROOT/scratch/cnn_probe.py 512 > $ROOT/gpurun_out/pmc_conv_$i.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/pmc_conv_*/")):
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv_" not in k: continue
            name = k[k.index("conv_"):].split("(")[0] + " grid " + r["Grid_Size"]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, cs in acc.items():
    out[name] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[name]["launches"] = len(next(iter(cs.values())))
json.dump(out, open("gpurun_out/pmc_conv_summary.json", "w"), indent=1)
for name, cs in sorted(out.items()):
    if "block32" in name or "rw_kernel<1" in name: print(name, {k: round(v) for k, v in cs.items()})
PY
rm -rf gpurun_out/pmc_conv_[0-9]*/
