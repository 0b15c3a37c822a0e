#!/bin/bash
# SQ counters of cpx_cptv_inflate_kernel (separate passes, counters + kernel trace): scratch/inflate_pmc_probe.py N kind
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-4096}; KIND=${2:-synthetic}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_inf_$i -- python3 $ROOT/scratch/inflate_pmc_probe.py $N $KIND > $ROOT/gpurun_out/pmc_inf_$i.log 2>&1
done
cd $ROOT
python3 - $N $KIND <<'PY'
import csv, glob, collections, json, sys
acc = collections.defaultdict(list); dur = []
for d in sorted(glob.glob("gpurun_out/pmc_inf_*/")):
    for f in glob.glob(d + "*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "inflate" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for f in glob.glob(d + "*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "inflate" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: sum(v) / len(v) for c, v in acc.items()}
out["launches_counted"] = len(acc.get("SQ_WAVES", []))
out["kernel_ns_avg"] = sum(dur) / max(len(dur), 1)
probe = [json.loads(l) for l in open("gpurun_out/pmc_inf_1.log") if l.startswith("{")]
out["probe"] = probe[-1] if probe else None
json.dump(out, open("gpurun_out/pmc_inflate_%s_%s.json" % (sys.argv[1], sys.argv[2]), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc_inf_[0-9]*/
