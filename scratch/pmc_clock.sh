#!/bin/bash
# effective clock + SQ occupancy counters of the convolution kernels: one rocprofv3 --pmc pass with the kernel trace
# (durations) beside it.  clock = GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS give-back)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1024}
TAG=${2:-clock}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_$TAG -- python3 $ROOT/scratch/cnn_probe.py $N > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
cd $ROOT
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
dur = {}
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
# persistent kernels have one grid whatever the batch: the probe's calibration forwards (32 samples) would dilute the means --
# only launches that ran at least half as long as the kernel's longest one are kept
longest = collections.defaultdict(int)
for d, (ns, k) in dur.items():
    longest[k] = max(longest[k], ns)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_" not in k: continue
        if r["Dispatch_Id"] in dur and dur[r["Dispatch_Id"]][0] * 2 < longest[k]: continue
        name = k[k.index("conv_"):].split("(")[0] + " grid " + r["Grid_Size"]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            acc[name]["duration_ns"].append(dur[r["Dispatch_Id"]][0])
out = {}
for name, cs in acc.items():
    o = {c: sum(v) / len(v) for c, v in cs.items()}
    o["launches"] = len(cs["GRBM_GUI_ACTIVE"])
    if "duration_ns" in o and o["duration_ns"] > 0:
        o["clock_GHz"] = o["GRBM_GUI_ACTIVE"] / 8 / o["duration_ns"]
        o["mfma_busy_frac_of_cycles"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (o["GRBM_GUI_ACTIVE"] / 8)
        # 16-bit MFMA FLOPs issued against the 2.5 PFLOP/s dense peak: 16,384 per v_mfma_f32_16x16x32_f16, 32,768 per
        # v_mfma_f32_32x32x16_f16 (the flattened kernel and conv_bf3_kernel)
        per = 32768 if ("conv_bf3flat_kernel" in name or name.startswith("conv_bf3_kernel<")) else 16384
        o["mfma_flops_per_instruction"] = per
        o["mfma_issued_tflops"] = o["SQ_INSTS_MFMA"] * per / o["duration_ns"] / 1e3
        o["frac_of_16bit_mfma_peak"] = o["mfma_issued_tflops"] / 2500.0
    out[name] = o
json.dump(out, open(f"gpurun_out/pmc_{tag}_summary.json", "w"), indent=1)
for name, o in sorted(out.items()):
    print(name, {k: (round(v, 3) if v < 100 else round(v)) for k, v in o.items()})
PY
rm -rf gpurun_out/pmc_$TAG/
