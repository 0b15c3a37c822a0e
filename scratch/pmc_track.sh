#!/bin/bash
# SQ issue / wait counters and the effective clock of cpx_frame_kernel (scratch/track_probe.py B T): two rocprofv3 --pmc passes
# with the kernel trace beside them.  Output: gpurun_out/pmc_track_TAG_summary.json
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-4096}
T=${2:-60}
TAG=${3:-frame}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_track_${TAG}_a -- python3 $ROOT/scratch/track_probe.py $B $T > $ROOT/gpurun_out/pmc_track_$TAG.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA \
  --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_track_${TAG}_b -- python3 $ROOT/scratch/track_probe.py $B $T >> $ROOT/gpurun_out/pmc_track_$TAG.log 2>&1
cd $ROOT
python3 - $TAG <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = collections.defaultdict(dict)
for part in "ab":
    dur = {}
    for f in glob.glob(f"gpurun_out/pmc_track_{tag}_{part}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_track_{tag}_{part}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "cpx_frame_kernel" not in k and "cpx_median_kernel" not in k: continue
            name = k.split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
                acc[name]["duration_ns_" + part].append(dur[r["Dispatch_Id"]])
    for name, cs in acc.items():
        for c, v in cs.items():
            out[name][c if c != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE_" + part] = sum(v) / len(v)
        out[name]["launches"] = len(cs["GRBM_GUI_ACTIVE"])
for name, o in out.items():
    if "duration_ns_a" in o:
        o["clock_GHz"] = o["GRBM_GUI_ACTIVE_a"] / 8 / o["duration_ns_a"]
        cyc = o["GRBM_GUI_ACTIVE_a"] / 8            # cycles of the launch
        # SQ_ACTIVE_INST_VALU counts (per SIMD-ish unit) cycles a vector instruction is executing; 1024 SIMDs on the chip
        o["valu_active_frac_of_simd_cycles"] = o["SQ_ACTIVE_INST_VALU"] / 1024 / cyc * 4 if "SQ_ACTIVE_INST_VALU" in o else None
        o["valu_insts_per_simd_cycle"] = o["SQ_INSTS_VALU"] / 1024 / cyc
        o["wait_any_frac_of_wave_cycles"] = o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]
        o["wait_inst_any_frac_of_wave_cycles"] = o["SQ_WAIT_INST_ANY"] / o["SQ_WAVE_CYCLES"]
json.dump(out, open(f"gpurun_out/pmc_track_{tag}_summary.json", "w"), indent=1)
for name, o in sorted(out.items()):
    print(name, {k: (round(v, 3) if v < 100 else round(v)) for k, v in o.items() if v is not None})
PY
rm -rf gpurun_out/pmc_track_${TAG}_a gpurun_out/pmc_track_${TAG}_b
