#!/bin/bash
# HBM traffic of the convolution kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes over scratch/cnn_probe.py N
# (FETCH_SIZE is doubled when read: gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
N=${1:-1536}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $ROOT/gpurun_out/pmc_traffic_$c -- python3 $ROOT/scratch/cnn_probe.py $N > $ROOT/gpurun_out/pmc_traffic_$c.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    dur = {}
    for f in glob.glob(f"gpurun_out/pmc_traffic_{c}/*/*_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
    longest = collections.defaultdict(int)
    for d, (ns, k) in dur.items():
        longest[k] = max(longest[k], ns)
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_traffic_{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv" not in k or r["Counter_Name"] != c: continue
            if r["Dispatch_Id"] in dur and dur[r["Dispatch_Id"]][0] * 2 < longest[k]: continue
            name = k[k.index("conv"):].split("(")[0]
            acc[name].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], (0,))[0]))
    for name, v in acc.items():
        kb = sum(x[0] for x in v) / len(v)   # the counter's unit is KB
        out[name][c + "_KB_per_launch_raw"] = kb
        out[name]["launches"] = len(v)
        out[name]["duration_ns_" + c] = sum(x[1] for x in v) / len(v)
for name, o in out.items():
    if "FETCH_SIZE_KB_per_launch_raw" in o:
        o["hbm_read_GB_per_launch"] = 2 * o["FETCH_SIZE_KB_per_launch_raw"] * 1024 / 1e9   # (doubled: see above)
    if "WRITE_SIZE_KB_per_launch_raw" in o:
        o["hbm_write_GB_per_launch"] = o["WRITE_SIZE_KB_per_launch_raw"] * 1024 / 1e9
json.dump(out, open("gpurun_out/pmc_conv_traffic.json", "w"), indent=1)
for name, o in sorted(out.items()):
    print(name[:70], {k: round(v, 3) for k, v in o.items() if "GB" in k or k == "launches"})
PY
rm -rf gpurun_out/pmc_traffic_FETCH_SIZE gpurun_out/pmc_traffic_WRITE_SIZE
