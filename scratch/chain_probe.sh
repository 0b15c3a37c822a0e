#!/bin/bash
# time from the end of cpx_frame_kernel to the start of the first convolution kernel of the last bench step (kernel trace)
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-clips 0 --no-extras --from-files 0 > $GRAFT_REPO_ROOT/gpurun_out/tl.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tl/*/*_kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
fk = [r for r in rows if "cpx_frame_kernel" in r[2]][-1]
after = [r for r in rows if r[0] >= fk[1]]
conv = [r for r in after if "conv_" in r[2]][0]
med = [r for r in after if "cpx_median_kernel" in r[2]][0]
names = {}
for s, e, n in after:
    if s >= conv[0]: break
    k = n.split("(")[0].split("::")[-1][:28]
    names[k] = names.get(k, 0) + (e - s) / 1e6
print("frame end -> first conv: %.2f ms; median %.2f ms; %s" % ((conv[0] - fk[1]) / 1e6, (med[1] - med[0]) / 1e6,
      ", ".join("%s %.2f" % kv for kv in names.items() if kv[1] > 0.3)))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tl
