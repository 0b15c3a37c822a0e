#!/bin/bash
# re-takes profiles/${R}_conv_sq_counters.json and profiles/${R}_conv_traffic.json (GPU box): scratch/pmc_clock.sh and
# scratch/pmc_conv_traffic.sh over scratch/cnn_probe.py 1536, the `what` texts of the committed files kept
R=${CPX_ROUND:-r06}
cd "$(dirname "$0")/.."
scratch/pmc_clock.sh 1536 $R > gpurun_out/pmc_clock_$R.txt 2>&1
scratch/pmc_conv_traffic.sh 1536 > gpurun_out/pmc_traffic_$R.txt 2>&1
python3 - $R <<'PY'
import json, sys
r = sys.argv[1]
for prof, fresh in ((f"profiles/{r}_conv_sq_counters.json", f"gpurun_out/pmc_{r}_summary.json"),
                    (f"profiles/{r}_conv_traffic.json", "gpurun_out/pmc_conv_traffic.json")):
    old = json.load(open(prof))
    new = json.load(open(fresh))
    json.dump({"what": old["what"], "kernels": new}, open("gpurun_out/" + prof.split("/")[1], "w"), indent=1)
    print(prof, len(new), "kernels")
PY
