"""Timeline of the LAST bench step in a rocprofv3 --kernel-trace CSV (the step starts at the last cpx_median_kernel): runs of
launches of one kernel with their busy time, and every idle gap of at least 0.1 ms.  usage: step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, x in enumerate(iv) if "cpx_median_kernel" in x[2]]
i0 = starts[-1]
iv = iv[i0:]
t0 = iv[0][0]
def short(n):
    n = n.replace("cpx::(anonymous namespace)::", "").replace("cpx::", "").replace("void ", "")
    return n.split("(")[0][:60]
cur_end = iv[0][0]
run = None
idle = 0
busy = 0
out = []
for s, e, n in iv:
    gap = s - cur_end
    if gap >= 100000:
        if run: out.append(run); run = None
        out.append(("   idle", gap, s - gap - t0, 0))
    if gap > 0: idle += gap
    if run and run[0] == short(n):
        run = (run[0], run[1] + (e - s), run[2], run[3] + 1)
    else:
        if run: out.append(run)
        run = (short(n), e - s, s - t0, 1)
    busy += max(0, e - max(s, cur_end))
    cur_end = max(cur_end, e)
if run: out.append(run)
for name, ns, at, cnt in out:
    print("+%8.2f ms  %8.2f ms  x%-4d %s" % (at / 1e6, ns / 1e6, cnt, name))
print("step window %.2f ms, busy %.2f ms, idle %.2f ms" % ((cur_end - t0) / 1e6, busy / 1e6, idle / 1e6))
