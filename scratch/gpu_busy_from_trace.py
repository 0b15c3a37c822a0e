"""Union of kernel intervals in a rocprofv3 --kernel-trace CSV: how much of the wall time the GPU had at least one kernel
running (and how much at least two).  usage: python scratch/gpu_busy_from_trace.py <kernel_trace.csv> [last_seconds]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in iv)
span = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else None
if span:
    iv = [x for x in iv if x[0] >= t_end - span]
t0 = iv[0][0]
events = []
for s, e, _ in iv:
    events.append((s, 1)); events.append((e, -1))
events.sort()
busy1 = busy2 = 0; depth = 0; last = t0
for t, d in events:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    depth += d; last = t
wall = t_end - t0
print("window %.3f s: GPU busy (>= 1 kernel) %.3f s = %.1f %%, >= 2 kernels %.3f s; kernel-time sum %.3f s" % (
    wall / 1e9, busy1 / 1e9, 100.0 * busy1 / wall, busy2 / 1e9, sum(e - s for s, e, _ in iv) / 1e9))
# the largest gaps
gaps = []; cur_end = iv[0][1]
for s, e, n in iv[1:]:
    if s > cur_end: gaps.append((s - cur_end, cur_end - t0, n))
    cur_end = max(cur_end, e)
gaps.sort(reverse=True)
print("idle total %.3f s in %d gaps; largest:" % (sum(g[0] for g in gaps) / 1e9, len(gaps)))
for g in gaps[:12]:
    print("  %.1f ms at +%.3f s before %s" % (g[0] / 1e6, g[1] / 1e9, g[2][:70]))
