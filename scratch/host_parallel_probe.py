"""How much host parallelism does the GPU box really give?  zlib inflate of a fixture CPTV on T threads
(zlib releases the GIL) and on T processes.  Prints one JSON line per T."""
import json, os, sys, time, zlib, threading
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = open(os.path.join(REPO, "tests", "golden", "possum.cptv"), "rb").read()
out_len = len(zlib.decompressobj(47).decompress(raw))
print(json.dumps({"nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)),
                  "cpu_max": open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None}))
def work(n):
    for _ in range(n):
        zlib.decompressobj(47).decompress(raw)
for T in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    if T > 2 * (os.cpu_count() or 1):
        break
    per = 12
    th = [threading.Thread(target=work, args=(per,)) for _ in range(T)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(json.dumps({"threads": T, "MBps_out": round(T * per * out_len / dt / 1e6, 1),
                      "frames_per_s": round(T * per * 161 / dt, 1), "per_thread_MBps": round(per * out_len / dt / 1e6, 1)}), flush=True)
