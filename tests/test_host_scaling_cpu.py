"""Eight ranks' HOST stages on one node: the file-fed path runs, per rank, a staging copy of the
recording's bytes and the metadata text of its tracks on the host while the device decodes / tracks / classifies.
This test runs exactly those two stages for 1, 2, 4 and 8 concurrent worker processes (one per rank) on the CPUs this
container grants, checks that nothing serialises them (aggregate rate grows with the workers up to the CPU count), and
prints the rank count at which a node's CPU quota saturates for the per-rank device rate measured on the GPU
(profiles/r04_bench_e2e.json from_files: ~1,900 recordings/s per rank).  No GPU."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "classifier-pipeline_amd")
DEVICE_RECORDINGS_PER_S_PER_RANK = 1900.0   # from_files, one MI355X (synthetic 270-frame recordings and the fixtures alike)


def _host_half(n_recordings, seed):
    """Staging copy + metadata text for n recordings of 270 frames with one track of 60 regions and one prediction
    entry each (what the headline's clips carry on average is less: 0.7 tracks)."""
    sys.path.insert(0, PKG)
    from datetime import datetime

    from cpx.config import Config
    from cpx.track.bulk import BulkTracker
    from cpx.track.clip import Clip
    from cpx.tracking import REGION_DTYPE, TRACK_SUMMARY_DTYPE

    rng = np.random.default_rng(seed)
    cfg = Config.get_defaults()
    tracker = BulkTracker(cfg)
    from cpx import _lib

    tracker.lib = _lib.load()     # cpx_format_regions is host code of the C-ABI library (no GPU needed)
    blob = rng.integers(0, 256, 3_490_000, dtype=np.uint8)      # the size of a synthetic level-6 recording
    stage = np.empty(blob.size + 16, np.uint8)
    regs = np.zeros(60, REGION_DTYPE)
    regs["x"], regs["y"] = rng.integers(0, 100, 60), rng.integers(0, 80, 60)
    regs["width"], regs["height"], regs["mass"] = 20, 18, rng.integers(10, 200, 60)
    regs["frame_number"] = np.arange(40, 100)
    regs["pixel_variance"] = rng.random(60) * 30
    summ = np.zeros(1, TRACK_SUMMARY_DTYPE)[0]
    summ["id"], summ["start_frame"], summ["score"] = 1, 40, 321.5
    pred = [{"model_id": 1, "tag": "possum", "confidence": 0.9, "clarity": 0.8, "all_class_confidences": {str(i): 0.05 for i in range(17)},
             "predictions": [{"frames": list(range(25)), "prediction": [5] * 17, "mass": 2000}] * 3}]
    t0 = time.perf_counter()
    n_bytes = 0
    for i in range(n_recordings):
        stage[: blob.size] = blob                                  # the staging copy (pinned memory on a GPU box)
        clip = Clip(tracker.tcfg, "/tmp/r%05d.cptv" % i)
        clip.frames_per_second = 9
        clip.set_res(160, 120)
        clip.set_model("lepton3")
        clip.set_video_stats(datetime.fromtimestamp(1_600_000_000 + i).astimezone(Clip.local_tz))
        text = tracker.metadata_text(clip, 270, [dict(summary=summ, regions=regs, thumb=None, predictions=pred)], None,
                                     "/tmp/r%05d.cptv" % i, 0.01, None, 4)
        n_bytes += len(text)
    return time.perf_counter() - t0, n_bytes


def _worker(args):
    return _host_half(*args)


def test_host_stages_of_eight_ranks_scale_with_the_cpus():
    sys.path.insert(0, PKG)
    from cpx.sharding import host_threads_per_rank, usable_cpus

    cpus = usable_cpus()
    n = 48
    ctx = mp.get_context("spawn")

    def measure():
        rates = {}
        for workers in (1, 2, 4, 8):
            with ctx.Pool(workers) as pool:
                pool.map(_worker, [(2, 1)] * workers)                  # imports, first touch
                t0 = time.perf_counter()
                res = pool.map(_worker, [(n, 100 + w) for w in range(workers)])
                wall = time.perf_counter() - t0
            assert all(nb > 1000 * n for _, nb in res)
            rates[workers] = workers * n / wall
        return rates

    def scales(rates):
        return all(rates[w] >= 0.5 * w * rates[1] for w in (2, 4, 8) if w <= cpus)

    # (a wall-clock property measured on a machine other things run on: a measurement that something else disturbed is
    # taken again, twice at most -- a real serialisation fails all three)
    rates = measure()
    for _ in range(2):
        if scales(rates):
            break
        rates = measure()
    per_recording_cpu_s = 1.0 / rates[1]
    saturate_ranks = cpus / (DEVICE_RECORDINGS_PER_S_PER_RANK * per_recording_cpu_s)
    report = {"usable_cpus": cpus, "host_recordings_per_s": {str(k): round(v, 1) for k, v in rates.items()},
              "host_cpu_ms_per_recording": round(per_recording_cpu_s * 1e3, 3),
              "device_recordings_per_s_per_rank": DEVICE_RECORDINGS_PER_S_PER_RANK,
              "ranks_this_cpu_quota_feeds": round(saturate_ranks, 1),
              "cpus_needed_for_8_ranks": round(8 * DEVICE_RECORDINGS_PER_S_PER_RANK * per_recording_cpu_s, 1)}
    print("\nHOST-SCALING " + json.dumps(report))
    # nothing serialises the ranks' host stages: with w <= CPUs workers the aggregate rate is at least 0.5 w x one worker's
    for workers in (2, 4, 8):
        if workers <= cpus:
            assert rates[workers] >= 0.5 * workers * rates[1], (workers, rates)
    # the thread cap shares the quota among the ranks of a node
    os.environ["LOCAL_WORLD_SIZE"] = "8"
    try:
        assert host_threads_per_rank(16) == max(1, min(16, cpus // 8))
    finally:
        del os.environ["LOCAL_WORLD_SIZE"]
    assert host_threads_per_rank(16) == max(1, min(16, cpus))


def test_ranks_pin_to_their_gpus_numa_node(tmp_path):
    """cpx.sharding.pin_to_gpu_numa on a fake sysfs of a two-socket node with four GPUs (and another vendor's display
    function in between): the rank's card by PCI order, HIP_VISIBLE_DEVICES honoured, numa_node -1 and unknown cards left
    alone, the affinity only ever narrowed to CPUs the process already had."""
    import os

    from cpx import sharding

    sysfs = tmp_path / "sys"
    cards = [("0000:05:00.0", "0x1002", 0), ("0000:26:00.0", "0x1002", 0), ("0000:30:00.0", "0x1a03", 0),
             ("0000:85:00.0", "0x1002", 1), ("0000:a6:00.0", "0x1002", 1)]
    for k, (addr, vendor, node) in enumerate(cards):
        dev = sysfs / "devices" / "pci0000:00" / addr
        dev.mkdir(parents=True)
        (dev / "vendor").write_text(vendor + "\n")
        (dev / "numa_node").write_text("%d\n" % node)
        card = sysfs / "class" / "drm" / ("card%d" % k)
        card.mkdir(parents=True)
        os.symlink(dev, card / "device")
    allowed = sorted(os.sched_getaffinity(0))
    half = max(1, len(allowed) // 2)
    lists = {0: allowed[:half], 1: allowed[half:] or allowed[:1]}
    for node, cpus in lists.items():
        d = sysfs / "devices" / "system" / "node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + "\n")
    assert [sharding.gpu_numa_node(r, str(sysfs), visible="") for r in range(5)] == [0, 0, 1, 1, None]
    assert sharding.gpu_numa_node(0, str(sysfs), visible="3,2") == 1
    assert sharding.gpu_numa_node(0, str(sysfs), visible="9") is None
    got = sharding.pin_to_gpu_numa(2, str(sysfs), apply=False)
    assert got["node"] == 1 and got["cpus"] == len(set(lists[1]) & set(allowed))
    assert sharding._parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    # applied in a child process (the affinity of THIS test process stays as it is)
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); from cpx import sharding; r = sharding.pin_to_gpu_numa(0, %r); "
            "print(r['pinned'], sorted(os.sched_getaffinity(0)) == %r)" % (
                os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "classifier-pipeline_amd"),
                str(sysfs), sorted(lists[0])))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    if len(allowed) > 1:
        assert out.stdout.split() == ["True", "True"], out.stdout
    # a host whose sysfs names no node: nothing changes
    (sysfs / "devices" / "pci0000:00" / "0000:05:00.0" / "numa_node").write_text("-1\n")
    assert sharding.pin_to_gpu_numa(0, str(sysfs), apply=False) == {"node": None, "cpus": len(allowed), "pinned": False}
