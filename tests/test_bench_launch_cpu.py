"""bench.py --gpus N must produce N ranks by itself: without a launcher in the environment it
starts `torch.distributed.run` as a child process and relays rank 0's line.  --dry-launch stops after the rendezvous
(gloo), so the launch path is checked on a CPU host."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_gpus_2_spawns_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "3"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = lines[0]
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1] and line["steps"] == 3 and line["dry_launch"] is True


def test_gpus_8_spawns_eight_ranks_and_assembles_the_line():
    """The first real 8-GPU run must not be the first time eight ranks meet: --gpus 8 --dry-launch runs the host side of
    the multi-GPU bench on gloo -- LPT partition of configs[3], one gather_records exchange, the per-rank from_files
    aggregation, per-rank timings -- and rank 0's line has the keys of the real one."""
    sys.path.insert(0, REPO)
    import bench

    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dry-launch", "--steps", "2"],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    (line,) = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert line["n_gpus"] == 8 and line["ranks"] == list(range(8)) and line["dry_launch"] is True
    for k in bench.LINE_KEYS + bench.MULTI_RANK_KEYS:
        assert k in line, k
    c4 = line["config4"]
    assert len(c4["frames_per_rank"]) == 8 and sum(c4["frames_per_rank"]) == c4["frames_total"]
    assert 1.0 <= c4["imbalance"] < 1.001                       # LPT over 10,000 clips of 90-540 frames
    assert c4["records_gathered"] == 10000 and c4["record_width"] == 2 + bench.N_LABELS
    ff = line["from_files"]
    assert ff["ranks"] == 8 and len(ff["per_rank"]) == 8 and all("numa" in r for r in ff["per_rank"])
    pr = line["per_rank"]
    assert len(pr["ms_per_step"]) == 8 and pr["ms_per_step_max"] >= pr["ms_per_step_min"] and pr["imbalance"] == 1.0


def test_gpus_1_skeleton_has_the_default_lines_keys():
    """N = 1 of the driver's scaling run is compared with the default run: same keys (minus the multi-rank objects)."""
    sys.path.insert(0, REPO)
    import bench

    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-launch"],
                         env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    (line,) = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert all(k in line for k in bench.LINE_KEYS) and not any(k in line for k in bench.MULTI_RANK_KEYS)
    # ... and the real line is assembled with the same names (the source names every key of the skeleton)
    src = open(os.path.join(REPO, "bench.py")).read()
    for k in bench.LINE_KEYS:
        assert '"%s"' % k in src, k


def test_gpus_1_runs_in_process():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--dry-launch"],
                         env=_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    (line,) = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert line["n_gpus"] == 1 and line["ranks"] == [0]


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_launcher_world_is_adopted_without_gpus_flag():
    """`torchrun --nproc-per-node 2 bench.py` (no --gpus): the launcher's WORLD_SIZE is the answer."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--dry-launch"], capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1]


def test_stray_world_size_without_a_launcher_is_ignored():
    """WORLD_SIZE=1 left in the environment (no RANK: nobody launched us) must not veto --gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["WORLD_SIZE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == [0, 1]


def test_per_rank_from_files_aggregation():
    """bench.py --gpus N, N > 1: every rank runs the file-fed leg on its share; rank 0 reports the per-rank split and
    frames of all ranks over the slowest rank's time (the gather itself is torch's all_gather_object)."""
    sys.path.insert(0, REPO)
    import bench

    def rank(seconds, frames=1000, **kw):
        return dict({"files": 10, "frames": frames, "seconds": seconds, "frames_per_s": frames / seconds,
                     "split_s": {"metadata_host": 0.1}, "roofline_inflate": {"frac": 0.68}, "what": "x"}, **kw)

    agg = bench.aggregate_from_files([rank(2.0), rank(4.0, frames=3000), {"error": "RuntimeError: boom"}, None], 1024, 16)
    assert agg["ranks"] == 4 and agg["frames"] == 4000 and agg["seconds_slowest_rank"] == 4.0 and agg["frames_per_s"] == 1000.0
    assert [r.get("error") for r in agg["per_rank"]] == [None, None, "RuntimeError: boom", "no result"]
    assert agg["per_rank"][1]["split_s"] == {"metadata_host": 0.1} and agg["host_cpus_usable"] == 16
    assert "frames_per_s" not in bench.aggregate_from_files([{"error": "x"}], 512, 8)
