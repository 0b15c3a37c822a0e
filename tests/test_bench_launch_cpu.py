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
