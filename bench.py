#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X track stage.

    python bench.py --gpus N --steps K --warmup W [--clips B] [--frames T]

Workload (BASELINE.json configs[1]): B synthetic
160x120 uint16 clips of T=270 frames (30 s at 9 fps) per GPU, lepton3
thresholds, resident in HBM; one *step* = one pass of the track hot path over the
batch (background update, filtered frame, 8-bit blur / threshold / close,
8-connected labelling, component statistics + delta variance, label and filtered
images written).  Whole-job frames/s = N * B * T * K / max-over-ranks time.

For N > 1 the driver launches one rank per GPU through torch.distributed.run;
clips shard across ranks with no data-path collective; the per-clip result
records are all-gathered once per step over RCCL (north_star).

The JSON line also carries
  roofline     : cpx_frame_kernel, HBM-bound; achieved = 614,400 algorithmic bytes
                 per frame x B frames per launch / average launch
                 duration measured with HIP events on the handle's stream
  cpu_baseline : the NumPy oracle ("port", 1 core) timed on a bounded sample of
                 the same workload on this host (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(REPO, "classifier-pipeline_amd"), os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

ALGO_BYTES_PER_FRAME = 614400  # 32 B / pixel at 160x120
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def synth_on_device(torch, device, n_clips, n_frames, seed, h=120, w=160, chunk=64):
    """Synthetic clips generated on the GPU (same recipe as cpx.synth): smooth
    background 2900 +- 40, sensor noise N(0, 4), up to 3 warm Gaussian blobs on
    a random walk.  -> int16-bit-pattern uint16 tensor [n_clips*n_frames, h, w]."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_clips * n_frames, h, w), dtype=torch.int16, device=device)
    yy = torch.arange(h, device=device, dtype=torch.float32).view(1, 1, h, 1)
    xx = torch.arange(w, device=device, dtype=torch.float32).view(1, 1, 1, w)
    for c0 in range(0, n_clips, chunk):
        nc = min(chunk, n_clips - c0)
        coarse = torch.randn((nc, 1, 7, 7), generator=g, device=device)
        bg = 2900.0 + 40.0 * torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True)
        fr = bg.expand(nc, n_frames, h, w).clone()
        fr += 4.0 * torch.randn((nc, n_frames, h, w), generator=g, device=device)
        nblob = torch.randint(0, 4, (nc,), generator=g, device=device)
        for k in range(3):
            on = (nblob > k).float().view(nc, 1, 1, 1)
            sigma = 3.0 + 5.0 * torch.rand((nc, 1, 1, 1), generator=g, device=device)
            amp = 60.0 + 340.0 * torch.rand((nc, 1, 1, 1), generator=g, device=device)
            x0 = torch.rand((nc, 1), generator=g, device=device) * w
            y0 = torch.rand((nc, 1), generator=g, device=device) * h
            vel = 3.0 * (torch.rand((nc, 2), generator=g, device=device) - 0.5)
            walk = 0.4 * (torch.rand((nc, n_frames, 2), generator=g, device=device) - 0.5)
            v = (vel.view(nc, 1, 2) + torch.cumsum(walk, dim=1)).clamp_(-3.0, 3.0)
            pos = torch.cumsum(v, dim=1)
            start = torch.randint(0, max(1, n_frames // 2), (nc, 1), generator=g, device=device)
            t = torch.arange(n_frames, device=device).view(1, n_frames)
            alive = (t >= start).float().view(nc, n_frames, 1, 1)
            px = (x0 + pos[:, :, 0]).view(nc, n_frames, 1, 1)
            py = (y0 + pos[:, :, 1]).view(nc, n_frames, 1, 1)
            d2 = (yy - py) ** 2 + (xx - px) ** 2
            fr += on * alive * amp * torch.exp(-d2 / (2.0 * sigma * sigma))
        q = fr.round_().clamp_(0, 65535).to(torch.int32)
        out[c0 * n_frames:(c0 + nc) * n_frames] = q.view(nc * n_frames, h, w).to(torch.int16)  # keeps the low 16 bits
        del fr, q
    return out


def cpu_baseline(n_clips, n_frames, seed):
    """The oracle restatement (NumPy, single core) over a bounded sample of the same workload."""
    import numpy as np

    import track_oracle as to
    from cpx import synth

    frames, offs = synth.make_batch(n_clips, n_frames, seed=seed)
    cfg = to.OracleConfig("lepton3")
    t0 = time.perf_counter()
    for b in range(n_clips):
        to.track_clip(frames[offs[b]:offs[b + 1]], cfg=cfg, keep=True, do_tracking="regions")
    dt = time.perf_counter() - t0
    return {
        "value": round(n_clips * n_frames / dt, 1),
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": "%d synthetic clips x %d frames (oracle/track_oracle.py pixel stage + regions, NumPy, %.1f s)"
        % (n_clips, n_frames, dt),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default 4096, reduced if HBM is short)")
    ap.add_argument("--frames", type=int, default=270)
    ap.add_argument("--cpu-clips", type=int, default=12, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-outputs", action="store_true", help="do not write label / filtered images")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    from cpx.engine import TrackEngine

    H, W, T = 120, 160, args.frames
    P = H * W
    want_out = not args.no_outputs
    B = args.clips or 4096
    free, _ = torch.cuda.mem_get_info(device)
    per_clip = T * P * 2 + T * 64 * 32 + T * 80 + 6 * P * 4 + (T * P * 8 if want_out else 0)
    while B > 64 and B * per_clip > 0.80 * free:
        B //= 2
    eng = TrackEngine(width=W, height=H, model="lepton3", device=local_rank, max_components=64,
                      max_frames=max(T, 45))
    frames = synth_on_device(torch, device, B, T, seed=1234 + rank)
    offs = (np.arange(B + 1, dtype=np.int64) * T).astype(np.int32)
    t_on = [100000 + 114 * i for i in range(T)]
    ffc = [40000] * T
    meta = np.tile(eng.make_meta(T, t_on, ffc), B)
    total = B * T
    comps = torch.empty(total * 64 * 8, dtype=torch.int32, device=device)
    info = torch.empty(total * 20, dtype=torch.int32, device=device)
    labels = torch.empty((total, H, W), dtype=torch.int32, device=device) if want_out else None
    filt = torch.empty((total, H, W), dtype=torch.float32, device=device) if want_out else None
    outputs = (comps, info, labels, filt, None)
    gather_in = torch.zeros((B, 4), dtype=torch.int32, device=device)
    gather_out = torch.empty((world * B, 4), dtype=torch.int32, device=device) if world > 1 else None

    def step():
        res = eng.track_batch(frames, offs, meta, outputs=outputs)
        eng.synchronize()
        if world > 1:
            # per-clip result records -> every rank (north_star: RCCL all-gather of per-clip results)
            nc = info.view(total, 20)[:, 1].view(B, T).sum(dim=1)
            gather_in[:, 0] = rank
            gather_in[:, 1] = torch.arange(B, device=device, dtype=torch.int32)
            gather_in[:, 2] = nc.to(torch.int32)
            dist.all_gather_into_tensor(gather_out, gather_in)
        return res

    def fence():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    kernel_ms, kernel_launches = 0.0, 0
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        ms, n = eng.last_kernel_timing()
        kernel_ms += ms
        kernel_launches += n
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    res.check()
    n_regions = int(res.info["n_components"].sum())

    if rank == 0:
        frames_done = world * B * T * args.steps
        avg_launch_s = kernel_ms / 1e3 / max(kernel_launches, 1)
        bytes_per_launch = (ALGO_BYTES_PER_FRAME if want_out else ALGO_BYTES_PER_FRAME - 76800) * B
        achieved = bytes_per_launch / avg_launch_s / 1e9
        line = {
            "metric": "CPTV frames/s (track stage: background + region-label HIP kernels) at 160x120",
            "value": round(frames_done / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16/i32 (f32 normalise, f64 background weights)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: synthetic 160x120 uint16 clips, background + region-label kernels",
                "clips_per_gpu": B,
                "frames_per_clip": T,
                "camera_model": "lepton3",
                "outputs": "components + label image + filtered image" if want_out else "components only",
                "components_found": n_regions,
                "sharding": "clips across ranks, all_gather of per-clip records" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": "cpx_frame_kernel",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "avg_launch_us": round(avg_launch_s * 1e6, 2),
                "launches": kernel_launches,
                "algorithmic_bytes_per_launch": bytes_per_launch,
            },
        }
        if world == 1 and args.cpu_clips > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_clips, T, seed=1234)
        print(json.dumps(line))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
