#!/usr/bin/env python3
"""bench.py -- headline benchmark: CPTV frames/s end-to-end (track + classify) at 160x120.

    python bench.py --gpus N --steps K --warmup W [--clips B] [--frames T] [--stage e2e|track]

Workload (BASELINE.json metric; configs[1] + [2] chained as in configs[3]): B synthetic 160x120
uint16 clips of T = 270 frames (30 s at 9 fps) per GPU, lepton3 thresholds, resident in HBM when
the timed region starts.  One *step* = one pass of the whole hot path over the batch:
  track stage (background, filtered, blur/threshold/close, labelling, statistics, delta variance)
  -> association (region filter, matching, Kalman) -> end-of-clip filtering -> segment plan
  -> crop / resize / normalise / 5x5 tile -> WR-ResNet-22-4 forward (float32, 17 labels, seeded
  random weights: no checkpoint can be downloaded) -> per-track aggregation.
value = N * B * T * K / max-over-ranks time (barrier + device sync on both sides).

N > 1: the driver launches one rank per GPU (torch.distributed.run); clips shard across ranks with
no data-path collective; the per-track result records are all-gathered once per step over RCCL.

Extra objects on the JSON line:
  roofline      the kernel the step spends most of its time in (its `dominant` field says which and how long).  e2e, default
                fp16x2 math: cpx_frame_kernel (one launch per step, HBM-bound: 384,000 moved bytes per frame without the label
                image) -- the convolutions stopped being the largest kernel when the fp16x2 mode halved their products and the
                stage-2 blocks became one launch each.  Every other kernel above a tenth of the step stands under
                roofline.kernels: conv_stage3 (conv_bf3w_kernel<..., 2, ...>), conv_block (conv_block32_kernel), conv_stage2
                (conv_bf3w_kernel<..., 1, ...>), each with algorithmic FLOPs (2*M*N*K of the float32 convolution) and bytes of its
                launches / their HIP-event time on the handle's stream, against both roofs (`bound` = the nearer one): HBM 8 TB/s;
                the dense 16-bit MFMA peak / products per float32 multiply-add (3 in fp16x2 / bf16x2, 6 in bf16x3), the fp32-MFMA
                dense peak with --cnn-math f32.  --stage track: cpx_frame_kernel, 614,400 algorithmic bytes per frame.
  roofline_conv the largest convolution kernel of the step, the same object as its entry under roofline.kernels.
  roofline_track  (e2e) the same HBM accounting for cpx_frame_kernel inside the same run.
  cpu_baseline  the oracle chain ("port": NumPy tracker + NumPy crop/tile + PyTorch-CPU forward, 1 core)
                timed on a bounded sample of the same workload on this host (rank 0, N = 1 only);
  cpu_baseline_all_cores  the same chain on one worker process per CPU this process may use (the scheduler affinity,
                capped by the cgroup's CPU quota: 16 on the GPU boxes of this pool although 256 are visible).  Both run
                before this process initialises the GPU.

--gpus N > 1 without WORLD_SIZE in the environment: bench.py starts the N ranks itself (a child
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py ...`, before anything touches the GPU) and
relays rank 0's line; WORLD_SIZE set and != N is an error.  For N > 1 the line also carries "config4": the strong-scaling
workload below at a reduced size, in the same processes.

  from_files    (default run, N = 1) the file-fed form of the same path: synthetic CPTV byte strings in host memory ->
                upload -> gzip inflate + section index + frame decode on the device -> track -> segments -> crop/tile +
                network -> thumbnails -> metadata JSON text per recording (cpx.track.bulk.run_files_bulk); frames/s and
                the split of the wall time.

--config4: BASELINE configs[3] -- 10,000 seeded clips of 90-540 frames, sharded over the
ranks by greedy longest-processing-time on the frame counts, processed in device batches, one all_gather per step of
[clip_id, track_id, 17 x f32] records; value = all frames x steps / max-over-ranks time ("scaling": "strong").
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

ALGO_BYTES_PER_FRAME = 614400  # 32 B / pixel at 160x120 (labels + filtered written)
LABEL_BYTES_PER_FRAME = 76800  # the int32 label image, not consumed by the classifier
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32-input MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA
BF16X3_PRODUCTS = 6            # bf16 MFMA products per float32 multiply-add (csrc/cpx_cnn_bf3.hip)
# 16-bit MFMA products per float32 multiply-add in the stride-1 layers of stages 2-4, by math mode (include/cpx.h);
# the fp16 MFMA forms run at the bf16 forms' rate (MI355X_MICROARCH.md, matrix table): same 2500 TFLOP/s dense peak
PRODUCTS = {"bf16x3": 6, "bf16x2": 3, "fp16x2": 3}
DEFAULT_CNN_MATH = "fp16x2"
MATH_DTYPE = {
    "fp16x2": "stride-1 3x3 convs of stages 2-4 (+ the stride-2 one): two fp16 planes per operand rounded to nearest "
              "(11 + 11 significand bits), power-of-two range scaling, 3 fp16 MFMAs per K step, device-side rerun in the "
              "exact split if an activation leaves fp16's range; the other 3x3 convs: exact 3-way bf16 split; f32 accumulate "
              "-- float32-class: against a float64 convolution at the exact split's and the fp32 MFMA's error level "
              "(tests/test_cnn_gpu.py::test_bf16x3_is_f32_accurate)",
    "bf16x3": "3x3 convs: exact 3-way bf16 operand split, 6 bf16 MFMAs per K step, f32 accumulate",
    "bf16x2": "stride-1 3x3 convs of stages 2-4: two bf16 planes per operand rounded to nearest, 3 bf16 MFMAs per K step; "
              "the other 3x3 convs: exact 3-way split; f32 accumulate",
    "f32": "f32 MFMA"}
N_LABELS = 17


def conv_layer_table(conv, math, steps):
    """Every convolution shape of the step (key = Cin_g * 10000 + Cout_g * 10 + stride, + 5 for 1x1) with the matrix pipe
    it runs on in this math mode, its float32-equivalent TFLOP/s and the fraction of THAT pipe's dense peak -- so that
    the layers on the slow pipe (the stride-3 convolution, the 1x1 shortcuts when they are launches of their own) do not
    hide inside an average."""
    out = {}
    for k, (n, ms, fl) in sorted(conv.items()):
        if ms <= 0:
            continue
        cin_g, cout_g, st = k // 10000, (k % 10000) // 10, (k % 10) % 5
        one = (k % 10) >= 5
        block = (k % 10) % 5 == 4  # conv_block32_kernel: a residual block's two stride-1 3x3 convolutions in one launch (fp16x2)
        if cin_g == 1:
            pipe, peak = "vector (direct kernel: 8 B in, 64 B out per pixel -- HBM-bound)", None
        elif block:
            pipe, peak = "fp16 MFMA, 3 products", MFMA_BF16_PEAK_TFLOPS / 3
        elif math == "f32" or one or (st == 3 and math != "fp16x2"):  # (fp16x2: the stride-3 layer runs conv_rw_kernel<3, 2, 4>)
            pipe, peak = "fp32 MFMA", MFMA_F32_PEAK_TFLOPS
        elif cin_g == 8:
            pipe, peak = "bf16 MFMA, 6 products (exact split)", MFMA_BF16_PEAK_TFLOPS / 6
        else:
            pr = PRODUCTS[math]
            pipe, peak = "%s MFMA, %d products" % ("fp16" if math == "fp16x2" else "bf16", pr), MFMA_BF16_PEAK_TFLOPS / pr
        tf = fl / (ms / 1e3) / 1e12
        if block:
            st = 1
        out[str(k)] = {"channels_per_group": [cin_g, cout_g], "stride": st,
                       "kernel": "3x3 + 3x3 (a residual block in one launch)" if block else "1x1" if one else "3x3",
                       "launches_per_step": n // max(steps, 1), "ms_per_step": round(ms / steps, 2),
                       "tflops_f32_equivalent": round(tf, 2), "pipe": pipe,
                       "frac_of_pipe_peak": round(tf / peak, 4) if peak else None}
    return out


def bf3w_kernel_name(cout_g, math):
    """The instantiation of conv_bf3w_kernel a stride-1 layer with 32 / 64 channels per group runs, as the rocprofv3
    kernel trace prints it -- the rule of csrc/cpx_cnn_bf3.hip:launch_bf3w_t: <WALK, LDSBN, NH, NG, PL, H, PERSIST> with
    NH = 2 for 64 columns per group, PL planes per operand, H = fp16 planes, LDSBN = NH > 1 or PL == 2."""
    nh = 2 if cout_g == 64 else 1
    pl = 2 if PRODUCTS[math] == 3 else 3
    b = lambda v: "true" if v else "false"
    return "conv_bf3w_kernel<false, %s, %d, 1, %d, %s, false>" % (b(nh > 1 or pl == 2), nh, pl, b(math == "fp16x2"))


def synthetic_network_weights(torch, wr, eng, frames, offs, meta, outputs, frame_size, n_clips=96, n_cal=48, seed=0):
    """Seeded random WR-ResNet-22-4 kernels with BatchNorm statistics fitted to THIS workload's own network inputs (the
    crops of the first clips), as training would have left them: activations and logits are O(1) as in any trained
    model (plain random statistics give logits beyond 100 and activations no BatchNorm would let through).
    cpx.ml_tools.wrresnet.calibrate_bn_device runs the calibration through the HIP convolutions; set-up, untimed."""
    from cpx.pipeline import BatchPipeline

    w = wr.random_weights(N_LABELS, seed=seed)
    nb = min(n_clips, len(offs) - 1)
    T = int(offs[1] - offs[0])
    pipe0 = BatchPipeline(eng, None, n_labels=N_LABELS, fp_index=4, cnn_chunk=n_cal, frame_size=frame_size)
    r = pipe0.run(frames, offs[: nb + 1], meta[: int(offs[nb])], outputs=outputs, keep_samples=True)
    if r.samples_dev is None or int(r.samples_dev.shape[0]) < 4:
        return w
    n_all = int(r.samples_dev.shape[0])   # evenly spread over every clip's segments, not the first clips' only
    pick = torch.linspace(0, n_all - 1, min(n_cal, n_all), device=r.samples_dev.device).round().long()
    x = r.samples_dev[pick].contiguous()
    prev = eng.get_cnn_math()
    eng.set_cnn_math("bf16x3")
    try:
        w = wr.calibrate_bn_device(eng, w, x, N_LABELS)
    finally:
        eng.set_cnn_math(prev)
    del r, x
    torch.cuda.empty_cache()
    return w


def usable_cpus():
    """CPUs this process can actually run on: the scheduler affinity, capped by the cgroup v2 / v1 CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = int(fh.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def spawn_ranks(n, argv):
    """--gpus n without a launcher: start the n ranks as a child torch.distributed.run (never exec: this process may
    not be replaced), relay the child's output, return its exit code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# keys every default (e2e) line carries, whatever N: the driver compares the N = 1 line of the scaling run with the
# default run's; --dry-launch emits the same skeleton (values None) so that a CPU host can check the assembly
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "config", "roofline", "roofline_conv", "roofline_track", "cnn", "per_rank")
MULTI_RANK_KEYS = ("config4", "from_files")


def per_rank_summary(times_s, steps, loads=None):
    """What attributes a scaling result: every rank's own ms per step (before the max), their spread, and the load
    imbalance of the partition (max over mean of the ranks' frame counts; 1.0 for the weak-scaling leg by construction)."""
    ms = [round(t / max(steps, 1) * 1e3, 3) for t in times_s]
    out = {"ms_per_step": ms, "ms_per_step_min": min(ms), "ms_per_step_max": max(ms),
           "slowest_over_fastest": round(max(ms) / max(min(ms), 1e-9), 4)}
    if loads:
        out["frames_per_rank"] = [int(v) for v in loads]
        out["imbalance"] = round(max(loads) / (sum(loads) / len(loads)), 4)
    return out


def dry_launch(args, numa=None):
    """--dry-launch: the ranks rendezvous over gloo and run the HOST side of a multi-GPU bench run -- the NUMA pinning,
    the LPT partition of configs[3], one gather_records exchange of records of its width, the per-rank from_files
    aggregation (all_gather_object), the per-rank timing summary -- with the device work left out; rank 0 prints the line
    skeleton.  The launch path of --gpus N checked without a GPU (tests/test_bench_launch_cpu.py spawns 2 and 8 ranks)."""
    import torch
    import torch.distributed as dist

    from cpx.sharding import gather_records, partition_clips

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    seen = [rank]
    T = args.frames
    lengths = config4_lengths(10000)
    shards = partition_clips(lengths, world)
    loads = [int(lengths[s].sum()) for s in shards]
    mine = shards[rank]
    rec = torch.empty((len(mine), 2 + N_LABELS), dtype=torch.int32)
    rec[:, 0] = torch.tensor(mine, dtype=torch.int32)
    rec[:, 1:] = rank
    ff = {"files": max(512, args.from_files // world) if world > 1 else args.from_files, "frames": 270 * 512,
          "seconds": 1.0 + 0.05 * rank, "frames_per_s": 270 * 512 / (1.0 + 0.05 * rank), "split_s": {}, "what": "dry launch",
          "roofline_inflate": {"frac": None}, "numa": numa}
    times = [1.0 + 0.01 * rank]
    gathered = rec
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = rank + 1
        dist.all_reduce(t)
        seen = [int(v) - 1 for v in t]
        # capacity from the plan: the largest shard's clip count (one record per clip here)
        gathered = gather_records(rec, dist, capacity=max(len(sh) for sh in shards)).records()
        tt = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tt, torch.tensor([times[0]], dtype=torch.float64))
        times = [float(v.item()) for v in tt]
        per_rank = [None] * world
        dist.all_gather_object(per_rank, ff)
        dist.barrier()
        dist.destroy_process_group()
    else:
        per_rank = [ff]
    if rank == 0:
        line = {k: None for k in LINE_KEYS}
        line.update({"metric": "CPTV frames/s end-to-end (track+classify) at 160x120", "unit": "frames/s",
                     "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
                     "scaling": "weak", "data": "synthetic", "dry_launch": True, "ranks": seen,
                     "config": {"workload": "dry launch: host logic only", "frames_per_clip": T},
                     "per_rank": per_rank_summary(times, 1, [4096 * T] * world)})
        if world > 1:
            line["config4"] = {"scaling": "strong", "clips": 10000, "frames_total": int(lengths.sum()),
                               "frames_per_rank": loads, "imbalance": round(max(loads) / (sum(loads) / len(loads)), 4),
                               "records_gathered": int(gathered.shape[0]), "record_width": int(gathered.shape[1])}
            line["from_files"] = aggregate_from_files(per_rank, ff["files"], usable_cpus())
        print(json.dumps(line), flush=True)


def synth_on_device(torch, device, n_clips, n_frames, seed, h=120, w=160, chunk=64):
    """Synthetic clips generated on the GPU (same recipe as cpx.synth): smooth background 2900 +- 40,
    sensor noise N(0, 4), up to 3 warm Gaussian blobs on a random walk.
    -> int16-bit-pattern uint16 tensor [n_clips*n_frames, h, w]."""
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


def _oracle_clip(job):
    """One clip through the oracle chain (a worker of cpu_baseline; imports happen here so that a spawned process
    needs nothing from the parent).  job = (stage, seed, n_frames, frame_size, weights or None) -> (frames, samples)."""
    import numpy as np
    import torch

    import classify_oracle as co
    import cnn_oracle as cnn
    import track_oracle as to
    from cpx import synth

    stage, seed, n_frames, frame_size, weights = job
    torch.set_num_threads(1)
    clip = synth.make_clip(np.random.default_rng(seed), n_frames)
    cfg = to.OracleConfig("lepton3")
    H, W = clip.shape[1:]
    if stage == "track":
        to.track_clip(clip, cfg=cfg, keep=True, do_tracking="regions")
        return n_frames, 0
    out = to.track_clip(clip, cfg=cfg, keep=True)
    fr = out["frames"]
    n_samples = 0
    for t in out["tracks"]:
        usable = [r.frame_number for r in t.bounds if not r.blank and r.mass > 0 and r.width > 0 and r.height > 0]
        if not usable:
            continue
        nseg = max(1, (len(usable) + 12) // 25)
        segs = []
        for s in range(nseg):
            run = usable[25 * s: 25 * s + 25]
            segs.append(np.array([run[(j * len(run)) // 25] for j in range(25)]))
        by_frame = {r.frame_number: r for r in t.bounds}
        x, _ = co.preprocess_segments(lambda q: clip[q], lambda q: fr[q]["filtered"].astype(np.float64), by_frame,
                                      t.bounds, segs, frame_size, (1, 1, W - 2, H - 2))
        _, probs = cnn.forward(weights, x)
        co.classified_track(probs, prediction_frames=segs)
        n_samples += nseg
    return n_frames, n_samples


def cpu_baseline(stage, lengths, seed, weights, frame_size=32, cores=1):
    """The oracle chain over a bounded sample of the same workload: clips of `lengths` frames, seeded like the
    device-side recipe, on `cores` worker processes (1 = in this process).  Runs BEFORE this process touches the GPU
    (spawned workers and HIP do not mix)."""
    jobs = [(stage, seed + i, int(n), frame_size, weights) for i, n in enumerate(lengths)]
    if cores <= 1:
        _oracle_warm(0)
        t0 = time.perf_counter()
        res = [_oracle_clip(j) for j in jobs]
    else:
        import multiprocessing as mp

        with mp.get_context("spawn").Pool(cores) as pool:
            pool.map(_oracle_warm, range(cores))      # interpreter start + imports are not the baseline
            t0 = time.perf_counter()
            res = pool.map(_oracle_clip, jobs, chunksize=1)
    dt = time.perf_counter() - t0
    frames, n_samples = sum(r[0] for r in res), sum(r[1] for r in res)
    what = ("pixel stage + regions (NumPy)" if stage == "track" else
            "track + classify (NumPy tracker, NumPy crop/tile, PyTorch-CPU fp32 forward; %d samples)" % n_samples)
    return {"value": round(frames / dt, 1), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic clips, %d frames, oracle %s, %.1f s" % (len(jobs), frames, what, dt)}


def _oracle_warm(_):
    import torch  # noqa: F401

    import classify_oracle  # noqa: F401
    import cnn_oracle  # noqa: F401
    import track_oracle  # noqa: F401
    return 0


# algorithmic FLOPs of one stage-2 3x3 convolution per sample as the library counts them (cpx_conv_timing_report):
# 2 * Ho*Wo * Cout * Cin/groups * k*k = 2 * 160*160 * 64 * 32 * 9
STAGE2_CONV_FLOPS_PER_SAMPLE = 2.0 * 160 * 160 * 64 * 32 * 9


PMC_SUMMARY = "profiles/r06_e2e_pmc.json"
PMC_NOTE = ("NOT measured in this run: bytes per unit from the committed rocprofv3 PMC summary %s (separate FETCH_SIZE / "
            "WRITE_SIZE passes over `python3 bench.py --steps 1 --warmup 0` at the bench's own 4,096 clips, FETCH_SIZE doubled per the gfx950 note of "
            "MI355X_MICROARCH.md), rescaled to this run's units per launch" % PMC_SUMMARY)


def pmc_traffic(section, units_per_launch):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (PMC_SUMMARY: separate FETCH_SIZE /
    WRITE_SIZE passes over this same bench command, gfx950 correction applied), scaled to this run's launch size.
    PMC collection needs the profiler, so it cannot be taken live here; None when the summary is absent.
    The bench line says so in roofline.traffic_source."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), *PMC_SUMMARY.split("/"))
    try:
        with open(path) as fh:
            sec = json.load(fh)[section]
        return round(sec["hbm_traffic_bytes_per_launch"] / sec["units_per_launch"] * units_per_launch, 1)
    except (OSError, KeyError, ValueError):
        return None


def aggregate_from_files(per_rank, share, host_cpus):
    """rank 0's `from_files` object for N > 1: the per-rank results of bench_from_files (gathered with
    all_gather_object; a rank that failed contributes {"error": ...}) -> per-rank split + the aggregate rate, frames of
    all ranks over the slowest rank's time (tests/test_bench_launch_cpu.py)."""
    world = len(per_rank)
    ok = [r for r in per_rank if r and "error" not in r]
    agg = {"what": "every rank: %d synthetic recordings (its share) through run_files_bulk on its own GPU, host stages of "
                   "all ranks on the node's CPUs at the same time" % share,
           "ranks": world, "host_cpus_usable": host_cpus,
           "per_rank": [{"rank": i, "numa": (r or {}).get("numa"),
                         **({k: r[k] for k in ("files", "frames", "seconds", "frames_per_s", "split_s")}
                            if r and "error" not in r else {"error": (r or {}).get("error", "no result")})}
                        for i, r in enumerate(per_rank)]}
    if ok:
        slowest = max(r["seconds"] for r in ok)
        agg["frames"] = int(sum(r["frames"] for r in ok))
        agg["seconds_slowest_rank"] = round(slowest, 3)
        agg["frames_per_s"] = round(agg["frames"] / slowest, 1)
        agg["roofline_inflate"] = ok[0].get("roofline_inflate")
    return agg


INFLATE_COUNTERS = "profiles/r04_inflate_sq_counters.json"


def inflate_roofline():
    """The decode kernel's place against the roof that applies to it: one wavefront per recording
    walks a serial DEFLATE symbol chain out of scalar instructions, and a CU issues at most one scalar instruction per
    cycle -- no HBM or matrix roof comes near (34 GB/s of output).  From the committed rocprofv3 SQ counters (PMC needs
    the profiler: not taken live): scalar instructions per output byte x bytes / (256 CUs x cycles of the launch)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), *INFLATE_COUNTERS.split("/"))
    try:
        with open(path) as fh:
            prof = json.load(fh)
    except (OSError, ValueError):
        return None
    out = {"kernel": "cpx_cptv_inflate_kernel (one wavefront per recording, 16 per CU)",
           "bound": "scalar instruction issue: one SALU instruction per cycle and CU",
           "source": "NOT measured in this run: %s (scratch/pmc_inflate.sh, 4096 recordings per launch)" % INFLATE_COUNTERS}
    for key in ("fixture_recordings", "synthetic_level6_270_frames"):
        d = prof.get(key, {}).get("derived")
        if d:
            out[key] = {"frac": d["scalar_issue_frac"], "scalar_instructions_per_output_byte": d["salu_per_output_byte"],
                        "vector_instructions_per_output_byte": d["valu_per_output_byte"], "output_GBps": d["GBps_out"],
                        "kernel_ms_per_4096_recordings": d["kernel_ms"], "clock_GHz": d["clock_GHz"],
                        "wave_time_parked_on_waitcnt": d["wave_time_parked"]}
    out["frac"] = out.get("fixture_recordings", {}).get("frac")
    return out


def bench_ir(args, torch, np, dist, device, rank, world, local_rank):
    """--stage ir: S synthetic 640x480 uint8 videos per GPU through the WHOLE IR tracker as one device batch
    (IRTrackExtractor.parse_frames_batch: cpx_mog2_apply, cpx_ir_detect, cpx_ir_merge per frame step in lockstep, one
    cpx_associate_batch, then the host's Track / Region objects, trap replay and track filtering; frames resident in
    HBM).  value = frames/s of that; "detect_stage" = the device front half alone (MOG2 + detection), what this stage
    timed up to round 2.  roofline: the MOG2 kernel, HBM-bound, 124 algorithmic bytes per pixel (61 B of mixture state
    read and written, 1 B frame, 1 B mask)."""
    import ctypes as C

    from cpx.engine import TrackEngine
    from cpx.track.irdetect import MOG2Background

    H, W = 480, 640
    S = args.clips or 64
    T = min(args.frames, 32)
    eng = TrackEngine(model="lepton3", device=local_rank)
    g = torch.Generator(device=device)
    g.manual_seed(99 + rank)
    scene = torch.randint(40, 200, (S, 1, H, W), generator=g, device=device, dtype=torch.int16)
    video = (scene + torch.randint(-2, 3, (S, T, H, W), generator=g, device=device, dtype=torch.int16))
    for t in range(T):  # a bright block crossing every scene
        x0 = (11 * t) % (W - 60)
        video[:, t, 150:200, x0:x0 + 60] = 235
    video = video.clamp_(0, 255).to(torch.uint8).permute(1, 0, 2, 3).contiguous()  # [T, S, H, W]
    bg = MOG2Background(eng, W, H, n_streams=S)
    mask = torch.empty((S, H, W), dtype=torch.uint8, device=device)
    cap = 1024
    comps = torch.empty((S, cap, 8), dtype=torch.int32, device=device)
    counts = torch.zeros((T, S), dtype=torch.int32, device=device)
    status = torch.zeros((T, S), dtype=torch.int32, device=device)

    from cpx.config import Config
    from cpx.track.clip import Clip
    from cpx.track.irtrackextractor import IRTrackExtractor

    tracker = IRTrackExtractor(Config.get_defaults().tracking, device=local_rank)
    n_tracks = [0]

    def step():
        clips = []
        for v in range(S):
            clip = Clip(tracker.config, "ir-%d.mp4" % v, type="IR")
            clip.frames_per_second = 10
            clips.append(clip)
        tracker.parse_frames_batch(clips, video)
        n_tracks[0] = sum(len(c.tracks) for c in clips)

    def front_step():
        for t in range(T):
            rc = eng.lib.cpx_mog2_apply(bg._m, C.c_void_p(video[t].data_ptr()), -1.0, C.c_void_p(mask.data_ptr()))
            assert rc == 0, eng._err()
            rc = eng.lib.cpx_ir_detect(eng.h, C.c_void_p(mask.data_ptr()), S, W, H, 0, cap, C.c_void_p(comps.data_ptr()),
                                       C.c_void_p(counts[t].data_ptr()), C.c_void_p(status[t].data_ptr()), None)
            assert rc == 0, eng._err()
        eng.synchronize()

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    torch.cuda.synchronize(device)
    for _ in range(max(args.warmup, 1)):
        step()
    # the MOG2 kernel alone, timed on the handle's stream by wall clock around back-to-back launches
    eng.synchronize()
    t0 = time.perf_counter()
    reps = 4 * T
    for r in range(reps):
        eng.lib.cpx_mog2_apply(bg._m, C.c_void_p(video[r % T].data_ptr()), -1.0, C.c_void_p(mask.data_ptr()))
    eng.synchronize()
    mog2_s = (time.perf_counter() - t0) / reps
    front_step()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(args.steps, 1)):
        front_step()
    front_s = (time.perf_counter() - t0) / max(args.steps, 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert int(status.max().item()) == 0 and int(counts[T // 2:].max().item()) >= 1
    if rank == 0:
        algo = S * H * W * 124.0
        gbs = algo / mog2_s / 1e9
        line = {"metric": "IR 640x480 frames/s through the IR tracker (background model, detection, merge, association, "
                          "tracks: the IR half of configs[4])",
                "value": round(world * S * T * args.steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "u8 frames / masks, f32 mixture state (MOG2), 1-bit rows + i32 statistics (detection)",
                "data": "synthetic",
                "config": {"workload": "synthetic 640x480 uint8 videos: IRTrackExtractor.parse_frames_batch, "
                                       "streams in lockstep", "streams_per_gpu": S, "frames_per_stream": T,
                           "tracks_found": n_tracks[0]},
                "detect_stage": {"frames_per_s": round(S * T / front_s, 1), "ms_per_step": round(front_s * 1e3, 3),
                                 "what": "cpx_mog2_apply + cpx_ir_detect alone (rank 0)"},
                "roofline": {"kernel": "cpx_mog2_apply_kernel", "bound": "hbm", "achieved": round(gbs, 1),
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                             "avg_launch_us": round(mog2_s * 1e6, 2), "algorithmic_bytes_per_launch": algo}}
        if world == 1 and args.cpu_clips != 0:
            import ir_oracle as iro
            import mog2_oracle as mo

            om = mo.MOG2(W, H)
            host = video[:, 0].cpu().numpy()
            t0 = time.perf_counter()
            nfr = min(T, 24)
            for t in range(nfr):
                iro.detect_objects_ir(om.apply(host[t]), threshold=0)
            dt = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(nfr / dt, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                                    "sample": "%d frames of one stream: oracle MOG2 (C) + detect_objects_ir (NumPy), %.1f s"
                                              % (nfr, dt)}
        print(json.dumps(line), flush=True)
    bg.close()
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_from_files(args, torch, np, local_rank, weights, T, n_files=None, with_fixtures=True, meta_pool=None):
    """The file-fed form of the headline path: `--from-files` synthetic recordings as CPTV byte
    strings in host memory (32 distinct clips of the headline's generator, T frames, gzip level 6, replicated) -> cpx.track.bulk.run_files_bulk
    with a ClipClassifier: upload, gzip inflate + section index + frame decode on the device, track, segments,
    crop/tile + network, thumbnails, metadata JSON text per recording.  One warm-up pass over the same recordings, then a timed one."""
    import tempfile

    from cpx import synth
    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.config import Config
    from cpx.config.config import ModelConfig
    from cpx.cptv import encode_cptv
    from cpx.ml_tools import wrresnet as wr
    from cpx.track.bulk import run_files_bulk

    labels = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid",
              "penguin", "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]
    tmp = tempfile.mkdtemp(prefix="cpx_bench_model_")
    wr.save_model(os.path.join(tmp, "wr"), weights, labels, hyperparams={"frame_size": 32})
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr-bench", "model_file": os.path.join(tmp, "wr.npz")})]
    cfg.classify.meta_to_stdout = False
    t0 = time.perf_counter()
    t_on, ffc = synth.frame_times(T)
    # gzip level 6 = zlib's default, what the fixture recordings' headers say their writer used (XFL 0); level 1 finds
    # three times as many (short) matches in sensor noise, and a match is the decoder's expensive symbol
    # the headline's own clips (synth_on_device: the same generator, so the same tracks and segments per frame as
    # `value` above), written out as recordings
    n_distinct = 32
    device = torch.device("cuda", local_rank)
    clips_host = synth_on_device(torch, device, n_distinct, T, seed=4321).cpu().numpy().view(np.uint16).reshape(n_distinct, T, 120, 160)
    distinct = [encode_cptv(clips_host[i], t_on, ffc, level=6) for i in range(n_distinct)]
    del clips_host
    encode_s = time.perf_counter() - t0
    n = n_files or args.from_files
    batch = 2048   # recordings per decode launch (tracked in groups of 1024)
    cc = ClipClassifier(cfg)

    def overflow_forwards():
        from cpx.track import cliptrackextractor as cte

        return int(sum(e.cnn_overflow_forwards(reset=True) for e in cte._ENGINES.values() if e.h))

    def measure(blobs, names, cc=cc, cfg=cfg):
        torch.cuda.empty_cache()   # (the previous workload's cached blocks have other sizes: start from a clean pool)
        # warm-up = the same pass once: engines, model, and the pinned / device allocators grown to the pipeline's
        # working set (three batches in flight), as in a service that has been running
        run_files_bulk(names, cfg, save_meta=False, want_text=True, device=local_rank, batch_files=batch,
                       clip_classifier=cc, blobs=blobs, meta_pool=meta_pool)
        torch.cuda.synchronize()
        overflow_forwards()
        t0 = time.perf_counter()
        out, tracker = run_files_bulk(names, cfg, save_meta=False, want_text=True, device=local_rank,
                                      batch_files=batch, clip_classifier=cc, blobs=blobs, meta_pool=meta_pool)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        bad = [k for k, v in out.items() if v.startswith("error")]
        assert not bad, bad[:3]
        tracker.timings["fp16_overflow_forwards"] = overflow_forwards()
        return out, tracker.timings, dt

    blobs = [distinct[i % len(distinct)] for i in range(n)]
    names = ["synthetic_%05d.cptv" % i for i in range(n)]
    out, tm, dt = measure(blobs, names)
    # the same path over real recordings: copies of the two fixture clips (reference tests/clips), when they are here
    fixtures = None
    gold = os.path.join(REPO, "tests", "golden")
    if with_fixtures and all(os.path.exists(os.path.join(gold, f + ".cptv")) for f in ("possum", "hedgehog")):
        real = [open(os.path.join(gold, f + ".cptv"), "rb").read() for f in ("possum", "hedgehog")]
        # a network's BatchNorm statistics belong to its data: the synthetic network of this leg gets them from the
        # fixture recordings' own crops (as the headline's got them from the synthetic clips' crops) -- statistics of another
        # distribution put real crops' activations far outside what the layers were normalised for (and, in fp16x2, beyond
        # the range scaling: every such forward then pays the bf16x3 rerun)
        cc_fx, cfg_fx = cc, cfg
        try:
            from cpx.cptv import CptvReader
            from cpx.engine import TrackEngine

            eng_fx = TrackEngine(model="lepton3", device=local_rank, max_frames=512)
            fr, lens, metas = [], [], []
            for name in ("possum", "hedgehog"):
                frames_f = CptvReader(os.path.join(gold, name + ".cptv")).read_all()
                fr.append(np.stack([f.pix for f in frames_f]).astype(np.uint16))
                lens.append(len(frames_f))
                metas.append(eng_fx.make_meta(len(frames_f), [f.time_on for f in frames_f], [f.last_ffc_time for f in frames_f],
                                              [bool(f.background_frame) for f in frames_f]))
            offs_f = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            w_fx = synthetic_network_weights(torch, wr, eng_fx, eng_fx.upload_frames(np.concatenate(fr)), offs_f,
                                             np.concatenate(metas), None, 32, n_clips=2)
            eng_fx.close()
            wr.save_model(os.path.join(tmp, "wr_fx"), w_fx, labels, hyperparams={"frame_size": 32})
            cfg2 = Config.get_defaults()
            cfg2.tracking["thermal"].denoise = False
            cfg2.classify.models = [ModelConfig.load({"id": 1, "name": "wr-bench", "model_file": os.path.join(tmp, "wr_fx.npz")})]
            cfg2.classify.meta_to_stdout = False
            cc_fx, cfg_fx = ClipClassifier(cfg2), cfg2
        except Exception as e:  # noqa: BLE001 -- the leg still runs, on the synthetic clips' statistics
            sys.stderr.write("from_files: fixture calibration failed (%s: %s)\n" % (type(e).__name__, e))
        # eight decode batches of 2048: the three pipeline stages overlap as they do on a large directory (8,192 copies:
        # 297-333 k frames/s, 4,096: 275 k -- the first decode and the last metadata are not overlapped with anything)
        nr = min(2 * n, 16384)
        outr, tmr, dtr = measure([real[i % 2] for i in range(nr)], ["fixture_%05d.cptv" % i for i in range(nr)], cc=cc_fx, cfg=cfg_fx)
        fixtures = {"what": "%d copies of the reference's two fixture recordings (tests/clips/possum.cptv, hedgehog.cptv: "
                            "161 / 120 frames, 1.1 MB each) through the same call" % nr,
                    "files": nr, "frames": int(tmr["frames"]), "seconds": round(dtr, 3),
                    "frames_per_s": round(tmr["frames"] / dtr, 1), "files_per_s": round(nr / dtr, 1),
                    "network": "the same seeded kernels, BatchNorm statistics fitted to these recordings' own crops" if cc_fx is not cc
                               else "the synthetic clips' network",
                    "fp16_overflow_forwards": tmr.get("fp16_overflow_forwards"),
                    "split_s": {"stage_pinned_copy": round(tmr.get("stage_s", 0.0), 3),
                                "upload_inflate_index_unpack": round(tmr["decode_s"], 3),
                                "main_thread_waiting_for_decode": round(tmr.get("wait_decode_s", 0.0), 3),
                                "track_classify_thumbnails_device": round(tmr["device_s"], 3),
                                "metadata_host": round(tmr["host_s"], 3),
                                "metadata_workers": meta_pool.workers if meta_pool is not None else 0,
                                "metadata_submit_main_thread": round(tmr.get("host_submit_s", 0.0), 3),
                                "metadata_collect_main_thread": round(tmr.get("host_collect_s", 0.0), 3)}}
    n_tracks = sum(text.count('"tracking_score"') for text in out.values())
    n_pred = sum(text.count('"all_class_confidences"') for text in out.values())
    return {"what": "%d synthetic recordings (%d frames each: %d distinct clips of the headline's generator, gzip level 6, %.2f MB per file) as byte strings "
                    "in host memory -> upload -> inflate + index + decode on the device -> track -> segments -> "
                    "crop/tile + WR-ResNet -> thumbnails -> metadata JSON text per recording; decode launches of %d recordings, tracking groups of 1024"
                    % (n, T, n_distinct, len(distinct[0]) / 1e6, batch),
            "fixture_recordings": fixtures,
            "roofline_inflate": inflate_roofline(),
            "files": n, "frames": int(tm["frames"]), "seconds": round(dt, 3),
            "frames_per_s": round(tm["frames"] / dt, 1), "files_per_s": round(n / dt, 1),
            "tracks_in_metadata": n_tracks, "tracks_with_predictions": n_pred,
            "fp16_overflow_forwards": tm.get("fp16_overflow_forwards"),
            "metadata_bytes": int(sum(len(v) for v in out.values())),
            "split_s": {"note": "the first two run in a worker thread beside the others (a HIP stream of its own)",
                        "stage_pinned_copy": round(tm.get("stage_s", 0.0), 3),
                        "upload_inflate_index_unpack": round(tm["decode_s"], 3),
                        "main_thread_waiting_for_decode": round(tm.get("wait_decode_s", 0.0), 3),
                        "track_classify_thumbnails_device": round(tm["device_s"], 3),
                        "metadata_host": round(tm["host_s"], 3), "collect": round(tm["write_s"], 3),
                        "metadata_workers": meta_pool.workers if meta_pool is not None else 0,
                        "metadata_submit_main_thread": round(tm.get("host_submit_s", 0.0), 3),
                        "metadata_collect_main_thread": round(tm.get("host_collect_s", 0.0), 3)},
            "encode_synthetic_files_s": round(encode_s, 2), "denoise": False,
            **({"device_split_s": {k: round(v, 3) for k, v in tm["device_split_s"].items()}} if "device_split_s" in tm else {})}


def config4_lengths(n_clips, seed=1234, lo=90, hi=540):
    """Frames per clip of the configs[3] workload: seeded, uniform in [lo, hi] (10 s - 60 s at 9 fps)."""
    import numpy as np

    return np.random.default_rng(seed).integers(lo, hi + 1, size=n_clips).astype(np.int64)


class Config4Workload:
    """BASELINE configs[3] / SURVEY section 8(d) config 4: `n_clips` seeded synthetic clips of varying length, sharded
    over the ranks by greedy longest-processing-time on the frame counts (cpx.sharding.partition_clips), every rank's
    shard processed in device batches that fit HBM (plan_sub_batches; frames resident), and ONE all-gather per step of
    the records [clip_id, track_id, n_labels x f32] (pack_records / gather_records).  A clip's pixels depend only on
    (seed, clip id), never on the rank that owns it."""

    H, W = 120, 160

    def __init__(self, torch, device, local_rank, rank, world, n_clips, seed=1234, lo=90, hi=540,
                 sub_frames=2048 * 270, cnn_chunk=2048, frame_size=32, cnn_math=DEFAULT_CNN_MATH, weights=None, n_labels=N_LABELS):
        import numpy as np

        from cpx.engine import TrackEngine
        from cpx.ml_tools import wrresnet as wr
        from cpx.pipeline import BatchPipeline
        from cpx.sharding import partition_clips, plan_sub_batches

        self.torch, self.np, self.device, self.rank, self.world = torch, np, device, rank, world
        self.seed, self.n_labels = seed, n_labels
        self.lengths = config4_lengths(n_clips, seed, lo, hi)
        self.shards = partition_clips(self.lengths, world)
        self.mine = self.shards[rank]
        self.subs = plan_sub_batches(self.lengths, self.mine, sub_frames)
        self.eng = TrackEngine(width=self.W, height=self.H, model="lepton3", device=local_rank, max_components=64,
                               max_frames=max(int(hi), 45))
        self.eng.set_cnn_math(cnn_math)
        self.batches = []  # (clip ids tensor, frames, offs, meta)
        cap = 0
        for ids in self.subs:
            lens = [int(self.lengths[i]) for i in ids]
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            frames = torch.empty((int(offs[-1]), self.H, self.W), dtype=torch.int16, device=device)
            metas = []
            for k, i in enumerate(ids):
                frames[offs[k]:offs[k + 1]] = self.clip_frames(i)
                metas.append(self.eng.make_meta(lens[k], [100000 + 114 * q for q in range(lens[k])], [40000] * lens[k]))
            self.batches.append((torch.tensor(ids, dtype=torch.int32, device=device), frames, offs, np.concatenate(metas)))
            cap = max(cap, int(offs[-1]))
        self.cap_frames = cap
        self.comps = torch.empty(max(cap, 1) * 64 * 8, dtype=torch.int32, device=device)
        self.info = torch.empty(max(cap, 1) * 20, dtype=torch.int32, device=device)
        self.filt = torch.empty((max(cap, 1), self.H, self.W), dtype=torch.float32, device=device)
        if weights is None and self.batches:  # BatchNorm statistics fitted to the first batch's own crops (untimed set-up)
            _, fr0, offs0, meta0 = self.batches[0]
            tot0 = int(offs0[-1])
            weights = synthetic_network_weights(torch, wr, self.eng, fr0, offs0, meta0,
                                                (self.comps[: tot0 * 64 * 8], self.info[: tot0 * 20], None, self.filt[:tot0], None),
                                                frame_size)
        self.weights = weights if weights is not None else wr.random_weights(n_labels, seed=0)
        self.net = wr.WRResNetDevice(self.eng, self.weights, n_labels)
        self.pipe = BatchPipeline(self.eng, self.net, n_labels=n_labels, fp_index=4, cnn_chunk=cnn_chunk,
                                  frame_size=frame_size)
        self.frames_local = int(sum(int(self.lengths[i]) for i in self.mine))
        self.frames_total = int(self.lengths.sum())
        # the collective's slab size, from the plan alone (identical on every rank): the largest shard's clips x the
        # tracks a clip may keep
        self.record_capacity = max(len(sh) for sh in self.shards) * self.pipe.tp.max_tracks
        self.last = None

    def clip_frames(self, clip_id):
        """The clip's frames on the device (uint16 bits in int16), a function of (seed, clip id) only."""
        return synth_on_device(self.torch, self.device, 1, int(self.lengths[clip_id]), seed=self.seed * 1000003 + int(clip_id),
                               h=self.H, w=self.W, chunk=1)

    def step(self, dist=None):
        """One pass over this rank's shard + the all-gather.  -> records of ALL ranks, int32 [n_tracks, 2 + n_labels]."""
        from cpx.sharding import gather_records, pack_records

        t = self.torch
        recs, results = [], []
        for ids_dev, frames, offs, meta in self.batches:
            total = int(offs[-1])
            outputs = (self.comps[: total * 64 * 8], self.info[: total * 20], None, self.filt[:total], None)
            res = self.pipe.run(frames, offs, meta, outputs=outputs)
            results.append(res)
            if res.n_tracks and res.scores is not None:
                recs.append(pack_records(ids_dev[res.track_clip[:, 0].long()], res.track_clip[:, 1], res.scores))
        rec = t.cat(recs) if recs else t.empty((0, 2 + self.n_labels), dtype=t.int32, device=self.device)
        self.last = results
        t.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        out = gather_records(rec, dist, capacity=self.record_capacity)
        t.cuda.synchronize(self.device)
        self.last_gather_s = time.perf_counter() - t0
        return out

    def close(self):
        self.net.close()
        self.eng.close()


def bench_config4(args, torch, np, dist, device, rank, world, local_rank, cpu, as_sub_object=False, n_clips=None,
                  steps=None):
    """--config4: see Config4Workload.  value = frames of ALL clips x steps / max-over-ranks time (strong scaling: the
    clip set is fixed, ranks share it).  as_sub_object: return the line (rank 0; None elsewhere) instead of printing it
    and keep the process group -- the "config4" entry of a multi-GPU default run."""
    n_clips = n_clips or args.clips or 10000
    steps = steps or args.steps
    # every rank needs its shard resident: shrink the clip set when one GPU cannot hold its share
    free, _ = torch.cuda.mem_get_info(device)
    budget = 0.42 * free
    while n_clips > 64 and config4_lengths(n_clips).sum() / world * 38400 > budget:
        n_clips //= 2
    wl = Config4Workload(torch, device, local_rank, rank, world, n_clips, sub_frames=args.sub_frames,
                         cnn_chunk=args.cnn_chunk, frame_size=args.frame_size, cnn_math=args.cnn_math)

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        wl.step(dist)
    fence()
    t0 = time.perf_counter()
    gather_s = 0.0
    for _ in range(steps):
        gathered = wl.step(dist)
        gather_s += wl.last_gather_s
    fence()
    elapsed = time.perf_counter() - t0
    rank_times = [elapsed]
    if dist is not None:
        every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(every, torch.tensor([elapsed], dtype=torch.float64, device=device))
        rank_times = [float(v.item()) for v in every]
        elapsed = max(rank_times)
    for r in wl.last:
        r.track.check()
        r.assoc.check()
    gathered = gathered.records()  # outside the timed region: the step itself never reads the device
    line = None
    if rank == 0:
        loads = [int(wl.lengths[s].sum()) for s in wl.shards]
        line = {"metric": "CPTV frames/s end-to-end (track+classify) at 160x120",
                "value": round(wl.frames_total * steps / elapsed, 1), "unit": "frames/s", "n_gpus": world,
                "steps": steps, "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "u16/i32 track stage (f32 normalise, f64 gates); f32 crop/tile; f32 CNN (%s)" % MATH_DTYPE[args.cnn_math],
                "data": "synthetic",
                "config": {"workload": "BASELINE configs[3]: %d seeded synthetic clips of 90-540 frames, LPT-sharded by frame "
                                       "count over the ranks, device batches of <= %d frames, track -> 25-frame segments -> "
                                       "crop/tile + WR-ResNet-22-4, all_gather of [clip_id, track_id, %d x f32]"
                                       % (n_clips, args.sub_frames, N_LABELS),
                           "clips": int(n_clips), "frames_total": wl.frames_total, "frames_per_rank": loads,
                           "imbalance": round(max(loads) / (sum(loads) / len(loads)), 4),
                           "device_batches_rank0": len(wl.subs), "records_gathered": int(gathered.shape[0]),
                           "record_width": int(gathered.shape[1]), "frame_size": args.frame_size,
                           "cnn_chunk": args.cnn_chunk, "n_labels": N_LABELS,
                           "gather_ms_per_step_rank0": round(gather_s / steps * 1e3, 3)},
                "per_rank": per_rank_summary(rank_times, steps, loads)}
        line.update(cpu)
        if not as_sub_object:
            print(json.dumps(line), flush=True)
    wl.close()
    if as_sub_object:
        return line
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="GPUs of this node (default: WORLD_SIZE when a launcher started this process, else 1)")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default 4096, reduced if HBM is short)")
    ap.add_argument("--frames", type=int, default=270)
    ap.add_argument("--stage", choices=("e2e", "track", "ir"), default="e2e",
                    help="e2e: track + classify (the BASELINE metric); track: configs[1] kernels only; ir: the 640x480 "
                         "IR front half of configs[4] (MOG2 background model + detection stage, SURVEY section 8 f4)")
    ap.add_argument("--cpu-clips", type=int, default=-1, help="clips in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cnn-chunk", type=int, default=0,
                    help="samples per CNN forward (default 2048 at frame size 32 = 54 GB of activations, 512 at 64)")
    ap.add_argument("--frame-size", type=int, default=32, choices=(32, 64),
                    help="side of one tile of the 5x5 network input (SURVEY 8(d) config 3 asks for 32 and 64)")
    ap.add_argument("--cnn-math", choices=("fp16x2", "bf16x3", "f32", "bf16x2"), default=DEFAULT_CNN_MATH,
                    help="how the 3x3 stride-1 convolutions multiply their float32 operands (include/cpx.h: cpx_set_cnn_math)")
    ap.add_argument("--sub-batches", type=int, default=1,
                    help="groups of clips per step: the track stage of group k+1 is issued on a second stream beside the "
                         "network of group k (measured: no gain on MI355X, see DESIGN.md section 6; 1 = off)")
    ap.add_argument("--denoise", action="store_true",
                    help="tracking.denoise = true (the reference's default: NLM kernel between normalise and blur)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the default_config (denoise on) and fs64 measurements taken after the timed region")
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE configs[3]: --clips (default 10000) seeded clips of 90-540 frames, LPT-sharded over the "
                         "ranks, processed in device batches, all_gather of [clip_id, track_id, 17 x f32] (strong scaling)")
    ap.add_argument("--sub-frames", type=int, default=2048 * 270,
                    help="--config4: frames per device batch (their per-frame outputs must fit HBM beside the clips)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="only launch the ranks (gloo rendezvous, no GPU work): checks the --gpus N path on a CPU host")
    ap.add_argument("--meta-workers", type=int, default=2,
                    help="metadata worker processes of the from_files leg (cpx.track.bulk.MetaPool: what a directory run "
                         "of this size starts; 0 = the stage stays in the process)")
    ap.add_argument("--from-files", type=int, default=8192,
                    help="recordings of the from_files measurement of the default run (0 = skip)")
    args = ap.parse_args()

    # ---- --gpus N: this process becomes the launcher of N ranks unless a launcher already started us ----
    # A launcher (torch.distributed.run) exports RANK and WORLD_SIZE together; a stray WORLD_SIZE alone is not one.
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not launched:
        os.environ.pop("WORLD_SIZE", None)
        if args.gpus is None:
            args.gpus = 1
        if args.gpus > 1:
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    elif args.gpus is None:
        args.gpus = int(os.environ["WORLD_SIZE"])  # `torchrun --nproc-per-node 8 bench.py`: the launcher's world is the answer
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:  # only an explicit, contradicting --gpus is an error
        sys.stderr.write("bench.py: --gpus %d but the launcher started %s ranks (WORLD_SIZE)\n"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
        raise SystemExit(2)
    # a launched rank keeps its host threads on the CPUs of its GPU's NUMA node (cpx.sharding.pin_to_gpu_numa: sysfs only,
    # before numpy / torch start their thread pools -- threads inherit the affinity; no numactl wrapper, no re-exec)
    numa = None
    if launched and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        from cpx.sharding import pin_to_gpu_numa

        numa = pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")))
    if args.dry_launch:
        return dry_launch(args, numa)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.cnn_chunk <= 0:
        args.cnn_chunk = 2048 if args.frame_size == 32 else 512
    # ---- CPU baselines first: worker processes are spawned while this process has not touched the GPU yet ----
    cpu = {}
    if world == 1 and rank == 0 and args.cpu_clips != 0 and args.stage != "ir":
        from cpx.ml_tools import wrresnet as wr0

        w0 = wr0.random_weights(N_LABELS, seed=0)
        e2e0 = args.stage == "e2e"
        n1 = args.cpu_clips if args.cpu_clips > 0 else (20 if e2e0 else 16)  # ~15 s of single-core work
        lens = [int(v) for v in config4_lengths(10000)[:n1]] if args.config4 else [args.frames] * n1
        cpu["cpu_baseline"] = cpu_baseline(args.stage, lens, 1234, w0, args.frame_size, cores=1)
        host = os.cpu_count() or 1
        cores = usable_cpus()  # affinity and cgroup quota: the CPUs this process really has
        if cores > 1:
            many = [lens[i % len(lens)] for i in range(max(2 * cores, n1))]
            allc = cpu_baseline(args.stage, many, 1234, w0, args.frame_size, cores=cores)
            allc["host_cores_visible"] = host
            allc["note"] = "one worker per usable CPU (scheduler affinity capped by the cgroup CPU quota)"
            cpu["cpu_baseline_all_cores"] = allc
    # the file-fed leg formats its metadata text in worker processes (cpx.track.bulk.MetaPool: spawned children that
    # never touch the GPU): started here, ahead of this process's first call that initialises it
    meta_pool = None
    if args.meta_workers > 0 and args.stage == "e2e" and args.from_files > 0 and not args.no_extras and not args.denoise \
            and args.frame_size == 32 and not args.config4 and torch.cuda.device_count() > 0:
        from cpx.track.bulk import MetaPool

        meta_pool = MetaPool.make(args.meta_workers)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("CPX_BENCH_FORCE_DIST"):  # the env var exercises the RCCL path on one GPU
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=device)

    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr
    from cpx.pipeline import BatchPipeline
    from cpx.sharding import gather_records, pack_records

    if args.stage == "ir":
        return bench_ir(args, torch, np, dist, device, rank, world, local_rank)
    if args.config4:
        return bench_config4(args, torch, np, dist, device, rank, world, local_rank, cpu)
    e2e = args.stage == "e2e"
    H, W, T = 120, 160, args.frames
    P = H * W
    B = args.clips or 4096
    free, _ = torch.cuda.mem_get_info(device)
    per_clip = T * P * 2 + T * 64 * 32 + T * 80 + 6 * P * 4 + T * P * 4 + (T * 16 * 56 if e2e else T * P * 4)
    reserve = (args.cnn_chunk * 30e6 * (args.frame_size / 32) ** 2) if e2e else 0
    while B > 64 and B * per_clip + reserve > 0.80 * free:
        B //= 2
    eng = TrackEngine(width=W, height=H, model="lepton3", device=local_rank, max_components=64, max_frames=max(T, 45),
                      denoise=args.denoise)
    frames = synth_on_device(torch, device, B, T, seed=1234 + rank)
    offs = (np.arange(B + 1, dtype=np.int64) * T).astype(np.int32)
    t_on = [100000 + 114 * i for i in range(T)]
    ffc = [40000] * T
    meta = np.tile(eng.make_meta(T, t_on, ffc), B)
    total = B * T
    comps = torch.empty(total * 64 * 8, dtype=torch.int32, device=device)
    info = torch.empty(total * 20, dtype=torch.int32, device=device)
    filt = torch.empty((total, H, W), dtype=torch.float32, device=device)
    labels = None if e2e else torch.empty((total, H, W), dtype=torch.int32, device=device)
    outputs = (comps, info, labels, filt, None)
    weights = (synthetic_network_weights(torch, wr, eng, frames, offs, meta, outputs, args.frame_size) if e2e
               else wr.random_weights(N_LABELS, seed=0))
    # the network lives on a second handle (= second HIP stream) so that the HBM-bound track stage of one group of
    # clips overlaps the MFMA-bound network of the previous group (BatchPipeline sub_batches)
    overlap = e2e and args.sub_batches > 1
    ceng = TrackEngine(width=W, height=H, model="lepton3", device=local_rank, max_frames=45) if overlap else eng
    ceng.set_cnn_math(args.cnn_math)
    net = wr.WRResNetDevice(ceng, weights, N_LABELS) if e2e else None
    pipe = BatchPipeline(eng, net, n_labels=N_LABELS, fp_index=4, cnn_chunk=args.cnn_chunk, frame_size=args.frame_size)
    state = {}

    def step():
        if e2e:
            res = pipe.run(frames, offs, meta, outputs=outputs, sub_batches=args.sub_batches)
            state["res"] = res
            state["track_ms"], state["track_n"] = res.track_timing
            if dist is not None:  # every rank enters the collective, also one whose clips produced no track
                # the record of SURVEY section 8(d) config 4: [clip_id, track_id, n_labels x f32]
                if res.n_tracks:
                    rec = pack_records(res.track_clip[:, 0] + rank * B, res.track_clip[:, 1], res.scores)
                else:
                    rec = torch.empty((0, 2 + N_LABELS), dtype=torch.int32, device=device)
                state["gathered"] = gather_records(rec, dist, capacity=B * pipe.tp.max_tracks)
            return res.track
        res = eng.track_batch(frames, offs, meta, outputs=outputs)
        eng.synchronize()
        state["track_ms"], state["track_n"] = eng.last_kernel_timing()
        if dist is not None:
            nc = info.view(total, 20)[:, 1].view(B, T).sum(dim=1).to(torch.int32)
            rec = torch.stack([torch.arange(B, device=device, dtype=torch.int32) + rank * B, nc], dim=1)
            state["gathered"] = gather_records(rec, dist, capacity=B)
        return res

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    kernel_ms, kernel_launches = 0.0, 0
    ceng.conv_timing(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        kernel_ms += state["track_ms"]
        kernel_launches += state["track_n"]
    fence()
    elapsed = time.perf_counter() - t0
    conv = ceng.conv_timing() if e2e else {}
    ceng.conv_timing(False)
    rank_times = [elapsed]
    if dist is not None:
        every = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
        dist.all_gather(every, torch.tensor([elapsed], dtype=torch.float64, device=device))
        rank_times = [float(v.item()) for v in every]  # each rank's own time: what a < N-fold result is attributed with
        elapsed = max(rank_times)
    res.check()

    if rank == 0:
        frames_done = world * B * T * args.steps
        avg_launch_s = kernel_ms / 1e3 / max(kernel_launches, 1)
        # a launch walks every clip of its group (the whole batch, or one of the sub-batches) through all of its frames
        # (one workgroup per clip; CPX_TRACK_PER_STEP=1: one launch per frame step): clip-frames per launch
        clips_per_launch = B * T * args.steps / max(kernel_launches, 1)
        bytes_per_launch = (ALGO_BYTES_PER_FRAME - (LABEL_BYTES_PER_FRAME if e2e else 0)) * clips_per_launch
        hbm = bytes_per_launch / avg_launch_s / 1e9
        # what the kernel really moves: the background is kept as uint16 (a floor of a mean of uint16 frames), so its
        # read + write cost 76,800 B per frame instead of the 153,600 B of SURVEY's int32 count
        moved_per_launch = bytes_per_launch - 76800 * clips_per_launch
        # round 6: a fresh batch of at most 1023 frames per clip keeps the per-pixel kept-frame count in the window sum's top ten
        # bits (cpx_frame_kernel<true>): the uint16 count array is neither read nor written -- another 76,800 B per frame less
        packed_state = T <= 1023 and os.environ.get("CPX_TRACK_PACKED_STATE", "1") != "0"
        if packed_state:
            moved_per_launch -= 76800 * clips_per_launch
        hbm_moved = moved_per_launch / avg_launch_s / 1e9
        # headline fraction: the bytes the kernel moves (uint16 background, count packed into the window sum: 460,800 B per frame,
        # 384,000 without the label image); the figure on SURVEY section 8(d)'s count (int32 background, a count array: 614,400 B)
        # stays beside it
        track_roof = {"kernel": "cpx_frame_kernel", "bound": "hbm", "achieved": round(hbm_moved, 1), "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": round(hbm_moved / HBM_PEAK_GBS, 4),
                      "achieved_survey_bytes": round(hbm, 1), "frac_survey_bytes": round(hbm / HBM_PEAK_GBS, 4),
                      "traffic": pmc_traffic("frame_kernel_e2e" if e2e else "frame_kernel_track", clips_per_launch),
                      "traffic_source": PMC_NOTE,
                      "avg_launch_us": round(avg_launch_s * 1e6, 2), "launches": kernel_launches,
                      "algorithmic_bytes_per_launch": bytes_per_launch,
                      "moved_bytes_per_launch": moved_per_launch, "achieved_moved": round(hbm_moved, 1),
                      "frac_moved": round(hbm_moved / HBM_PEAK_GBS, 4),
                      "frac_moved_of_achievable": round(hbm_moved / 6290.0, 4),
                      "packed_state": packed_state,
                      "note": "achieved / frac count the bytes the kernel moves (uint16 background, the kept-frame count "
                              "packed into the window sum: the algorithmic bytes of THIS data layout, DESIGN.md section 4); "
                              "*_survey_bytes count SURVEY section 8(d)'s int32 background and separate count array; 6.29 TB/s is "
                              "the measured float4-copy rate of the guide"}
        line = {
            "metric": "CPTV frames/s end-to-end (track+classify) at 160x120" if e2e else
                      "CPTV frames/s (track stage only: background + region-label HIP kernels) at 160x120",
            "value": round(frames_done / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("u16/i32 track stage (f32 normalise, f64 gates); f32 crop/tile; f32 CNN (%s)"
                      % MATH_DTYPE[args.cnn_math]) if e2e else "u16/i32 (f32 normalise, f64 background weights)",
            "data": "synthetic",
            "per_rank": dict(per_rank_summary(rank_times, args.steps, [B * T] * world), numa=numa),
            "config": {
                "workload": ("synthetic 160x120 uint16 clips -> track (BASELINE configs[1]) -> 25-frame segments -> "
                             "crop/tile + WR-ResNet-22-4 forward (configs[2]/[3]), seeded random kernels with BatchNorm "
                             "statistics fitted to the workload's own crops") if e2e else
                            "BASELINE.json configs[1]: synthetic 160x120 uint16 clips, background + region-label kernels",
                "clips_per_gpu": B,
                "frames_per_clip": T,
                "camera_model": "lepton3",
                "denoise": bool(args.denoise),
                "sharding": "clips across ranks, all_gather of [clip_id, track_id, 17 x f32] records" if world > 1 else "single GPU",
            },
        }
        if e2e:
            r = state["res"]
            line["config"].update({"kept_tracks_per_step": int(r.n_tracks), "classified_segments_per_step": int(r.n_samples),
                                   "frame_size": args.frame_size, "n_labels": N_LABELS, "cnn_chunk": args.cnn_chunk,
                                   "sub_batches": args.sub_batches})
            key = 32 * 10000 + 32 * 10 + 1  # the stage-2 3x3 convolutions launched on their own (32 -> 32 channels per group, stride 1)
            key4 = 32 * 10000 + 32 * 10 + 4  # conv_block32_kernel: a stage-2 residual block (two of those convolutions) in one launch
            key0 = 8 * 10000 + 32 * 10 + 4  # conv_block32_kernel<true>: the stage's first block (8 -> 32 -> 32 channels per group + the 1x1 shortcut)
            key3 = 64 * 10000 + 64 * 10 + 1  # the stage-3 ones: same FLOPs per sample, half the bytes
            if (key in conv and conv[key][1] > 0) or (key4 in conv and conv[key4][1] > 0):
                bf3 = args.cnn_math in PRODUCTS
                products = PRODUCTS.get(args.cnn_math, BF16X3_PRODUCTS)
                area = (args.frame_size / 32.0) ** 2  # map area relative to the 160 x 160 maps of frame size 32
                side = 5 * args.frame_size
                tensor2 = side * side * 64 * 4  # bytes of one stage-2 activation tensor of one sample (float32 NHWC)
                peak = round(MFMA_BF16_PEAK_TFLOPS / products, 1) if bf3 else MFMA_F32_PEAK_TFLOPS
                fused = key4 in conv and conv[key4][1] > 0

                def leg(k, pmc_key, kernel, tensors, note, convs=1):
                    """One convolution kernel of the step: its launches' algorithmic FLOPs and bytes against their HIP-event
                    time; the roof it sits nearer to is `bound`, the other one stays beside it."""
                    n_, ms_, fl_ = conv[k]
                    spl = fl_ / n_ / (convs * STAGE2_CONV_FLOPS_PER_SAMPLE * area)  # samples per launch
                    ab = spl * tensor2 * tensors
                    tf_ = fl_ / (ms_ / 1e3) / 1e12
                    gbs = ab / (ms_ / n_ / 1e3) / 1e9
                    hbm_bound = bf3 and gbs / HBM_PEAK_GBS > tf_ / peak
                    tr = pmc_traffic(pmc_key, spl * area)
                    return {"kernel": kernel, "bound": "hbm" if hbm_bound else "mfma",
                            "achieved": round(gbs, 1) if hbm_bound else round(tf_, 2),
                            "peak": HBM_PEAK_GBS if hbm_bound else peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 4) if hbm_bound else round(tf_ / peak, 4),
                            "mfma": {"achieved": round(tf_, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf_ / peak, 4)},
                            "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
                            "traffic": tr, "traffic_source": PMC_NOTE if tr is not None else None,
                            "avg_launch_us": round(ms_ / n_ * 1e3, 2), "launches": n_, "ms_per_step": round(ms_ / args.steps, 2),
                            "algorithmic_flops_per_launch": fl_ / n_, "algorithmic_bytes_per_launch": ab,
                            "algorithmic_bytes_note": note,
                            "peak_note": ("HBM3E 8 TB/s; " if bf3 else "") +
                                         ("dense 16-bit MFMA peak 2500 TFLOP/s / %d products per float32 multiply-add" % products
                                          if bf3 else "dense fp32-input MFMA peak")}
                kernels = {}
                if key in conv and conv[key][1] > 0:
                    if bf3:
                        kernels["conv_stage2"] = leg(
                            key, "conv_stage2",
                            "%s (the 16x16x32 MFMA form of the split-operand kernel, one 32-column slice per workgroup: the stage-2 "
                            "3x3 convs launched on their own, 64->64 ch at %dx%d, groups 2)" % (bf3w_kernel_name(32, args.cnn_math), side, side),
                            2.25 if fused else 2.6,
                            "the second convolution of the stage's first block: mid in, output out, the fused 1x1 shortcut's 16-channel input"
                            if fused else "input + output (+ residual in 3 of the 5 launches of this shape per forward)")
                    else:
                        kernels["conv_stage2"] = leg(key, "conv_stage2", "conv_mfma_kernel<8,1,1,3,2,16> (stage-2 3x3 conv, 64->64 ch, "
                                                     "groups 2, %dx%d)" % (side, side), 2.6, "input + output (+ residual in 3 of 5)")
                if fused:
                    kernels["conv_block"] = leg(
                        key4, "conv_block",
                        "conv_block32_kernel (a stage-2 residual block past the first in one launch: two 3x3 convs 64->64 ch at %dx%d, "
                        "groups 2, the tensor between them kept in LDS; FLOPs of both, the recomputed halo not counted)" % (side, side),
                        2.0, "the block's input in, its output out (the residual is the input: halo re-reads and the residual come from L2)",
                        convs=2)
                if key0 in conv and conv[key0][1] > 0:
                    # round 6: no conv1_1 launch of its own = the layer is computed inside this kernel (its FLOPs are in the record)
                    c1_inside = (1 * 10000 + 8 * 10 + 1) not in conv
                    kernels["conv_block0"] = leg(
                        key0, "conv_block0",
                        ("conv_block32_kernel<true, true> (conv1_1 2->16 ch computed while the patch is staged + the first residual block of "
                         "stage 2 in one launch: 3x3 conv 16->64 ch, 3x3 conv 64->64 ch + the 1x1 shortcut 16->64, %dx%d, groups 2; FLOPs of "
                         "the three 3x3 convolutions)" if c1_inside else
                         "conv_block32_kernel<true> (the first residual block of stage 2 in one launch: 3x3 conv 16->64 ch, 3x3 conv "
                         "64->64 ch + the 1x1 shortcut 16->64, %dx%d, groups 2; FLOPs of the two 3x3 convolutions)") % (side, side),
                        (2 + 64) / 64.0 if c1_inside else 1.25,
                        "the 2-channel sample in, the block's 64-channel output out" if c1_inside else
                        "the block's 16-channel input in, its 64-channel output out",
                        convs=1.25 + (16.0 / (64 * 32) if c1_inside else 0.0))
                if bf3 and key3 in conv:
                    k3name = ("conv_rw_kernel<1, 2, 16, *, *> (weights resident in registers, one wave per SIMD, pixel-row fragments "
                              "shared by the three tap rows; the one launch in five that carries the stage's 1x1 shortcut stays on %s: "
                              "the stage-3 3x3 convs, 128->128 ch at %dx%d, groups 2)" % (bf3w_kernel_name(64, args.cnn_math), side // 2, side // 2)
                              if args.cnn_math == "fp16x2" and os.environ.get("CPX_CNN_RW", "7") not in ("0", "2", "4", "6") else
                              "%s (both 32-column slices of a group from one staged patch: the stage-3 3x3 convs, 128->128 ch at %dx%d, groups 2)"
                              % (bf3w_kernel_name(64, args.cnn_math), side // 2, side // 2))
                    kernels["conv_stage3"] = leg(
                        key3, "conv_stage3", k3name, 0.5 * 2.6,
                        "input + output (+ residual in 3 of the 5 launches of this shape per forward); half of a stage-2 tensor each")
                # `roofline` is the kernel the step spends most of its time in -- since the fp16x2 mode and the fused blocks that
                # is the track kernel (one launch per step), not a convolution; every kernel above a tenth of the step stands
                # beside it under `kernels`, `roofline_conv` names the largest convolution kernel
                kernels["track"] = dict(track_roof, ms_per_step=round(kernel_ms / args.steps, 2))
                dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
                conv_dom = max((k for k in kernels if k != "track"), key=lambda k: kernels[k]["ms_per_step"])
                line["roofline"] = dict(kernels[dom], dominant="%s: %.1f ms of the %.1f ms step" % (
                    dom, kernels[dom]["ms_per_step"], elapsed * 1e3 / args.steps),
                    kernels={k: v for k, v in kernels.items() if k != dom})
                line["roofline_conv"] = dict(kernels[conv_dom], which=conv_dom)
                # the whole network against the matrix pipe: float32-equivalent FLOPs of every convolution of
                # the step / the step's convolution time, against the 16-bit MFMA peak over the products one float32 multiply-add
                # costs in this mode -- conv1 (vector kernel) and whatever runs on the fp32 MFMA are inside the time, so this
                # is a lower bound of what the split-operand kernels reach
                tot_ms_c = sum(v[1] for v in conv.values())
                tot_fl_c = sum(v[2] for v in conv.values())
                agg_tf = tot_fl_c / (tot_ms_c / 1e3) / 1e12
                line["roofline"]["conv_aggregate"] = {
                    "bound": "mfma", "achieved": round(agg_tf, 2), "peak": peak, "unit": "TFLOP/s (float32-equivalent)",
                    "frac": round(agg_tf / peak, 4), "mfma_issued_tflops": round(agg_tf * products, 1) if bf3 else None,
                    "frac_of_16bit_mfma_peak": round(agg_tf * products / MFMA_BF16_PEAK_TFLOPS, 4) if bf3 else None,
                    "ms_per_step": round(tot_ms_c / args.steps, 2), "share_of_step": round(tot_ms_c / args.steps / (elapsed * 1e3 / args.steps), 4),
                    "note": "all convolution launches of the step (cpx_conv_timing_report: HIP events on the handle's stream)"}
                if key in conv and conv[key][1] > 0:
                    line["roofline"]["stage2_tflops"] = round(conv[key][2] / (conv[key][1] / 1e3) / 1e12, 2)
                tot_ms = sum(v[1] for v in conv.values())
                tot_fl = sum(v[2] for v in conv.values())
                line["cnn"] = {"samples_per_s": round(int(r.n_samples) / max(tot_ms / args.steps / 1e3, 1e-9), 1),
                               "samples_per_s_note": "classified segments of a step / the step's convolution time (SURVEY config 3)",
                               "conv_time_ms_per_step": round(tot_ms / args.steps, 2),
                               "conv_tflops_all_layers": round(tot_fl / (tot_ms / 1e3) / 1e12, 2),
                               "track_kernel_ms_per_step": round(kernel_ms / args.steps, 2),
                               "math": args.cnn_math,
                               "layers_ms_per_step": {str(k): round(v[1] / args.steps, 2) for k, v in sorted(conv.items())},
                               "layers": conv_layer_table(conv, args.cnn_math, args.steps)}
            else:
                line["roofline"] = track_roof
                line["roofline_conv"] = None
            line["roofline_track"] = track_roof
        else:
            line["config"]["outputs"] = "components + label image + filtered image"
            line["roofline"] = track_roof
        line.update(cpu)  # measured before the GPU was initialised (top of main)
        if e2e and world == 1 and not args.no_extras and not args.denoise and args.frame_size == 32:
            # ---- the same run, after the timed region: the two configurations the headline does not cover ----
            # (a) the reference's DEFAULT tracking configuration (denoise = true: NLM between normalise and blur,
            #     SURVEY F7), track stage over a slice of the resident clips
            # (b) frame size 64 (BASELINE north_star's 64 x 64 crops): end to end over a slice
            # (c) the opt-in math mode CPX_CNN_MATH_BF16X2 over the headline's own step: stages 2-4 with two bf16 planes
            #     per operand rounded to nearest and three products per K step (include/cpx.h) -- its rate, and how far
            #     its logits are from the default mode's on the step's own classified segments
            if args.cnn_math in PRODUCTS and not overlap:
                ref = state["res"]
                ref_logits, ref_probs = ref.logits.clone(), ref.probs.clone()
                ref_overflow = eng.cnn_last_overflow() if args.cnn_math == "fp16x2" else False
                line["cnn"]["fp16_overflow_rerun_in_last_forward"] = bool(ref_overflow)
                what = {"bf16x3": "the exact mode: every operand split into three bf16 planes, six products per K step",
                        "bf16x2": "opt-in: the stride-1 3x3 layers of stages 2-4 multiply two bf16 planes per operand, rounded "
                                  "to nearest (<= 2^-16 relative per operand), in three products per K step; NOT float32 per element",
                        "fp16x2": "two fp16 planes per operand (<= 2^-22 relative per operand), three products per K step"}
                for other in ("bf16x3", "bf16x2", "fp16x2"):
                    if other == args.cnn_math:
                        continue
                    eng.set_cnn_math(other)
                    pipe.run(frames, offs, meta, outputs=outputs)                  # warm-up
                    eng.conv_timing(True)
                    torch.cuda.synchronize(device)
                    t1 = time.perf_counter()
                    r2 = pipe.run(frames, offs, meta, outputs=outputs)
                    torch.cuda.synchronize(device)
                    dt2 = time.perf_counter() - t1
                    c2 = eng.conv_timing()
                    eng.conv_timing(False)
                    eng.set_cnn_math(args.cnn_math)
                    k2, k3 = 32 * 10000 + 32 * 10 + 1, 64 * 10000 + 64 * 10 + 1
                    leg = {"what": "the same step with CPX_CNN_MATH=%s (%s); every other layer as in the headline's mode (%s)"
                                   % (other, what[other], args.cnn_math),
                           "frames_per_s": round(B * T / dt2, 1), "ms_per_step": round(dt2 * 1e3, 2),
                           "classified_segments": int(r2.n_samples),
                           "conv_time_ms_per_step": round(sum(v[1] for v in c2.values()), 2),
                           "samples_per_s": round(int(r2.n_samples) / (sum(v[1] for v in c2.values()) / 1e3), 1),
                           "max_abs_logit_difference_to_headline_mode": float((r2.logits - ref_logits).abs().max()),
                           "max_abs_probability_difference_to_headline_mode": float((r2.probs - ref_probs).abs().max()),
                           "max_abs_logit": float(ref_logits.abs().max()),
                           "tracks": int(ref.best.numel())}
                    # tracks whose best label changed, and how close their two best scores were in the headline's mode
                    # (seeded random kernels: near-ties exist; a real model's margins are what its accuracy rests on)
                    moved = (r2.best != ref.best).nonzero().flatten()
                    leg["tracks_with_another_best_label"] = int(moved.numel())
                    if moved.numel():
                        top2 = ref.scores[moved].float().topk(2, dim=1).values
                        leg["largest_score_margin_among_them"] = float((top2[:, 0] - top2[:, 1]).max())
                    for name, k in (("stage2", k2), ("stage3", k3), ("stage4", 128 * 10000 + 128 * 10 + 1)):
                        if k in c2 and c2[k][1] > 0:
                            tfe = c2[k][2] / (c2[k][1] / 1e3) / 1e12
                            leg[name] = {"float32_equivalent_tflops": round(tfe, 2),
                                         "frac_of_pipe_peak": round(tfe / (MFMA_BF16_PEAK_TFLOPS / PRODUCTS[other]), 4),
                                         "avg_launch_us": round(c2[k][1] / c2[k][0] * 1e3, 2)}
                    line[other] = leg
                    del r2
                del ref_logits, ref_probs
            net.close()
            nb = min(B, 512)
            o2 = offs[: nb + 1]
            deng = TrackEngine(width=W, height=H, model="lepton3", device=local_rank, max_components=64,
                               max_frames=max(T, 45), denoise=True)
            deng.track_batch(frames, o2, meta[: nb * T], outputs=outputs)       # warm-up
            deng.synchronize()
            t1 = time.perf_counter()
            deng.track_batch(frames, o2, meta[: nb * T], outputs=outputs)
            deng.synchronize()
            dt = time.perf_counter() - t1
            kms, kn = deng.last_kernel_timing()
            # the WHOLE path with the reference's default configuration: track (NLM inside) + classify, same slice
            net_dn = wr.WRResNetDevice(deng, weights, N_LABELS)
            pipe_dn = BatchPipeline(deng, net_dn, n_labels=N_LABELS, fp_index=4, cnn_chunk=args.cnn_chunk, frame_size=32)
            pipe_dn.run(frames, o2, meta[: nb * T], outputs=outputs)            # warm-up
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            r_dn = pipe_dn.run(frames, o2, meta[: nb * T], outputs=outputs)
            torch.cuda.synchronize(device)
            dt_e2e = time.perf_counter() - t1
            net_dn.close()
            deng.close()
            # the same slice without denoise: the difference is the NLM kernel
            eng.track_batch(frames, o2, meta[: nb * T], outputs=outputs)
            eng.synchronize()
            t1 = time.perf_counter()
            eng.track_batch(frames, o2, meta[: nb * T], outputs=outputs)
            eng.synchronize()
            dt0 = time.perf_counter() - t1
            nlm_s = dt - dt0
            # Instruction census of cpx_nlm_kernel<10,160> (profiles/r04_nlm_sq_counters.json):
            # the bound is the MINIMAL packed instruction count of this algorithm form -- paired offsets, 16-bit pairs,
            # row sums through LDS, sliding column sums -- at the vector pipes' nominal rate; `frac` = bound / achieved,
            # so it says how far the shipped kernel is from that count (it was the kernel's own count until round 3).
            #   row pass, per 12 columns and offset PAIR: 18 byte differences (SDWA) + 9 squares + 8 shifted copies +
            #   26 window adds + 6 caps + 5 alignments of the b row = 72 -> 3.0 per (pixel, offset)
            #   column pass, per 2 pixels and offset PAIR: 2 shifts + 2 caps + 2 table addresses (dot2) + 4 transposes +
            #   4 accumulations (dot2) + 4 sliding updates = 18 -> 4.5, + window set-up 0.3
            px_off = H * W * 441.0
            valu_min, valu_shipped, lds_shipped = 7.8, 12.2, 2.65
            clock, cus = 2.4e9, 256.0
            t_min = valu_min * px_off / 128.0 / clock / cus        # 4 SIMD-32 per CU: 128 lane-operations per cycle
            t_shipped = valu_shipped * px_off / 128.0 / clock / cus
            t_lds = lds_shipped * px_off / 32.0 / clock / cus       # ds_read_b32 / u16: 32 lanes per cycle and CU
            nlm_per_frame = nlm_s / (nb * T)
            line["default_config"] = {
                "what": "the reference's DEFAULT configuration (tracking.denoise = true, SURVEY F7) over %d of the resident "
                        "clips x %d frames, same run: end to end (track with the NLM kernel + classify), and the track "
                        "stage alone" % (nb, T),
                "frames_per_s": round(nb * T / dt_e2e, 1), "ms_per_step": round(dt_e2e * 1e3, 2),
                "classified_segments": int(r_dn.n_samples),
                "track_stage_frames_per_s": round(nb * T / dt, 1),
                "nlm_us_per_frame": round(nlm_per_frame * 1e6, 3),
                "nlm_share_of_step": round(nlm_s / dt_e2e, 3),
                "frames_per_s_denoise_off_same_slice": round(nb * T / dt0, 1)}
            line["roofline_nlm"] = {
                "kernel": "cpx_nlm_kernel<10,160>", "bound": "vector instruction issue at the MINIMAL instruction count of the "
                          "paired sliding-sum form (no HBM or matrix roof applies: 19 KB in, 19 KB out per frame, integer "
                          "arithmetic on an LDS-resident frame)",
                "model": {"pixel_offsets_per_frame": px_off, "minimal_vector_lane_ops_per_pixel_offset": valu_min,
                          "shipped_vector_lane_ops_per_pixel_offset": valu_shipped,
                          "shipped_lds_lane_ops_per_pixel_offset": lds_shipped,
                          "vector_lane_ops_per_cycle_per_cu": 128.0, "lds_lanes_per_cycle_per_cu": 32, "clock_hz": clock,
                          "cus": cus, "census": "DESIGN.md section 5; SQ_INSTS_VALU / SQ_INSTS_LDS of "
                                                "profiles/r04_nlm_sq_counters.json (rocprofv3 --pmc, separate passes); "
                                                "round 3 shipped 15.1 / 2.3"},
                "achieved_us_per_frame": round(nlm_per_frame * 1e6, 3),
                "minimal_count_bound_us_per_frame": round(t_min * 1e6, 3),
                "shipped_count_at_nominal_rate_us_per_frame": round(t_shipped * 1e6, 3),
                "lds_issue_us_per_frame": round(t_lds * 1e6, 3),
                "frac": round(t_min / nlm_per_frame, 4), "unit": "minimal-count vector issue time / achieved time",
                "shipped_over_minimal_count": round(valu_shipped / valu_min, 2),
                "note": "one 1024-thread workgroup (a whole frame) per CU, 16 waves; measured issue cost of every "
                        "instruction of the kernel: 2.74 cycles per wave-instruction and SIMD at 4 waves per SIMD "
                        "(scratch/valu_cost_probe.hip), i.e. 0.73 of the nominal rate; LDS busy 56 % (a third of it bank "
                        "conflicts of the weight-table gather)"}
            nb64 = min(B, 1024)
            net64 = wr.WRResNetDevice(eng, weights, N_LABELS)
            pipe64 = BatchPipeline(eng, net64, n_labels=N_LABELS, fp_index=4, cnn_chunk=512, frame_size=64)
            o64 = offs[: nb64 + 1]
            pipe64.run(frames, o64, meta[: nb64 * T], outputs=outputs)          # warm-up (arena for 320 x 320 inputs)
            eng.conv_timing(True)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            r64 = pipe64.run(frames, o64, meta[: nb64 * T], outputs=outputs)
            torch.cuda.synchronize(device)
            dt64 = time.perf_counter() - t1
            c64 = eng.conv_timing()
            eng.conv_timing(False)
            f64 = {"what": "end to end at frame_size 64 (320 x 320 network input) over %d of the resident clips, same run" % nb64,
                   "frames_per_s": round(nb64 * T / dt64, 1), "classified_segments": int(r64.n_samples)}
            k2, k3 = 32 * 10000 + 32 * 10 + 1, 64 * 10000 + 64 * 10 + 1
            if k2 in c64 and k3 in c64 and c64[k2][1] + c64[k3][1] > 0:
                tf64 = (c64[k2][2] + c64[k3][2]) / ((c64[k2][1] + c64[k3][1]) / 1e3) / 1e12
                f64["stage2_3_conv_tflops"] = round(tf64, 2)
                f64["stage2_3_conv_frac"] = round(tf64 / (MFMA_BF16_PEAK_TFLOPS / PRODUCTS.get(args.cnn_math, BF16X3_PRODUCTS)), 4)
                f64["math"] = args.cnn_math
            line["fs64"] = f64
            net64.close()
    # ---- after the timed region, in the same process(es): what the headline does not cover ----
    extras = e2e and not args.no_extras and not args.denoise and args.frame_size == 32
    if extras and (world > 1 or args.from_files > 0):
        # the resident workload makes room first
        if net is not None:
            try:
                net.close()
            except Exception:  # (already closed by the fs64 measurement)
                pass
        pipe = None
        state.clear()
        del frames, comps, info, filt, labels, outputs, res
        eng.close()
        if overlap:
            ceng.close()
        torch.cuda.empty_cache()
    if extras and world > 1:
        # the weak-scaling value above is trivially N-fold (identical clips per rank): next to it the strong-scaling,
        # imbalanced workload of configs[3] -- 10,000 clips of 90-540 frames LPT-sharded over the ranks
        sub = bench_config4(args, torch, np, dist, device, rank, world, local_rank, {}, as_sub_object=True,
                            n_clips=10000, steps=max(1, min(args.steps, 3)))
        if rank == 0:
            cfg4 = sub["config"]
            line["config4"] = {"what": cfg4["workload"], "scaling": "strong", "value": sub["value"], "unit": "frames/s",
                               "ms_per_step": sub["ms_per_step"], "steps": sub["steps"], "clips": cfg4["clips"],
                               "frames_total": cfg4["frames_total"], "frames_per_rank": cfg4["frames_per_rank"],
                               "imbalance": cfg4["imbalance"], "per_rank": sub.get("per_rank"),
                               "records_gathered": cfg4["records_gathered"],
                               "gather_ms_per_step_rank0": cfg4["gather_ms_per_step_rank0"]}
    if extras and args.from_files > 0:
        # N > 1: every rank runs the file-fed path over ITS share of the recordings (shard_files' partition: recordings
        # are independent), with its host stages beside the other ranks' on the node's CPUs -- the scaling risk of this
        # path is the host (staging, metadata text), not a collective, so the per-rank split is what is reported
        share = args.from_files if world == 1 else max(512, args.from_files // world)
        try:
            ff = bench_from_files(args, torch, np, local_rank, weights, T, n_files=share, with_fixtures=world == 1,
                                  meta_pool=meta_pool)
        except Exception as e:  # noqa: BLE001 -- the headline line above is complete: report, do not lose it
            ff = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
        if isinstance(ff, dict) and world > 1:
            ff["numa"] = numa
        if world == 1:
            line["from_files"] = ff
        else:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, ff)
            if rank == 0:
                line["from_files"] = aggregate_from_files(per_rank, share, usable_cpus())
    if meta_pool is not None:
        meta_pool.close()
    if rank == 0:
        print(json.dumps(line), flush=True)
    if not (extras and (world > 1 or args.from_files > 0)):
        eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
