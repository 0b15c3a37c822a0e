"""Builds profiles/<round>_e2e_pmc.json (round = $CPX_ROUND, default r02) from the rocprofv3 counter CSVs of four passes (run on the GPU box from the repo
root, after `cd /tmp && export TMPDIR=/tmp && cd -`):

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_e2e_fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0   (the bench's own 4,096 clips)
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_e2e_write -- python3 bench.py --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0   (the bench's own 4,096 clips)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_trk_fetch -- python3 bench.py --stage track --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0   (the bench's own 4,096 clips)
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_trk_write -- python3 bench.py --stage track --steps 1 --warmup 0 --cpu-clips 0 --no-extras --from-files 0   (the bench's own 4,096 clips)

FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-byte read requests at 64 bytes; Infinity-Cache hits are
counted too); WRITE_SIZE is taken as is.  Values are KiB per dispatch."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("CPX_ROUND", "r06")
CLIPS = int(os.environ.get("CPX_PMC_CLIPS", "4096"))  # clips per GPU of the profiled bench command (round 6: the bench's own size)


def rocprof_check():
    """--rocprof-check: fail when the committed PMC summary is older than the last commit that touched the kernels
    (classifier-pipeline_amd/csrc): bench.py quotes `roofline.traffic` from that file, so it must not go stale."""
    import subprocess

    summary = os.path.join("profiles", "%s_e2e_pmc.json" % ROUND)

    def last_commit_time(path):
        out = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%ct", "--", path], capture_output=True, text=True)
        return int(out.stdout.strip() or 0)

    t_src, t_sum = last_commit_time("classifier-pipeline_amd/csrc"), last_commit_time(summary)
    if not os.path.exists(os.path.join(ROOT, summary)) or t_sum == 0:
        print("rocprof-check: %s is not committed" % summary)
        return 1
    if t_sum < t_src:
        print("rocprof-check: %s (commit time %d) is older than the last csrc commit (%d): re-run scratch/refresh_pmc.sh "
              "on the GPU box" % (summary, t_sum, t_src))
        return 1
    print("rocprof-check: %s is newer than the last csrc commit" % summary)
    return 0


if "--rocprof-check" in sys.argv:
    sys.exit(rocprof_check())


def rows(dirname, counter, kernel_sub, phase=None):
    """(grid, value) of the kernel's dispatches in dispatch order.  phase = (period, keep): the dispatches of the
    kernel come in runs of `period` per layer group, alternating; keep the runs with (index // period) % 2 == keep."""
    path = max(glob.glob(os.path.join(ROOT, "gpurun_out", dirname, "*", "*_counter_collection.csv")),
               key=os.path.getmtime)  # the newest pass (gpurun merges outputs, older passes stay around)
    rs = [(int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(path))
          if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in ((kernel_sub,) if isinstance(kernel_sub, str) else kernel_sub))]
    rs.sort()
    out = [(g, v) for _, g, v in rs]
    if phase is not None:
        period, keep = phase
        out = [gv for i, gv in enumerate(out) if (i // period) % 2 == keep]
    return out


def common_grid(dirname, counter, kernel_sub):
    """the grid size most of the kernel's launches used"""
    grids = [g for g, _ in rows(dirname, counter, kernel_sub)]
    return max(set(grids), key=grids.count)


def avg(dirname, counter, kernel_sub, grid=None, phase=None):
    vals = [v for g, v in rows(dirname, counter, kernel_sub, phase) if grid is None or g == grid]
    if grid is None and vals:
        # persistent kernels launch one grid whatever the batch: the launches of the untimed BatchNorm calibration (48 samples)
        # are told apart by their traffic -- only launches with at least half the largest value count
        top = max(vals)
        vals = [v for v in vals if v * 2 >= top]
    return sum(vals) / len(vals), len(vals)


def section(kernel, fetch_dir, write_dir, kernel_sub, grid, units, unit, algo, note, phase=None):
    f, n = avg(fetch_dir, "FETCH_SIZE", kernel_sub, grid, phase)
    w, _ = avg(write_dir, "WRITE_SIZE", kernel_sub, grid, phase)
    rd, wr = 2 * f * 1024, w * 1024
    return {"kernel": kernel, "launches": n, "units_per_launch": units, "unit": unit, "FETCH_SIZE_KiB_avg": f,
            "WRITE_SIZE_KiB_avg": w, "hbm_read_bytes_per_launch_corrected": rd, "hbm_write_bytes_per_launch": wr,
            "hbm_traffic_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": algo,
            "algorithmic_bytes_note": note, "traffic_over_algorithmic": (rd + wr) / algo}


out = {"command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py [--stage track] "
                  "--clips %d --steps 1 --warmup 0 --cpu-clips 0 --no-extras (separate passes, scratch/make_pmc_profile.py)" % CLIPS,
       "note": "KiB per dispatch, averaged over the launches named; FETCH_SIZE doubled per MI355X_MICROARCH.md, "
               "WRITE_SIZE as is"}
CONV = "conv_bf3w_kernel<"
# stage 2 (64->64 ch, 160x160: 100 tiles per sample; instantiation <false, false, 1>) and stage 3 (128->128 ch, 80x80:
# 25 tiles, both 32-column slices in one workgroup; <false, true, 2>): 100 vs 25 workgroups per sample and group
# samples per launch from the stage-3 instantiation's grid (25 tiles x 2 groups x 512 threads per sample); the stage-2
# instantiation only launches when the stage's first block is not fused (CPX_CNN_BLOCK_FUSION < 2)
CONV3, CONV2 = CONV + "false, true, 2,", CONV + "false, true, 1,"
grid3 = max(g for g, _ in rows("pmc_e2e_fetch", "FETCH_SIZE", CONV3))
conv_n = grid3 // (25 * 2 * 512)
grid2 = 100 * 2 * 512 * conv_n
has2 = any(g == grid2 for g, _ in rows("pmc_e2e_fetch", "FETCH_SIZE", CONV2))
BLOCK = "conv_block32_kernel"
BLOCKS = "conv_block32s_kernel"  # round 6: the blocks past the stage's first (the two convolutions on different waves)
fused = bool(rows("pmc_e2e_fetch", "FETCH_SIZE", BLOCK)) or bool(rows("pmc_e2e_fetch", "FETCH_SIZE", BLOCKS))
if has2:
  out["conv_stage2"] = section(
      "conv_bf3w_kernel<false, true, 1, 1, 2, true, false> (fp16x2, the default math), stage-2 launches of %d samples (64->64 ch, 160x160)" % conv_n, "pmc_e2e_fetch",
      "pmc_e2e_write", CONV2, grid2, conv_n, "samples", conv_n * 160 * 160 * 64 * 4 * (2.25 if fused else 2.6),
      "the second convolution of the stage's first block (the others run inside conv_block32_kernel): mid in, output out, the "
      "fused shortcut's 16-channel input: N*160*160*64*4 B * 2.25" if fused else
      "input + output (+ residual in 3 of the 5 stage-2 convolutions of this shape): N*160*160*64*4 B * 2.6")
if rows("pmc_e2e_fetch", "FETCH_SIZE", BLOCK + "<true, true>"):
    # round 6: conv1_1 computed inside (conv_block32_kernel<true, true>): the 2-channel sample in, the 64-channel output out
    out["conv_block0"] = section(
        "conv_block32_kernel<true, true> (fp16x2), conv1_1 + the first residual block of stage 2 of %d samples in one launch" % conv_n,
        "pmc_e2e_fetch", "pmc_e2e_write", BLOCK + "<true, true>", None, conv_n, "samples", conv_n * 160 * 160 * (2 + 64) * 4,
        "the 2-channel sample in, the block's 64-channel output out: N*160*160*(2 + 64)*4 B")
elif rows("pmc_e2e_fetch", "FETCH_SIZE", BLOCK + "<true"):
    out["conv_block0"] = section(
        "conv_block32_kernel<true> (fp16x2), the first residual block of stage 2 of %d samples in one launch" % conv_n,
        "pmc_e2e_fetch", "pmc_e2e_write", BLOCK + "<true", None, conv_n, "samples", conv_n * 160 * 160 * 64 * 4 * 1.25,
        "the block's 16-channel input in, its 64-channel output out: N*160*160*64*4 B * 1.25")
if fused:
    out["conv_block"] = section(
        "conv_block32s_kernel / conv_block32_kernel<false> (fp16x2), a stage-2 residual block of %d samples in one launch (two 3x3 convs 64->64 ch, 160x160)" % conv_n,
        "pmc_e2e_fetch", "pmc_e2e_write", (BLOCKS, BLOCK + "<false"), None, conv_n, "samples", conv_n * 160 * 160 * 64 * 4 * 2.0,
        "the block's input in, its output out: N*160*160*64*4 B * 2 (halo re-reads and the residual are L2 hits by design)")
RW3 = "conv_rw_kernel<1, 2, 16"
if rows("pmc_e2e_fetch", "FETCH_SIZE", RW3):
    # round 6: four of the five stage-3 launches of this shape per forward run conv_rw_kernel<1, 2, 16, ...> (two with a BatchNorm
    # prologue, two with a residual), the one that carries the stage's 1x1 shortcut stays on conv_bf3w_kernel: the mean over all
    # of them is the per-launch traffic of bench.py's `conv_stage3` leg (its time is the mean over the same five)
    out["conv_stage3"] = section(
        "conv_rw_kernel<1, 2, 16, *, *> x 4 + conv_bf3w_kernel<false, true, 2, 1, 2, true, false> x 1 per forward (fp16x2), stage-3 "
        "launches of %d samples (128->128 ch, 80x80)" % conv_n, "pmc_e2e_fetch", "pmc_e2e_write", (RW3, CONV3), None, conv_n, "samples",
        conv_n * 80 * 80 * 128 * 4 * 2.6,
        "input + output (+ residual in 3 of the 5 stage-3 convolutions of this shape): N*80*80*128*4 B * 2.6")
    for name, sub, mult, note in (("conv_stage3_rw_bn", RW3 + ", true, false>", 2.0, "input + output: N*80*80*128*4 B * 2"),
                                  ("conv_stage3_rw_res", RW3 + ", false, true>", 3.0, "input + residual + output: N*80*80*128*4 B * 3")):
        if rows("pmc_e2e_fetch", "FETCH_SIZE", sub):
            out[name] = section(sub.replace("conv_rw_kernel", "conv_rw_kernel") + " (fp16x2), launches of %d samples" % conv_n,
                                "pmc_e2e_fetch", "pmc_e2e_write", sub, None, conv_n, "samples", conv_n * 80 * 80 * 128 * 4 * mult, note)
    for name, sub, algo, note in (
            ("conv_stride2_rw", "conv_rw_kernel<2, 1, 8", conv_n * (160 * 160 * 64 + 80 * 80 * 128) * 4,
             "the stride-2 first convolution of stage 3: N*(160*160*64 + 80*80*128)*4 B"),
            ("conv_stride3_rw", "conv_rw_kernel<3, 2, 4", conv_n * (80 * 80 * 128 + 27 * 27 * 256) * 4,
             "the stride-3 first convolution of stage 4: N*(80*80*128 + 27*27*256)*4 B")):
        if rows("pmc_e2e_fetch", "FETCH_SIZE", sub):
            out[name] = section(sub + ", true, false> (fp16x2), launches of %d samples" % conv_n, "pmc_e2e_fetch", "pmc_e2e_write", sub,
                                None, conv_n, "samples", algo, note)
else:
    out["conv_stage3"] = section(
        "conv_bf3w_kernel<false, true, 2, 1, 2, true, false> (fp16x2), stage-3 launches of %d samples (128->128 ch, 80x80)" % conv_n, "pmc_e2e_fetch",
        "pmc_e2e_write", CONV3, grid3, conv_n, "samples", conv_n * 80 * 80 * 128 * 4 * 2.6,
        "input + output (+ residual in 3 of the 5 stage-3 convolutions of this shape): N*80*80*128*4 B * 2.6")
# one launch walks CLIPS clips through their 270 frames (bench.py defaults): clip-frames per launch
FRAMES = int(os.environ.get("CPX_BENCH_FRAMES", "270"))
out["frame_kernel_e2e"] = section(
    "cpx_frame_kernel, one launch = %d clips x %d frames, no label image (end-to-end configuration)" % (CLIPS, FRAMES), "pmc_e2e_fetch",
    "pmc_e2e_write", "cpx_frame_kernel", max(g for g, _ in rows("pmc_e2e_fetch", "FETCH_SIZE", "cpx_frame_kernel")),  # (the BatchNorm calibration of the synthetic network tracks 96 clips first: not that launch)
    CLIPS * FRAMES, "clip-frames", (614400 - 76800) * CLIPS * FRAMES,
    "SURVEY 8(d): 614,400 B per frame minus the 76,800 B label image")
if glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_trk_fetch", "*", "*_counter_collection.csv")):
    out["frame_kernel_track"] = section(
        "cpx_frame_kernel, one launch = %d clips x %d frames, label image written (BASELINE configs[1])" % (CLIPS, FRAMES), "pmc_trk_fetch",
        "pmc_trk_write", "cpx_frame_kernel", None, CLIPS * FRAMES, "clip-frames", 614400 * CLIPS * FRAMES, "SURVEY 8(d): 614,400 B per frame")
json.dump(out, open(os.path.join(ROOT, "profiles", ROUND + "_e2e_pmc.json"), "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, "traffic/algorithmic = %.3f" % v["traffic_over_algorithmic"])
