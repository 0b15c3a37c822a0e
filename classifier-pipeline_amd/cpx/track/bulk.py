"""The file-fed extract path at device speed: what TrackExtractor.extract(directory) runs
(reference src/track/trackextractor.py:60-120 forks a pool of CPU workers over the files; here the files of a directory
go through ONE device in large batches and no per-frame or per-region Python object is built on the way).

Per batch of recordings:
  host     the files' bytes are read into pinned memory by a thread pool (readinto: no copy, no GIL), the next batch
           while the device works on this one
  device   cpx_cptv_inflate (gzip + DEFLATE + section walk, one wavefront per file) -> cpx_cptv_gather_index ->
           cpx_cptv_unpack -> cpx_track_batch -> cpx_associate_batch -> cpx_finalize_tracks (trim, statistics, score
           order, rejects) -> the kept tracks' regions gathered into one array -> cpx_thumb_stats for all of them
  host     per file: the thumbnail ranking (a few NumPy operations per track), the metadata JSON with the positions
           written by cpx_format_regions, one file write
A recording that fails anywhere (unreadable, corrupt gzip, malformed section, another resolution than its group's
engine, capacity overflow) is logged and retried alone through extract_file; when that fails too it is skipped -- the
other recordings of the batch are unaffected.

The metadata equals extract_file's field by field (tests/test_batch_files_gpu.py): the device's end-of-clip statistics
are bit-identical to the host's (scratch/final_exact_probe.py, 422 tracks), the thumbnail scores are computed by the
same float64 operations in the same order."""

import ctypes as C
import json
import logging
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from datetime import datetime, timedelta

import numpy as np

from .._lib import (CPTV_BACKGROUND_FRAME, CPTV_FILE_DTYPE, CPTV_HAS_LAST_FFC, CPTV_HAS_TIME_ON, CPTV_HEADER_BYTES,
                    CPTV_RESULT_DTYPE, CPTV_SLOT_DTYPE, CPTV_STATUS, FRAME_INFO_DTYPE, FRAME_META_DTYPE,
                    REGION_REF_DTYPE, THUMB_STAT_DTYPE, CpxError)
from ..ml_tools import tools
from ..tracking import (REGION_DTYPE, TRACK_SUMMARY_DTYPE, make_filter_params, make_track_params)
from .clip import Clip

THUMBNAIL_SIZE = 64


class StagedFiles:
    """A batch of files in pinned host memory."""

    def __init__(self, paths, stage, in_off, sizes, isize, errors):
        self.paths, self.stage, self.in_off, self.sizes, self.isize, self.errors = paths, stage, in_off, sizes, isize, errors


class FileStager:
    """Reads batches of files into (two alternating) pinned buffers with a thread pool; stage(paths) returns a future."""

    def __init__(self, torch, threads=None):
        self.torch = torch
        from ..sharding import host_threads_per_rank

        self.threads = threads or host_threads_per_rank(16)   # the node's CPU quota shared among its ranks
        self.pool = ThreadPoolExecutor(max_workers=self.threads)
        self.driver = ThreadPoolExecutor(max_workers=1)
        self.buffers = [None, None]
        self.turn = 0

    def close(self):
        self.pool.shutdown(wait=True)
        self.driver.shutdown(wait=True)

    def stage(self, paths):
        slot = self.turn
        self.turn ^= 1
        return self.driver.submit(self._stage, list(paths), slot)

    def _stage(self, paths, slot):
        t = self.torch
        n = len(paths)
        sizes = np.zeros(n, np.int64)
        errors = {}
        for i, p in enumerate(paths):
            try:
                sizes[i] = os.path.getsize(p)
            except OSError as e:
                errors[i] = "%s: %s" % (p, e)
        in_off = np.zeros(n + 1, np.int64)
        np.cumsum((sizes + 15) & ~15, out=in_off[1:])
        need = int(in_off[-1]) + 16
        buf = self.buffers[slot]
        if buf is None or buf.numel() < need:
            buf = t.empty(max(need, 1 << 20), dtype=t.uint8, pin_memory=True)
            self.buffers[slot] = buf
        view = memoryview(buf.numpy())
        isize = np.zeros(n, np.int64)

        def read(i):
            if i in errors or sizes[i] == 0:
                return
            try:
                with open(paths[i], "rb", buffering=0) as fh:
                    got = fh.readinto(view[in_off[i]:in_off[i] + sizes[i]])
                if got != sizes[i]:
                    errors[i] = "%s: short read (%d of %d bytes)" % (paths[i], got, sizes[i])
                    return
                if sizes[i] >= 18:
                    v = int.from_bytes(view[in_off[i] + sizes[i] - 4:in_off[i] + sizes[i]], "little")
                    # DEFLATE cannot expand by more than 1032 : 1: a larger figure is not a gzip trailer
                    isize[i] = v if v <= 1032 * sizes[i] + 64 else 0
            except OSError as e:
                errors[i] = "%s: %s" % (paths[i], e)

        list(self.pool.map(read, range(n)))
        return StagedFiles(paths, buf, in_off, sizes, isize, errors)


def stage_blobs(torch, blobs):
    """The same staging for byte strings already in memory (bench.py from_files, tests)."""
    n = len(blobs)
    sizes = np.array([len(b) for b in blobs], np.int64)
    in_off = np.zeros(n + 1, np.int64)
    np.cumsum((sizes + 15) & ~15, out=in_off[1:])
    buf = torch.empty(int(in_off[-1]) + 16, dtype=torch.uint8, pin_memory=True)
    view = buf.numpy()
    isize = np.zeros(n, np.int64)
    def copy(i):
        b = blobs[i]
        view[in_off[i]:in_off[i] + sizes[i]] = np.frombuffer(b, np.uint8)   # (NumPy releases the GIL for the copy)
        if sizes[i] >= 18:
            v = int.from_bytes(bytes(b[-4:]), "little")
            isize[i] = v if v <= 1032 * sizes[i] + 64 else 0

    if n >= 64:
        with ThreadPoolExecutor(max_workers=8) as pool:
            list(pool.map(copy, range(n)))
    else:
        for i in range(n):
            copy(i)
    return StagedFiles([None] * n, buf, in_off, sizes, isize, {})


class DecodedBatch:
    """Frames of the files that decoded, grouped by what one engine can track together."""

    def __init__(self):
        self.errors = {}      # file index -> message
        self.groups = []      # DecodedGroup
        self.results = None


class DecodedGroup:
    def __init__(self, key, files, headers, offs, slots, frames_dev):
        self.key = key                # (width, height, camera model)
        self.files = files            # indices into the staged batch
        self.headers = headers        # CptvHeader per file
        self.offs = offs              # int32 [len(files) + 1] frame ranges inside frames_dev
        self.slots = slots            # CPTV_SLOT_DTYPE [total]
        self.frames_dev = frames_dev  # uint16 bits [total, H, W]

    def split(self, max_clips, max_frames=None):
        """The group in runs of at most max_clips recordings and (when given) max_frames frames -- a single recording
        longer than that still forms a run of its own -- as views of the same device frames: the decode stage likes
        thousands of files per launch (one wavefront each: its time hardly grows until the chip is full), the tracking
        stage is sized by the memory its per-frame outputs take (label + filtered image + components: ~190 KB per
        160 x 120 frame)."""
        n = len(self.files)
        total = int(self.offs[-1])
        if n <= max_clips and (max_frames is None or total <= max_frames):
            yield self
            return
        b0 = 0
        while b0 < n:
            b1 = min(n, b0 + max_clips)
            if max_frames is not None:
                # the longest prefix within the frame budget (at least one recording)
                fit = int(np.searchsorted(self.offs[b0 + 1:b1 + 1], int(self.offs[b0]) + max_frames, side="right"))
                b1 = b0 + max(fit, 1)
            f0, f1 = int(self.offs[b0]), int(self.offs[b1])
            yield DecodedGroup(self.key, self.files[b0:b1], self.headers[b0:b1], (self.offs[b0:b1 + 1] - f0).astype(np.int32),
                               self.slots[f0:f1], self.frames_dev[f0:f1])
            b0 = b1


def plan_decode_batches(sizes, batch_files, max_bytes):
    """[(first, end)] runs of recordings for the decode launches: at most batch_files recordings and max_bytes of
    compressed data each (a larger single recording forms a run of its own); the FIRST run half as large when there
    are several -- nothing runs beside it, and a launch's time hardly falls below a lone wave's, so the pipeline fills
    sooner than the launch loses in occupancy."""
    n = len(sizes)
    runs = []
    a = 0
    first = batch_files // 2 if (n > 2 * batch_files and batch_files >= 512) else batch_files
    cum = np.concatenate([[0], np.cumsum(np.asarray(sizes, np.int64))])
    while a < n:
        cap = first if not runs else batch_files
        b = min(n, a + cap)
        fit = int(np.searchsorted(cum[a + 1:b + 1], cum[a] + max_bytes, side="right"))
        b = a + max(fit, 1)
        runs.append((a, b))
        a = b
    return runs


MAX_INFLATED = 1 << 30  # bytes one recording may inflate to on the batched path (a ten-minute 160 x 120 clip is ~0.2 GB)


def decode_staged(eng, staged, min_pixels=160 * 120, unpack_engine=None):
    """cpx_cptv_inflate over a staged batch, then per (resolution, camera model) group gather + unpack.  `eng`: any
    engine on the device (the inflate kernel does not depend on its geometry; the unpack of a group runs on an engine
    of that group's resolution, created on demand)."""
    return inflate_finish(eng, staged, inflate_launch(eng, staged, min_pixels), unpack_engine)


def inflate_launch(eng, staged, min_pixels=160 * 120):
    """First half of decode_staged: output / slot layout, device buffers and the cpx_cptv_inflate launch on `eng`'s
    stream -- no wait.  -> the context inflate_finish takes (an event marks the launch's end, so a further launch
    enqueued behind it on the same stream does not delay the wait)."""
    t, dev = eng.torch, eng.device
    n = len(staged.sizes)
    min_frame = 4 + (min_pixels - 1 + 7) // 8 + 8
    files = np.zeros(n, CPTV_FILE_DTYPE)
    sizes = np.where(np.isin(np.arange(n), list(staged.errors)), 0, staged.sizes)
    files["in_offset"], files["in_bytes"] = staged.in_off[:-1], sizes
    isize = np.where(sizes > 0, staged.isize, 0)
    # ISIZE is the last four bytes of an untrusted file: a claim beyond MAX_INFLATED (or beyond what DEFLATE can expand
    # the file to, 1032 x) does not size device buffers -- that recording goes to the one-file reader, which needs no
    # size up front
    absurd = (isize > MAX_INFLATED) | (isize > 1032 * np.maximum(sizes, 1) + 64)
    for i in np.nonzero(absurd)[0]:
        staged.errors.setdefault(int(i), "%s: gzip trailer claims %d inflated bytes" % (staged.paths[i], int(isize[i])))
    sizes = np.where(absurd, 0, sizes)
    isize = np.where(absurd, 0, isize)
    files["in_bytes"] = sizes
    out_off = np.zeros(n + 1, np.int64)
    np.cumsum(((isize + 15) & ~15) + 16, out=out_off[1:])
    files["out_offset"], files["out_capacity"] = out_off[:-1], isize
    slot_cap = isize // min_frame + 1
    slot_off = np.zeros(n + 1, np.int64)
    np.cumsum(slot_cap, out=slot_off[1:])
    files["slot_offset"], files["slot_capacity"] = slot_off[:-1], slot_cap
    n_in = int(staged.in_off[-1]) + 16
    in_dev = getattr(staged, "in_dev", None)
    if in_dev is not None:     # uploaded ahead (run_files_bulk's staging thread, a stream of its own)
        staged.in_event.synchronize()
    else:
        in_dev = staged.stage[:n_in].to(dev, non_blocking=True)
    files_dev = eng._to_dev(files)
    out_dev = t.empty(int(out_off[-1]) + 16, dtype=t.uint8, device=dev)
    slots_dev = t.empty(max(int(slot_off[-1]), 1) * 8, dtype=t.int32, device=dev)
    header_dev = t.empty((n, CPTV_HEADER_BYTES), dtype=t.uint8, device=dev)
    results_dev = t.zeros(n * 10, dtype=t.int32, device=dev)
    p = lambda x: C.c_void_p(x.data_ptr())
    eng.sync_inputs()
    rc = eng.lib.cpx_cptv_inflate(eng.h, p(in_dev), p(files_dev), n, p(out_dev), p(slots_dev), p(header_dev),
                                  p(results_dev))
    if rc != 0:
        raise CpxError(rc, eng._err())
    done = t.cuda.Event()
    done.record(eng.torch_stream())
    return dict(n=n, slot_off=slot_off, in_dev=in_dev, files_dev=files_dev, out_dev=out_dev, slots_dev=slots_dev,
                header_dev=header_dev, results_dev=results_dev, done=done)


def inflate_finish(eng, staged, ctx, unpack_engine=None):
    """Second half: wait for the launch, read the per-file results, group the files that decoded by (resolution, camera
    model) and run gather + unpack per group -- on `unpack_engine`'s stream when given (run_files_bulk: the NEXT batch's
    inflate is already running on `eng`'s)."""
    from ..cptv import parse_header_bytes

    t, dev = eng.torch, eng.device
    n, slot_off = ctx["n"], ctx["slot_off"]
    out_dev, slots_dev, header_dev, results_dev = ctx["out_dev"], ctx["slots_dev"], ctx["header_dev"], ctx["results_dev"]
    p = lambda x: C.c_void_p(x.data_ptr())
    out = DecodedBatch()
    out.errors.update(staged.errors)
    ctx["done"].synchronize()
    res = results_dev.cpu().numpy().view(CPTV_RESULT_DTYPE).reshape(-1)
    out.results = res
    hdr = header_dev.cpu().numpy()
    groups = {}
    headers = {}
    for i in range(n):
        if i in out.errors:
            continue
        st = int(res["status"][i])
        if st != 0:
            out.errors[i] = "%s: %s" % (staged.paths[i], CPTV_STATUS.get(st, "status %d" % st))
            continue
        if int(res["header_bytes"][i]) > CPTV_HEADER_BYTES:
            out.errors[i] = "%s: header section of %d bytes" % (staged.paths[i], res["header_bytes"][i])
            continue
        try:
            h = parse_header_bytes(hdr[i].tobytes())
        except Exception as e:  # a header the host parser refuses: the file goes the slow way
            out.errors[i] = "%s: %s" % (staged.paths[i], e)
            continue
        headers[i] = h
        groups.setdefault((int(res["width"][i]), int(res["height"][i]), h.model if h.model else None), []).append(i)
    for key, members in groups.items():
        # a group that cannot be unpacked (a resolution no engine takes, a kernel refusal, no memory for its frames)
        # sends ITS members to the one-file path; the other groups of the batch carry on
        try:
            out.groups.append(_unpack_group(eng, unpack_engine, key, members, headers, res, slot_off, slots_dev, out_dev))
        except Exception as e:  # noqa: BLE001 -- fault isolation
            if isinstance(e, t.cuda.OutOfMemoryError):
                t.cuda.empty_cache()
            for i in members:
                out.errors[i] = "%s: %s: %s" % (staged.paths[i], type(e).__name__, str(e).splitlines()[0] if str(e) else "")
    return out


def _unpack_group(eng, unpack_engine, key, members, headers, res, slot_off, slots_dev, out_dev):
    from .cliptrackextractor import get_engine

    t, dev = eng.torch, eng.device
    p = lambda x: C.c_void_p(x.data_ptr())
    W, H, _ = key
    ok = np.asarray(members)
    offs = np.zeros(len(ok) + 1, np.int32)
    np.cumsum(res["n_frames"][ok], out=offs[1:])
    total = int(offs[-1])
    # any engine of this resolution can unpack; the tracking engine is picked by the caller
    base = unpack_engine or eng
    ueng = base if (base.width, base.height) == (W, H) else get_engine(W, H, 20.0, 0.1, device=dev.index or 0)
    offs_dev = t.from_numpy(offs).to(dev)
    so_dev = t.from_numpy(np.ascontiguousarray(slot_off[:-1][ok])).to(dev)
    fo_dev = t.empty(total, dtype=t.int64, device=dev)
    bw_dev = t.empty(total, dtype=t.int32, device=dev)
    dense_dev = t.empty(total * 8, dtype=t.int32, device=dev)
    frames_dev = t.empty((total, H, W), dtype=t.int16, device=dev)
    ueng.sync_inputs()
    rc = ueng.lib.cpx_cptv_gather_index(ueng.h, p(slots_dev), p(so_dev), p(offs_dev), len(ok), p(fo_dev), p(bw_dev),
                                        p(dense_dev))
    if rc == 0:
        rc = ueng.lib.cpx_cptv_unpack(ueng.h, p(out_dev), p(fo_dev), p(bw_dev), p(offs_dev), len(ok), p(frames_dev))
    if rc != 0:
        raise CpxError(rc, ueng._err())
    ueng.synchronize()
    slots = dense_dev.cpu().numpy().view(CPTV_SLOT_DTYPE).reshape(-1)
    return DecodedGroup(key, members, [headers[i] for i in members], offs, slots, frames_dev)


def frame_meta_from_slots(slots, process_background=False):
    """cpx_frame_meta [total] for the track stage from the section index."""
    m = np.zeros(len(slots), FRAME_META_DTYPE)
    fl = slots["flags"]
    has = ((fl & CPTV_HAS_TIME_ON) != 0) & ((fl & CPTV_HAS_LAST_FFC) != 0)
    m["time_on_ms"] = np.where(has, slots["time_on_ms"], 0)
    m["last_ffc_ms"] = np.where(has, slots["last_ffc_ms"], 0)
    m["has_times"] = has
    if not process_background:
        m["background_frame"] = (fl & CPTV_BACKGROUND_FRAME) != 0
    return m


class BulkTracker:
    """Tracks decoded groups and writes the metadata; one instance per TrackExtractor.extract call."""

    def __init__(self, config, device=0):
        from .cliptrackextractor import ClipTrackExtractor

        self.config = config
        self.device = device
        # a real extractor supplies version / config exactly as the one-file path reports them
        self.extractor = ClipTrackExtractor(config.tracking, config.use_opt_flow, False, verbose=config.verbose,
                                            device=device)
        self.tcfg = self.extractor.config
        self.lib = None
        self.decode_engine = None   # a handle (= HIP stream) of its own for the decode stage of run_files_bulk
        self.cnn_chunk = 2048
        self._algorithm_text = {}
        self.timings = {"decode_s": 0.0, "device_s": 0.0, "host_s": 0.0, "write_s": 0.0, "files": 0, "frames": 0}

    # ---- device ----------------------------------------------------------------------------------------------
    def track_group(self, group, clips, classifiers=(), lane=0):
        """-> dict of host arrays for the group's clips (clips[k]: the Clip object of group.files[k]).
        classifiers: [(model config, interpreter)] -- every kept track's segments (cpx_plan_segments: the reference's
        get_segments under identity draws) are cropped, tiled and run through each model's network."""
        from ..pipeline import BatchPipeline
        from .cliptrackextractor import get_engine

        cfg = self.tcfg
        c0 = clips[0]
        W, H, _ = group.key
        weight_add = (1 if c0.camera_model == "lepton3.5" else 0.1) / self.extractor.weighting_percent
        longest = int(np.diff(group.offs).max())
        eng = get_engine(W, H, c0.background_thresh, weight_add, cfg.edge_pixels, self.device, max_frames=longest,
                         denoise=bool(cfg.denoise), lane=lane)
        # Everything this group allocates and every torch operation on it runs with the engine's stream as torch's
        # current stream: the caching allocator hands a freed block only to allocations on the stream it was made for,
        # so two lanes (two threads, two engines) never receive each other's blocks while the other's kernels -- which
        # torch does not know about: they are cpx launches on the handle's stream -- may still be using them
        with eng.torch.cuda.stream(eng.torch_stream()):
            return self._track_group(eng, group, clips, classifiers, cfg, c0, W, H)

    def _track_group(self, eng, group, clips, classifiers, cfg, c0, W, H):
        from ..pipeline import BatchPipeline

        t, dev = eng.torch, eng.device
        self.lib = eng.lib
        offs = group.offs
        B = len(offs) - 1
        total = int(offs[-1])
        meta = frame_meta_from_slots(group.slots)
        tp = make_track_params(c0.res_x, c0.res_y, cfg.edge_pixels, cfg.frame_padding, self.extractor.min_dimension,
                               cfg.cropped_regions_strategy, cfg.filter_regions_pre_match, cfg.aoi_min_mass,
                               cfg.aoi_pixel_variance, cfg.params, c0.frames_per_second)
        fp = make_filter_params(cfg.min_duration_secs, cfg.track_min_offset, cfg.track_min_mass, c0.track_min_delta,
                                c0.track_max_delta, cfg.min_moving_frames, cfg.max_blank_percent, cfg.max_jitter,
                                c0.frames_per_second, cfg.max_tracks, tp.max_active_tracks, tp.max_tracks)
        mt, ma = tp.max_tracks, tp.max_active_tracks
        sq = classifiers[0][1].params.square_width if classifiers else 5
        comps = t.empty(total * eng.cap * 8, dtype=t.int32, device=dev)
        info_dev = t.empty(total * 20, dtype=t.int32, device=dev)
        labels = t.empty((total, H, W), dtype=t.int32, device=dev)
        filt = t.empty((total, H, W), dtype=t.float32, device=dev) if classifiers else None
        lflags = classifiers[0][1].limits_flags() if classifiers else 0
        for _, interp in classifiers:
            if interp.params.square_width == 1:
                raise NotImplementedError("single-frame models go through ClipClassifier.process_files")
            if interp.limits_flags() != lflags:
                raise NotImplementedError("models with different normalisation variants in one run")
        pipe = BatchPipeline(eng, None, square_width=sq, track_params=tp, filter_params=fp, want_regions=True,
                             limits_flags=lflags)
        model_out = []
        prof = self.timings.setdefault("device_split_s", {}) if os.environ.get("CPX_BULK_PROFILE") else None
        tick = [time.time()]

        def lap(name):  # CPX_BULK_PROFILE=1: wall time per stage, with a device synchronisation at every boundary
            if prof is not None:
                eng.synchronize()
                t.cuda.synchronize(dev)
                now = time.time()
                prof[name] = prof.get(name, 0.0) + now - tick[0]
                tick[0] = now

        t.cuda.current_stream(dev).synchronize()
        lap("alloc")
        with t.cuda.stream(eng.torch_stream()):
            front = pipe._front(group.frames_dev, offs, meta, (comps, info_dev, labels, filt, None),
                                classify=bool(classifiers))
            lap("track_assoc_finalize_plan")
            for model, interp in classifiers:
                if interp.params.square_width != sq:
                    raise NotImplementedError("models with different square_width in one run")
                if interp.run_over_network or getattr(interp, "_weights", None) is None:
                    raise NotImplementedError("a model served over the network has no device network for the batched "
                                              "forward: classify such models per file (ClipClassifier.process_files)")
                fpi = interp.labels.index("false-positive") if "false-positive" in interp.labels else -1
                mp = BatchPipeline(eng, interp._network(eng), n_labels=len(interp.labels), fp_index=fpi,
                                   frame_size=interp.params.frame_size, square_width=sq, track_params=tp,
                                   filter_params=fp, cnn_chunk=self.cnn_chunk)
                t0 = time.time()
                probs = None
                if front.n_tracks and front.n_samples:
                    mp.classify_front(front, group.frames_dev)
                    probs = front.probs.cpu().numpy()
                model_out.append(dict(model=model, interp=interp, probs=probs, seconds=time.time() - t0))
            eng.synchronize()
        lap("crop_network")
        res, assoc = front.track, front.assoc
        info = info_dev.cpu().numpy().view(FRAME_INFO_DTYPE).reshape(-1)
        bad = np.nonzero((info["frame_number"] >= 0) & (info["status"] != 0))[0]
        ntr = assoc.ntracks_dev.cpu().numpy()
        astatus = assoc.status_dev.cpu().numpy()
        summ = front.summaries(mt)
        # clips whose capacities overflowed (components per frame, tracks per clip) go the slow way
        failed = {}
        clip_of_frame = np.repeat(np.arange(B), np.diff(offs))
        for f in bad:
            failed[int(clip_of_frame[f])] = "frame %d: more than %d components" % (int(f - offs[clip_of_frame[f]]), eng.cap)
        for b in np.nonzero(astatus != 0)[0]:
            failed[int(b)] = "track capacity exceeded"
        # ---- kept tracks in score order; their regions gathered on the device ----
        proc_mask = info["frame_number"] >= 0
        kept = []       # (clip, summary row) in output order = the order of the segment plan
        kept_pipe = []  # index of the track in the pipeline's arrays (clips that failed keep their slots there)
        prefix = np.concatenate([[0], np.cumsum(front.counts[:, 0])])
        for b in range(B):
            s = summ[b, : int(ntr[b])]
            k = np.nonzero(s["reject"] == 0)[0]
            if len(k) and b not in failed:
                k = k[np.argsort(s["rank"][k], kind="stable")]
                kept.extend((b, int(j)) for j in k)
                kept_pipe.extend(int(prefix[b]) + q for q in range(len(k)))
        rows = []
        tr_off = [0]
        for b, j in kept:
            s = summ[b, j]
            n = int(s["n_frames"])
            rows.append((int(offs[b]) + int(s["start_frame"]) + np.arange(n, dtype=np.int64)) * ma + int(s["slot"]))
            tr_off.append(tr_off[-1] + n)
        regions = np.zeros(0, REGION_DTYPE)
        if rows:
            idx = t.from_numpy(np.concatenate(rows)).to(dev)
            pool = assoc.pool_dev.view(-1, 14)
            regions = pool[idx].cpu().numpy().view(REGION_DTYPE).reshape(-1)
        lap("summaries_and_region_gather")
        # ---- processed-frame index per clip: frame number q -> index in the batch ----
        proc_idx = [np.nonzero(proc_mask[offs[b]:offs[b + 1]])[0] + int(offs[b]) for b in range(B)]
        # ---- thumbnails of the kept tracks: one cpx_thumb_stats over every usable region ----
        tr_off = np.asarray(tr_off, np.int64)
        stats = np.zeros(0, THUMB_STAT_DTYPE)
        usable = np.zeros(0, np.int64)
        if len(regions):
            clip_of_reg = np.repeat(np.array([b for b, _ in kept], np.int64), np.diff(tr_off))
            usable = np.nonzero(((regions["flags"] & 1) == 0) & (regions["mass"] != 0))[0]
            if len(usable):
                refs = np.zeros(len(usable), REGION_REF_DTYPE)
                r = regions[usable]
                fidx = np.empty(len(usable), np.int64)
                cb = clip_of_reg[usable]
                for b in np.unique(cb):
                    sel = cb == b
                    fidx[sel] = proc_idx[b][r["frame_number"][sel]]
                refs["frame"], refs["x"], refs["y"], refs["width"], refs["height"] = fidx, r["x"], r["y"], r["width"], r["height"]
                stats = eng.thumb_stats(group.frames_dev, res, refs)
        lap("thumbnail_kernel")
        # ---- clips without a kept track: the heaviest region ever seen, else the window search ----
        kept_clips = set(b for b, _ in kept)
        trackless = [b for b in range(B) if b not in kept_clips and b not in failed]
        best_region = {}
        if trackless:
            best_region = self._trackless(eng, group, res, assoc, info, offs, trackless, proc_idx)
        lap("trackless")
        # ---- classification: the samples of every kept track (frame numbers per segment, for the metadata) ----
        samples = None
        if classifiers and front.n_samples:
            from .._lib import CROP_REQ_DTYPE

            per = sq * sq
            reqs = front.reqs_dev.cpu().numpy().view(CROP_REQ_DTYPE).reshape(-1, per)
            samples = dict(sample_track=front.sample_track_dev.cpu().numpy(), frames=reqs["frame"])
        # (no device buffer leaves this function: label / filtered images of a thousand recordings are tens of GB, and
        # the next group's are allocated while the host still formats this one's metadata)
        return dict(info=info, summ=summ, ntr=ntr, kept=kept, kept_pipe=kept_pipe, tr_off=tr_off, regions=regions,
                    usable=usable, stats=stats, failed=failed, best_region=best_region, proc_idx=proc_idx,
                    model_out=model_out, samples=samples)

    def _trackless(self, eng, group, res, assoc, info, offs, clips, proc_idx):
        """best_trackless_thumb (classify/thumbnail.py:13-64) for the clips `clips`: the first region of maximal mass
        in (frame, region) order over the clip's region history, else the 64 x 64 window search on the frame of
        maximal mean."""
        t, dev = eng.torch, eng.device
        cap = eng.cap
        out = {}
        regs = assoc.regions_dev.view(-1, cap, 14)
        rc_dev = assoc.rcounts_dev
        ffc = res.info_dev.view(-1, 20)[:, 3]
        valid = (t.arange(cap, device=dev)[None, :] < rc_dev[:, None]) & (ffc[:, None] == 0)
        mass = t.where(valid, regs[:, :, 4], t.full((), -1, dtype=t.int32, device=dev))
        fmax, farg = mass.max(dim=1)   # per frame: heaviest region (first of equals)
        tmax = int(max(offs[b + 1] - offs[b] for b in clips))
        pad_idx = np.full((len(clips), tmax), -1, np.int64)
        for k, b in enumerate(clips):
            pad_idx[k, : offs[b + 1] - offs[b]] = np.arange(offs[b], offs[b + 1])
        pi = t.from_numpy(pad_idx).to(dev)
        m = t.where(pi >= 0, fmax[pi.clamp(min=0)], t.full((), -1, dtype=fmax.dtype, device=dev))
        best_f = m.argmax(dim=1)       # first frame of maximal mass
        best_m = m.gather(1, best_f[:, None])[:, 0]
        frame_abs = pi.gather(1, best_f[:, None])[:, 0]
        rsel = regs[frame_abs, farg[frame_abs]]
        best_m_h = best_m.cpu().numpy()
        rsel_h = rsel.cpu().numpy().view(REGION_DTYPE).reshape(-1)
        search = []
        for k, b in enumerate(clips):
            if best_m_h[k] >= 0:
                out[b] = ("region", rsel_h[k])
            else:
                search.append(b)
        if search:
            P = eng.width * eng.height
            pairs, rows = [], []
            for b in search:
                pf = proc_idx[b]
                if len(pf) == 0:
                    out[b] = ("none", None)
                    continue
                q = int(np.argmax(info["thermal_sum"][pf] / P))
                pairs.append((int(pf[q]), int(offs[b])))
                rows.append((b, q))
            if pairs:   # one launch: a workgroup per recording
                pairs_dev = t.from_numpy(np.asarray(pairs, np.int32)).to(dev)
                xy = t.zeros((len(pairs), 2), dtype=t.int32, device=dev)
                eng.sync_inputs()
                rc = eng.lib.cpx_trackless_thumb_batch(eng.h, C.c_void_p(group.frames_dev.data_ptr()),
                                                       C.c_void_p(pairs_dev.data_ptr()), len(pairs), C.c_void_p(xy.data_ptr()))
                if rc != 0:
                    raise CpxError(rc, eng._err())
                eng.synchronize()
                xy_h = xy.cpu().numpy()
                for k, (b, q) in enumerate(rows):
                    out[b] = ("window", (q, int(xy_h[k, 0]), int(xy_h[k, 1])))
        return out

    # ---- host: metadata ---------------------------------------------------------------------------------------
    def _fmt_regions(self, regs, indent, depth, as_list=True):
        n = len(regs)
        cap = 512 + 420 * max(n, 1)
        buf = C.create_string_buffer(cap)
        got = self.lib.cpx_format_regions(C.c_void_p(regs.ctypes.data), n, REGION_DTYPE.itemsize, indent, depth,
                                          1 if as_list else 0, buf, cap)
        if got < 0:
            buf = C.create_string_buffer(-got + 16)
            got = self.lib.cpx_format_regions(C.c_void_p(regs.ctypes.data), n, REGION_DTYPE.itemsize, indent, depth,
                                              1 if as_list else 0, buf, -got + 16)
        return buf.raw[:got].decode("ascii")

    def thumbnail_of(self, regs, use, st):
        """get_thumbnail_info (classify/thumbnail.py:137-197) for one track: regs = its regions, use = indices of the
        usable ones, st = their cpx_thumb_stat.  -> (region record, contours, median_diff, score) or None."""
        if len(regs) == 0:
            return None
        if len(use):
            keep = st["contours"] != 0
            use, st = use[keep], st[keep]
        if len(use) == 0:
            return regs[0], 0, 0, 0
        r = regs[use]
        mass = r["mass"].astype(np.int64)
        contours = st["contours"].astype(np.int64)
        md = st["median_diff"]
        max_mass = max(0, int(mass.max()))
        max_contour = max(0, int(contours.max()))
        max_md = max(0.0, float(md.max()))
        min_md = min(0.0, float(md.min()))
        mass_percent = mass / max_mass * 40
        pts = contours / max_contour * 50
        mid_x = r["x"] + r["width"] / 2
        mid_y = r["y"] + r["height"] / 2
        dx, dy = r["cx"] - mid_x, r["cy"] - mid_y
        centroid_mid = np.power(dx * dx + dy * dy, 0.5) * 2
        if max_md == 0:
            diff = np.zeros(len(r))
            if min_md != 0:
                diff = (md + abs(min_md)) / abs(min_md) * 40
        else:
            diff = md / max_md * 40
        total = mass_percent + pts + diff - centroid_mid
        border = (r["x"] <= 1) | (r["y"] <= 1) | (r["y"] + r["height"] >= 119) | (r["x"] + r["width"] >= 159)
        total = np.where(border, total - 1000, total)
        k = int(np.argmax(total))   # sorted(..., reverse=True)[0]: the first of equal scores
        return r[k], int(contours[k]), float(md[k]), float(total[k])

    def _dumps(self, obj, indent):
        """json.dumps(obj, indent=indent, cls=CustomJSONEncoder), character for character: CPython's C encoder (which
        only runs without an indent) + cpx_json_indent for the layout -- the indented form otherwise goes through the
        pure-Python encoder at about a microsecond per token (1.5 s per thousand recordings with predictions)."""
        text = json.dumps(obj, cls=tools.CustomJSONEncoder)
        if not indent:
            return text
        raw = text.encode("utf-8")
        cap = 4 * len(raw) + 256
        buf = C.create_string_buffer(cap)
        got = self.lib.cpx_json_indent(raw, len(raw), indent, 0, buf, cap)
        if got < 0:
            buf = C.create_string_buffer(-got + 16)
            got = self.lib.cpx_json_indent(raw, len(raw), indent, 0, buf, -got + 16)
        return buf.raw[:got].decode("utf-8")

    def algorithm_text(self, indent):
        """The "algorithm" entry is the same for every file of a run: encoded once."""
        if indent not in self._algorithm_text:
            d = {"tracker_version": self.extractor.tracker_version, "tracker_config": self.tcfg.as_dict()}
            self._algorithm_text[indent] = json.dumps(d, indent=indent or None, cls=tools.CustomJSONEncoder)
        return self._algorithm_text[indent]

    def metadata_text(self, clip, n_frames, tracks, trackless, source, tracking_time, existing, indent, models=None):
        """The JSON extract_file writes for this clip (trackextractor.get_metadata), as text.
        tracks: list of dict(summary, regions, thumb)."""
        head = {}
        if clip.camera_model:
            head["camera_model"] = clip.camera_model
        head["background_thresh"] = clip.background_thresh
        start = clip.video_start_time
        end = start + timedelta(seconds=n_frames / clip.frames_per_second)
        head["id"] = clip._id
        head["start_time"] = start.isoformat()
        head["end_time"] = end.isoformat()
        fps = clip.frames_per_second
        tlist = []
        subs = {}
        for k, tr in enumerate(tracks):
            s, regs = tr["summary"], tr["regions"]
            start_frame = int(s["start_frame"])
            end_frame = int(regs["frame_number"][-1])
            info = {"id": int(s["id"]), "tracker_version": self.extractor.tracker_version,
                    "start_s": round(start_frame / float(fps), 2), "end_s": round((end_frame + 1) / fps, 2),
                    "num_frames": len(regs), "frame_start": start_frame, "frame_end": end_frame,
                    "positions": "@@cpx-positions-%d@@" % k, "tracking_score": float(s["score"]),
                    "predictions": tr.get("predictions", [])}
            subs['"@@cpx-positions-%d@@"' % k] = self._fmt_regions(regs, indent, 4 if indent else 1)
            th = tr["thumb"]
            if th is None:
                info["thumbnail"] = None
            else:
                reg, contours, md, score = th
                info["thumbnail"] = {"region": "@@cpx-thumb-%d@@" % k, "contours": contours, "median_diff": md,
                                     "score": round(score)}
                subs['"@@cpx-thumb-%d@@"' % k] = self._fmt_regions(reg.reshape(1), indent, 5 if indent else 1, as_list=False)
            tlist.append(info)
        head["tracks"] = tlist
        if not tracks:
            head["thumbnail_region"] = trackless
        head["source"] = str(source)
        head["tracking_time"] = round(tracking_time, 1)
        head["algorithm"] = "@@cpx-algorithm@@"
        if existing is not None:
            existing.pop("tracks", None)
            existing.pop("Tracks", None)
            existing.update(head)
            head = existing
        if models is not None:  # ClipClassifier.save_metadata: one entry per model, with its classify time
            by_id = {m["id"]: m for m in head.get("models", [])}
            for d in models:
                by_id[d["id"]] = d
            head["models"] = list(by_id.values())
        text = self._dumps(head, indent)
        alg = self.algorithm_text(indent)
        if indent:  # nested one level deep: every line but the first moves right
            alg = alg.replace("\n", "\n" + " " * indent)
        subs['"@@cpx-algorithm@@"'] = alg
        for key, val in subs.items():
            text = text.replace(key, val, 1)
        return text


class _View:
    """Plain attribute bag: what the metadata stage reads of a Clip / interpreter / model, in a form that pickles."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


def clip_view(clip):
    return _View(camera_model=clip.camera_model, background_thresh=clip.background_thresh,
                 video_start_time=clip.video_start_time, frames_per_second=clip.frames_per_second, _id=clip._id)


def model_views(model_out):
    """model_out of track_group without the interpreter objects (they own device networks): per model the probabilities,
    the seconds, and what track_predictions reads of the interpreter and the model configuration."""
    out = []
    for mo in model_out:
        it = mo["interp"]
        out.append(dict(probs=mo["probs"], seconds=mo["seconds"], model=_View(id=mo["model"].id),
                        interp=_View(labels=list(it.labels), thresholds=it.thresholds,
                                     params=_View(smooth_predictions=it.params.smooth_predictions,
                                                  square_width=it.params.square_width))))
    return out


def format_group(tracker, job):
    """The metadata text of every recording of a tracked group: the host stage of the file-fed path as a function of plain
    data (`job`: what track_group returned + views of the clips / models), so that it runs in this process or in a
    metadata worker process (MetaPool) alike.  -> (texts {file index: text}, retry {file index: why}, frames, seconds)."""
    th = time.time()
    r, files, paths, offs = job["r"], job["files"], job["paths"], job["offs"]
    clips, existing, indent = job["clips"], job["existing"], job["indent"]
    texts, retry, frames = {}, {}, 0
    per_clip = {}
    for ti, (b, j) in enumerate(r["kept"]):
        per_clip.setdefault(b, []).append(ti)
    upos = np.searchsorted(r["usable"], r["tr_off"])   # usable-region index ranges per kept track
    n_kept = max(len(r["kept"]), 1)
    for b, i in enumerate(files):
        if b in r["failed"]:
            retry[i] = "%s: %s" % (paths[b], r["failed"][b])
            continue
        try:  # (fault isolation: a recording whose results do not serialise is retried on its own)
            n_proc = len(r["proc_idx"][b])
            tracks = []
            for ti in per_clip.get(b, ()):
                _, j = r["kept"][ti]
                regs = r["regions"][r["tr_off"][ti]:r["tr_off"][ti + 1]]
                use = r["usable"][upos[ti]:upos[ti + 1]] - r["tr_off"][ti]
                st = r["stats"][upos[ti]:upos[ti + 1]]
                tr = dict(summary=r["summ"][b, j], regions=regs, thumb=tracker.thumbnail_of(regs, use, st))
                if job["classify"]:
                    secs = sum(mo["seconds"] for mo in r["model_out"]) / n_kept
                    tr["predictions"] = track_predictions(r, ti, b, r["model_out"], secs)
                tracks.append(tr)
            trackless = None
            if not tracks:
                kind, payload = r["best_region"].get(b, ("none", None))
                trackless = _trackless_region(kind, payload)
            texts[i] = tracker.metadata_text(clips[b], n_proc, tracks, trackless, paths[b], job["tracking_time"],
                                             existing[b], indent, models=job["model_meta"])
            frames += int(offs[b + 1] - offs[b])
        except Exception as e:  # noqa: BLE001
            texts.pop(i, None)
            retry[i] = "%s: %s: %s" % (paths[b], type(e).__name__, e)
    return texts, retry, frames, time.time() - th


# ---- metadata worker processes ---------------------------------------------------------------------------------------
# The host stage is Python (dictionaries, json, a few NumPy calls per track): one interpreter lock.  With a classifier
# over real recordings it is as long as the device stage (0.23 ms per recording), and every moment it holds the lock the
# device thread cannot launch (16,384 fixture recordings: 7.1 s with the stage in this process, 5.9 s with the stage
# left out).  The reference spreads its files over a multiprocessing.Pool (trackextractor.py:80-85); here the DEVICE work
# stays in one process per GPU and only the text formatting goes to worker processes -- which never touch the GPU.
# They are started as NEW programs ("spawn": a child process that execs a fresh interpreter, as subprocess.run does --
# never a fork that keeps running with this process's GPU state, never an exec of this process itself), preferably before
# this process initialises the GPU: by TrackExtractor.extract / ClipClassifier.process / bench.py at their start -- or not
# at all, and the stage runs in-process as before.
_WORKER_TRACKERS = {}


def _worker_format(config_blob, job):
    import pickle

    tracker = _WORKER_TRACKERS.get(config_blob)
    if tracker is None:
        from .. import _lib

        tracker = BulkTracker(pickle.loads(config_blob), 0)
        tracker.lib = _lib.load()   # (the host entry points cpx_format_regions / cpx_json_indent: no GPU call)
        _WORKER_TRACKERS.clear()
        _WORKER_TRACKERS[config_blob] = tracker
    return format_group(tracker, job)


def _worker_ready(_):
    import numpy  # noqa: F401  (the imports a job needs, paid before the first one arrives)

    from .. import _lib

    _lib.load()
    return os.getpid()


_SPAWN_PATCH_LOCK = threading.Lock()


class MetaPool:
    """`workers` processes that format metadata text (format_group); make() returns None when they cannot be started
    (the stage then stays in-process)."""

    def __init__(self, workers):
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor

        import multiprocessing.spawn as mp_spawn

        self.workers = int(workers)
        self.broken = False     # set when a submit was refused (a worker died mid-run: BrokenProcessPool)
        self.pool = ProcessPoolExecutor(max_workers=self.workers, mp_context=multiprocessing.get_context("spawn"))
        # A spawned child normally re-imports the parent's __main__ -- i.e. runs the CALLER's script again, top to bottom,
        # unless that script hides its body behind `if __name__ == "__main__"` (a directory run started from such a script
        # ran once per worker more, concurrently).  The workers need nothing from __main__ (their functions live in this
        # module), so the start-up data goes without it.  The executor starts all its workers at the first submit.
        with _SPAWN_PATCH_LOCK:
            orig = mp_spawn.get_preparation_data

            def without_main(name):
                d = orig(name)
                d.pop("init_main_from_path", None)
                d.pop("init_main_from_name", None)
                return d

            mp_spawn.get_preparation_data = without_main
            try:
                # (not waited for: the processes are created inside submit; their imports -- about a second -- run
                # beside the first decode batch, and jobs queue behind these)
                self._ready = [self.pool.submit(_worker_ready, k) for k in range(4 * self.workers)]
            finally:
                mp_spawn.get_preparation_data = orig

    @property
    def pids(self):
        """process ids of the workers that answered (waits for their start-up)"""
        return sorted({f.result() for f in self._ready})

    def usable(self):
        """False once a worker's start-up is known to have failed (import error, killed process): the caller then
        formats in-process instead of sending every group to a pool that answers with exceptions."""
        return not self.broken and not any(f.done() and (f.cancelled() or f.exception() is not None) for f in self._ready)

    @staticmethod
    def make(workers=None):
        if workers is None:
            from ..sharding import host_threads_per_rank

            # two: on the 16-CPU GPU boxes of this pool four workers took enough CPU time from the staging / decode
            # threads to slow the run (bench from_files, 8,192 noisy recordings: 462 k frames/s without workers, 453 k
            # with two, 267-325 k with four); a rank of a shared node gets fewer
            workers = max(1, min(2, host_threads_per_rank(16) // 4))
        try:
            import torch

            if torch.cuda.is_initialized():
                # the workers are spawned children (fresh processes, nothing of HIP inherited), so this is allowed; callers
                # nevertheless start the pool BEFORE their first GPU work where they can (TrackExtractor.extract,
                # bench.py), and this line makes the other case visible
                logging.info("metadata worker pool started from a process that has initialised the GPU")
            return MetaPool(workers)
        except Exception as e:  # noqa: BLE001 -- no pool is a slower run, not a failed one
            logging.warning("metadata worker pool not started (%s: %s): formatting in-process", type(e).__name__, e)
            return None

    def submit(self, config_blob, job):
        """-> the job's future, or None when the executor refuses work (a worker killed mid-run breaks the whole
        ProcessPoolExecutor; every later submit raises at once): the pool is then marked unusable and the caller formats
        in-process."""
        try:
            return self.pool.submit(_worker_format, config_blob, job)
        except Exception as e:  # noqa: BLE001 -- BrokenProcessPool, RuntimeError after shutdown
            logging.warning("metadata worker pool refused a job (%s: %s): formatting in-process from here on",
                            type(e).__name__, e)
            self.broken = True
            return None

    def close(self):
        self.pool.shutdown(wait=True, cancel_futures=True)


def _trackless_region(kind, payload):
    """What best_trackless_thumb returns, as the metadata dictionary of that Region."""
    from .region import Region

    if kind == "region":
        return Region.from_record(payload).meta_dictionary()
    if kind == "window":
        q, x, y = payload
        return Region(x, y, THUMBNAIL_SIZE, THUMBNAIL_SIZE, frame_number=q,
                      centroid=(x + THUMBNAIL_SIZE // 2, y + THUMBNAIL_SIZE // 2)).meta_dictionary()
    return None


def track_predictions(r, ti, b, model_out, classify_seconds):
    """The "predictions" entry of kept track `ti` (ClipClassifier.save_metadata / TrackPrediction.get_metadata,
    reference src/classify/clipclassifier.py:305-383, src/classify/trackprediction.py:465-501) from the batch's
    network outputs: per model the sum / normalisation / low-evidence cap of Interpreter.track_prediction_from_raw
    (src/ml_tools/interpreter.py:151-168) over the track's segments."""
    from ..classify.trackprediction import TrackPrediction

    out = []
    smp = r["samples"]
    if smp is None:
        return out
    tp = r["kept_pipe"][ti]
    s0, s1 = np.searchsorted(smp["sample_track"], [tp, tp + 1])
    if s1 <= s0:
        return out  # no segment ("Skipping track"): no prediction entry
    regs = r["regions"][r["tr_off"][ti]:r["tr_off"][ti + 1]]
    first = int(regs["frame_number"][0])
    mass16 = regs["mass"].astype(np.uint16)
    frames = np.searchsorted(r["proc_idx"][b], smp["frames"][s0:s1])   # device frame index -> frame number
    masses = []
    for fr in frames:
        # SegmentHeader.mass: the uint16 sum over the segment after its first padding (distinct frames + the first
        # min(missing, have) of them once more), ml_tools/datasetstructures.py:1236-1249
        d = np.unique(fr)
        extra = min(len(fr) - len(d), len(d))
        masses.append(np.uint16(np.sum(mass16[d - first]) + np.sum(mass16[d[:extra] - first])))
    for mo in model_out:
        if mo["probs"] is None:
            continue
        interp = mo["interp"]
        pred = TrackPrediction(int(r["summ"][b, r["kept"][ti][1]]["id"]), interp.labels,
                               smooth_preds=interp.params.smooth_predictions)
        pred.classified_track(mo["probs"][s0:s1], [f for f in frames], masses)
        if len(frames) == 1 and len(set(frames[0].tolist())) < interp.params.square_width ** 2 / 4:
            if pred.predicted_tag() != "false-positive":
                pred.cap_confidences(0.5)
        pred.classify_time = classify_seconds
        pm = pred.get_metadata(interp.thresholds)
        pm["model_id"] = mo["model"].id
        out.append(pm)
    return out


def auto_batch_files(n_files):
    """Recordings per decode launch when the caller does not say: the inflate kernel runs one wavefront per recording
    and a launch of 1,024 leaves the chip at 4 waves per CU (12.7 GB/s of output against 20 at 2,048, 26.5 at 4,096
    and 29.5 at 8,192: profiles/r04_inflate_sq_counters.json), while the pipeline overlaps decode, tracking and
    metadata only across batches -- so: a quarter of the run, between 1,024 and 2,048.  Measured with the classifier
    (scratch/from_files_profile.py): 16,384 fixture copies 6.90 / 6.62 / 7.38 / 7.17 s at 1,024 / 2,048 / 3,072 / 4,096,
    8,192 synthetic recordings 6.08 / 4.97 / 5.34 s at 1,024 / 2,048 / 4,096; tracking only (8,192 fixture copies
    through TrackExtractor.extract): 456 k frames/s at 1,024, 542 k at 2,048, 469 k at 4,096."""
    return int(min(2048, max(1024, (n_files // 4 + 255) // 256 * 256)))


def run_files_bulk(filenames, config, to_stdout=False, save_meta=True, device=0, batch_files=None, want_text=False,
                   stager=None, tracker=None, clip_classifier=None, blobs=None, track_files=1024,
                   decode_bytes=8 << 30, track_frames=400000, meta_pool=None, device_lanes=2):
    """extract_file -- or, with a ClipClassifier, process_file(track=True) -- for many recordings at device speed.
    Writes <file>.txt (or prints with to_stdout) and returns ({filename: metadata text (want_text) or True, or an
    "error: ..." string for a skipped file}, tracker with timings).  Files that cannot take the batched path are
    retried through the one-file path.  blobs: the recordings as byte strings already in memory (names in
    `filenames`; nothing is read from disk; one the batch refuses is retried from its bytes by the host reader).
    batch_files: recordings per decode launch (and per read-ahead batch; None: auto_batch_files); track_files: recordings
    per tracking group."""
    import torch

    from .cliptrackextractor import default_engine
    from .trackextractor import extract_file

    filenames = [str(f) for f in filenames]
    if not batch_files:
        batch_files = auto_batch_files(len(filenames))
    own_stager = stager is None and blobs is None
    if blobs is None:
        stager = stager or FileStager(torch)
    tracker = tracker or BulkTracker(config, device)
    eng0 = default_engine(device)
    indent = None if (to_stdout or (clip_classifier is not None and config.classify.meta_to_stdout)) else 4
    classifiers, models = [], []
    if clip_classifier is not None:
        models = [clip_classifier.model] if clip_classifier.model else (config.classify.models or [])
        t0 = time.time()
        location = clip_classifier.first_location(filenames) if blobs is None else None
        classifiers = [(m, clip_classifier.get_classifier(m, location)) for m in models]
        tracker.timings["model_load_s"] = time.time() - t0
    out = {}
    # decode launches: at most batch_files recordings and decode_bytes of compressed data (inflated bytes + frames are
    # 4-6 x that on the device, and three batches are alive at once); tracking groups: at most track_files recordings
    # and track_frames frames -- ten-minute recordings do not take the device's memory with them
    n_all = len(filenames)
    if blobs is not None:
        sizes = [len(b) for b in blobs]
    else:
        sizes = []
        for f in filenames:
            try:
                sizes.append(os.path.getsize(f))
            except OSError:
                sizes.append(0)
    runs = plan_decode_batches(sizes, batch_files, decode_bytes)
    order = [a for a, _ in runs]
    bounds = [b for _, b in runs]
    batches = [filenames[a:b] for a, b in runs]
    # Two stages in flight: a worker thread stages batch k+1 (file reads / copies into pinned memory) and decodes it on
    # a handle of its own (its own HIP stream: upload, inflate + section index, unpack) while this thread tracks,
    # classifies and writes batch k.  The inflate kernel is bound by scalar issue and latency, the network by the matrix
    # pipe: the two share the chip well.
    from ..engine import TrackEngine

    deng = tracker.decode_engine
    if deng is None:
        deng = tracker.decode_engine = TrackEngine(width=eng0.width, height=eng0.height, device=device, max_frames=45)

    # (the staging of batch k+1 -- file reads / copies into pinned memory, host work -- runs beside the decode of batch k)
    stage_worker = ThreadPoolExecutor(max_workers=1)
    stage_futs = {}
    upload_stream = torch.cuda.Stream(device=deng.device)

    def stage(bi):
        t0 = time.time()
        if blobs is None:
            staged = stager.stage(batches[bi]).result()
        else:
            staged = stage_blobs(torch, blobs[order[bi]:bounds[bi]])
            staged.paths = batches[bi]
        # ... and so does its upload (a copy stream of its own: the DMA engine, beside the previous batch's inflate)
        with torch.cuda.stream(upload_stream):
            staged.in_dev = staged.stage[: int(staged.in_off[-1]) + 16].to(deng.device, non_blocking=True)
            staged.in_event = torch.cuda.Event()
            staged.in_event.record(upload_stream)
        return staged, time.time() - t0

    # (Measured and not adopted: enqueueing the inflate of batch k+1 -- inflate_launch -- before batch k's results are
    # read and unpacked on a second handle keeps the decode stream busy back to back, but the tracking and network
    # kernels of the other thread then never see the chip without an inflate on it: 8,192 noisy recordings 4.41 s
    # either way, the directory of fixture copies 2,600-2,700 files/s against 3,400.)
    def produce(bi):
        if bi not in stage_futs:
            stage_futs[bi] = stage_worker.submit(stage, bi)
        staged, stage_s = stage_futs.pop(bi).result()
        if bi + 1 < len(batches):
            stage_futs[bi + 1] = stage_worker.submit(stage, bi + 1)
        t1 = time.time()
        try:
            decoded = decode_staged(deng, staged)
        except Exception as e:  # noqa: BLE001 -- (device memory for the batch's buffers, a refused launch): every member
            # is retried on its own instead of the run unwinding with the open batches' metadata unwritten
            if isinstance(e, torch.cuda.OutOfMemoryError):
                torch.cuda.empty_cache()
            decoded = DecodedBatch()
            decoded.errors.update(staged.errors)
            why = "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")
            for i in range(len(staged.paths)):
                decoded.errors.setdefault(i, "%s: decode stage failed (%s)" % (staged.paths[i], why))
        return staged, decoded, stage_s, time.time() - t1

    worker = ThreadPoolExecutor(max_workers=1)
    # Device lanes: a device phase is a chain of launches with host steps between them (work counts that size the next
    # buffers, the kept tracks' region gather, thumbnail requests): while the host does those the device has nothing
    # of that group to run -- 24 % of the wall time with one group at a time (rocprofv3 kernel trace of 8,192 fixture
    # recordings, scratch/gpu_busy_from_trace.py).  Two groups in flight, each on an engine (handle = stream + tracking
    # workspace) of its own, fill each other's gaps.
    # How many lanes the run uses is decided once, from what its second batch had to wait for: a run bound by the DECODE stage
    # (large noisy recordings: this thread sits waiting for the inflate) keeps one lane -- a second one only takes device
    # time from the inflate kernel (8,192 synthetic recordings: 462 k frames/s with one lane, 295 k with two) -- a run
    # bound by the device phase (real recordings: many tracks and segments per frame) takes two (16,384 fixture
    # recordings: 322 k -> 409 k).  Two lanes hold the device memory of one: half-sized groups and network calls.
    n_lanes = max(1, int(os.environ.get("CPX_BULK_LANES", device_lanes)))
    lanes_now = [1]
    base_track_files, base_track_frames = track_files, track_frames
    dev_worker = ThreadPoolExecutor(max_workers=n_lanes)
    lane_free = list(range(n_lanes))
    lane_cond = threading.Condition()   # device phases running at once: at most lanes_now (groups are submitted one ahead)
    lanes_busy = [0]
    fut = worker.submit(produce, 0) if batches else None

    # Three stages in flight: the decode of batch k+1 (worker above), the device phase of group g+1 (dev_worker: track,
    # classify, thumbnails -- mostly waiting on the device with the GIL released) and the metadata of group g (this
    # thread).
    def device_phase(paths, group):
        td = time.time()
        clips, existing, pre_failed = [], [], {}
        for k, i in enumerate(group.files):
            # per-file work stays inside a try: a header without a timestamp, an absurd one, or a corrupt existing
            # .txt fails THAT recording (it goes the one-file way, which reports it) and nothing else
            clip = Clip(tracker.tcfg, paths[i])
            clip.frames_per_second = 9
            meta = None
            try:
                h = group.headers[k]
                clip.set_res(h.x_resolution, h.y_resolution)
                clip.set_model(h.model if h.model else None)
                clip.set_video_stats(datetime.fromtimestamp(h.timestamp / 1000000).astimezone(Clip.local_tz))
                mf = os.path.splitext(paths[i])[0] + ".txt"
                if blobs is None and os.path.exists(mf):
                    meta = tools.load_clip_metadata(mf)
            except Exception as e:  # noqa: BLE001 -- fault isolation
                pre_failed[k] = "%s: %s" % (type(e).__name__, e)
                clip.set_res(group.key[0], group.key[1])
                clip.set_model(group.key[2])
            clips.append(clip)
            existing.append(meta)
        # (lepton3 and "no model" files share thresholds but not the metadata's camera_model: grouped by model)
        with lane_cond:
            while lanes_busy[0] >= lanes_now[0]:
                lane_cond.wait()
            lanes_busy[0] += 1
            lane = lane_free.pop()
        try:
            r = tracker.track_group(group, clips, classifiers, lane=lane)
        except torch.cuda.OutOfMemoryError as e:   # (budgets too generous for this device: memory released, members retried)
            r = RuntimeError("device out of memory for a group of %d recordings / %d frames: %s"
                             % (len(group.files), int(group.offs[-1]), str(e).splitlines()[0]))
            if os.environ.get("CPX_BULK_OOM_DEBUG"):
                import sys as _sys
                free_b, total_b = torch.cuda.mem_get_info()
                _sys.stderr.write("OOM DEBUG: free %.1f GiB of %.1f; torch allocated %.1f reserved %.1f GiB; lanes %s\n%s\n" % (
                    free_b / 2**30, total_b / 2**30, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
                    lanes_now[0], torch.cuda.memory_summary(abbreviated=True)))
            torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001 -- the whole group failed (CpxError or anything else): every member goes the slow way
            r = e
        finally:
            with lane_cond:
                lanes_busy[0] -= 1
                lane_free.append(lane)
                lane_cond.notify_all()
        if not isinstance(r, Exception):
            for k, why in pre_failed.items():
                r["failed"].setdefault(k, why)
        return clips, existing, r, time.time() - td

    def host_phase(ctx, group, result):
        paths, texts, retry = ctx["paths"], ctx["texts"], ctx["retry"]
        clips, existing, r, dev_s = result
        tracker.timings["device_s"] += dev_s
        if isinstance(r, Exception):
            for i in group.files:
                retry[i] = "%s: %s" % (paths[i], r)
            return
        th = time.time()
        offs = group.offs
        if os.environ.get("CPX_BULK_SKIP_META"):  # experiment switch: how long the run takes without the metadata stage
            for b, i in enumerate(group.files):
                texts[i] = "{}"
                tracker.timings["frames"] += int(offs[b + 1] - offs[b])
            return
        model_meta = None
        if classifiers:
            model_meta = []
            for mo in r["model_out"]:
                d = mo["model"].as_dict()
                d["classify_time"] = float(round(mo["seconds"] / max(len(group.files), 1), 1))
                model_meta.append(d)
        keys = ("summ", "kept", "kept_pipe", "tr_off", "regions", "usable", "stats", "failed", "best_region", "proc_idx",
                "samples")
        job = dict(r=dict({k: r[k] for k in keys}, model_out=model_views(r["model_out"])), files=list(group.files),
                   paths=[paths[i] for i in group.files], offs=np.asarray(offs), clips=[clip_view(c) for c in clips],
                   existing=existing, indent=indent, classify=bool(classifiers), model_meta=model_meta,
                   tracking_time=(time.time() - ctx["t0"]) / max(ctx["n_ok"], 1))
        if meta_pool is not None and meta_pool.usable():   # formatted by a worker process; collected when the batch closes
            fut = meta_pool.submit(config_blob, job)
            if fut is not None:
                ctx["futures"].append(fut)
                tracker.timings["host_submit_s"] = tracker.timings.get("host_submit_s", 0.0) + time.time() - th
                return
        got_texts, got_retry, frames, _ = format_group(tracker, job)
        texts.update(got_texts)
        retry.update(got_retry)
        tracker.timings["frames"] += frames
        tracker.timings["host_s"] += time.time() - th

    def close_batch(ctx):
        paths, texts, retry = ctx["paths"], ctx["texts"], ctx["retry"]
        tw = time.time()
        for f in ctx["futures"]:   # the groups' texts from the metadata workers
            try:
                got_texts, got_retry, frames, secs = f.result()
            except Exception as e:  # noqa: BLE001 -- a worker that died: its recordings go the one-file way
                logging.warning("metadata worker failed (%s: %s)", type(e).__name__, e)
                continue
            texts.update(got_texts)
            retry.update(got_retry)
            tracker.timings["frames"] += frames
            tracker.timings["host_s"] += secs
        tracker.timings["host_collect_s"] = tracker.timings.get("host_collect_s", 0.0) + time.time() - tw
        if ctx["futures"]:   # (a recording no worker answered for is retried on its own, like any other failure)
            for i in range(len(paths)):
                if i not in texts and i not in retry and i in ctx["expected"]:
                    retry[i] = "%s: no metadata came back" % paths[i]
        tw = time.time()
        for i, text in texts.items():
            if indent is None and (to_stdout or clip_classifier is not None):
                print(text)
            elif save_meta:
                with open(os.path.splitext(paths[i])[0] + ".txt", "w") as fh:
                    fh.write(text)
            out[paths[i]] = text if want_text else True
            tracker.timings["files"] += 1
        tracker.timings["write_s"] += time.time() - tw
        for i, why in sorted(retry.items()):
            # a recording the batch refused goes through the one-file path: the host reader (zlib + section walk) is
            # the reference's one reader (cliptrackextractor.py:108-129) and never loses a valid file; an in-memory
            # recording is handed to it as bytes
            blob = blobs[ctx["base"] + i] if blobs is not None else None
            logging.warning("batched extract: %s -- retrying the recording on its own", why)
            try:
                if clip_classifier is not None:
                    res = clip_classifier.process_file(paths[i], track=True, calculate_thumbnails=True, device=device,
                                                       blob=blob)
                    if not res:
                        raise RuntimeError("process_file refused the file")
                else:
                    res = extract_file(paths[i], config, False, False, to_stdout, save_meta=save_meta, blob=blob)[2]
                out[paths[i]] = (json.dumps(res, indent=indent, cls=tools.CustomJSONEncoder) if want_text else True)
            except Exception as e:  # noqa: BLE001 -- fault isolation: one bad recording must not stop the directory
                logging.error("could not process %s: %s", paths[i], e)
                out[paths[i]] = "error: %s" % (e,)

    config_blob = None
    if meta_pool is not None:
        import pickle

        config_blob = pickle.dumps(config)
    pending = []   # (ctx, group, future of its device phase), oldest first: at most lanes_now of them
    last_batch_t0 = [time.time()]

    def drain():
        if pending:
            ctx, group, f = pending.pop(0)
            host_phase(ctx, group, f.result())
            ctx["open"] -= 1
            if ctx["open"] == 0 and ctx["submitted"]:
                closing.append(ctx)
        close_ready()

    closing = []   # batches whose groups are all through the device, in order; closed when their texts are there

    def close_ready(block=False):
        # (with metadata workers a batch's texts arrive later: this thread must not wait for them -- it is the one that
        # hands the next groups to the device thread)
        while closing and (block or all(f.done() for f in closing[0]["futures"])):
            close_batch(closing.pop(0))

    # Three Python threads share one interpreter lock: the metadata thread formats text (holds it), the device thread
    # comes back from every device wait / ctypes call needing it -- and CPython makes a waiter wait a full switch interval
    # (5 ms by default) before the holder is asked to yield.  A device phase is dozens of such returns per group: with the
    # default interval the device thread spent more time queueing for the lock than on the device (16,384 fixture
    # recordings: 10.2 s with the metadata stage, 5.9 s without it).  0.2 ms for the duration of the run.
    import sys as _sys

    switch_interval = _sys.getswitchinterval()
    _sys.setswitchinterval(float(os.environ.get("CPX_BULK_SWITCH_INTERVAL", "0.0002")))
    try:
        for bi, paths in enumerate(batches):
            t0 = time.time()
            staged, decoded, stage_s, decode_s = fut.result()
            fut = worker.submit(produce, bi + 1) if bi + 1 < len(batches) else None
            tracker.timings["stage_s"] = tracker.timings.get("stage_s", 0.0) + stage_s
            tracker.timings["decode_s"] += decode_s
            waited = time.time() - t0
            tracker.timings["wait_decode_s"] = tracker.timings.get("wait_decode_s", 0.0) + waited
            if n_lanes > 1 and bi == 1:
                # decided once, at the second batch (a lane's engine, network buffers and allocator pool stay behind when
                # it is dropped again, and a lane added later allocates on top of buffers sized for one: measured, out of
                # memory beside the bench's resident clips).  Decode-bound runs keep one lane -- two lanes slow their decode
                # (8,192 noise recordings: 6.1 s against 4.7 s).  The wait for this batch's decode alone does not tell:
                # the first batch's device phase carries the run's allocations and can hide a decode that steady batches
                # will wait for; the decode's own duration against that turn does
                lap_s = max(t0 - last_batch_t0[0], 1e-6)   # the first batch's turn of this loop
                decode_bound = waited > 0.2 * (lap_s + waited) or decode_s > 0.5 * lap_s
                lanes_now[0] = 1 if decode_bound else n_lanes
                tracker.timings["device_lanes"] = lanes_now[0]
            last_batch_t0[0] = time.time()
            k = lanes_now[0]
            track_files, track_frames = max(64, base_track_files // k), max(20000, base_track_frames // k)
            tracker.cnn_chunk = max(256, 2048 // k)
            groups = [sub for g in decoded.groups for sub in g.split(track_files, track_frames)]
            ctx = dict(paths=paths, base=order[bi], texts={}, retry=dict(decoded.errors), t0=t0, open=len(groups), submitted=False,
                       n_ok=sum(len(g.files) for g in decoded.groups), futures=[],
                       expected=set(i for g in groups for i in g.files))
            for gi, group in enumerate(groups):
                f = dev_worker.submit(device_phase, paths, group)
                ctx["submitted"] = gi + 1 == len(groups)
                pending.append((ctx, group, f))
                while len(pending) > lanes_now[0]:   # the oldest group's metadata, while the newer ones are on the device
                    drain()
            if not groups:
                closing.append(ctx)
                close_ready()
            del staged, decoded
        while pending:
            drain()
        close_ready(block=True)
    finally:  # (an exception that does escape must not leave three executors running)
        _sys.setswitchinterval(switch_interval)
        dev_worker.shutdown(wait=True)
        worker.shutdown(wait=True)
        stage_worker.shutdown(wait=True)
    if own_stager:
        stager.close()
    return out, tracker


def extract_files_bulk(filenames, config, to_stdout=False, save_meta=True, device=0, batch_files=None,
                       want_text=False, stager=None, tracker=None, meta_pool=None):
    """run_files_bulk without classification: what TrackExtractor.extract(directory) runs."""
    return run_files_bulk(filenames, config, to_stdout, save_meta, device, batch_files, want_text, stager, tracker,
                          meta_pool=meta_pool)
