"""Batched extract-and-classify pipeline: every stage of the hot path for B clips with no host
round trip of image data -- track stage, association, end-of-clip filtering, segment planning,
crop / tile, CNN forward, per-track aggregation.  This is what bench.py times; the per-clip
drop-in classes (cpx.track / cpx.classify) run the same kernels one clip at a time."""

import ctypes as C
import os

import numpy as np

from ._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE, TRACK_DEFER_MEDIANS, CpxError
from .tracking import TRACK_SUMMARY_DTYPE, make_filter_params, make_track_params


class BatchResult:
    def __init__(self):
        self.track = None          # TrackBatchResult
        self.assoc = None          # AssocBatchResult
        self.summaries_dev = None  # cpx_track_summary [B * max_tracks]
        self.counts = None         # host int32 [B,4]
        self.n_tracks = 0          # kept tracks in the batch
        self.n_samples = 0
        self.track_clip = None     # device int32 [n_tracks,2] (clip, track id)
        self.scores = None         # device float [n_tracks, L]
        self.best = None           # device int32 [n_tracks]
        self.samples_dev = None    # device float [n_samples, S, S, 2] (only when keep_samples)
        self.reqs_dev = None
        self.sample_track_dev = None
        self.limits_dev = None     # cpx_track_limits [n_tracks]
        self.probs = None          # device float [n_samples, L]
        self.parts = None          # overlapped run: the per-group results
        self.overflowed = []       # clips whose tables overflowed in this pass (run again on larger ones by run())
        self.regrown = []          # (clip, components capacity, max_active_tracks, max_tracks) of the clips run again
        self.clip0 = 0
        self.track_timing = (0.0, 0)

    def summaries(self, max_tracks):
        return self.summaries_dev.cpu().numpy().view(TRACK_SUMMARY_DTYPE).reshape(-1, max_tracks)


class BatchPipeline:
    def __init__(self, engine, network=None, n_labels=17, fp_index=-1, frame_size=32, square_width=5,
                 track_params=None, filter_params=None, cnn_chunk=512, want_regions=False, limits_flags=0):
        self.want_regions = want_regions  # per-frame region lists from the association (the trackless thumbnail)
        self.limits_flags = limits_flags  # _lib.LIMITS_*: the model's normalisation variant (cpx_track_limits_batch_ex)
        self.eng = engine
        self.net = network
        self.n_labels = n_labels
        self.fp_index = fp_index
        self.fs, self.sq = frame_size, square_width
        self.tp = track_params or make_track_params(engine.width, engine.height, engine.cfg.edge_pixels)
        self.fp = filter_params or make_filter_params(max_active_tracks=self.tp.max_active_tracks,
                                                      max_tracks_per_clip=self.tp.max_tracks)
        self.cnn_chunk = cnn_chunk
        self._sample_buf = None

    def _check(self, rc):
        if rc != 0:
            raise CpxError(rc, self.eng._err())

    def run(self, frames_dev, clip_offsets, meta, outputs=None, classify=True, keep_samples=False, sub_batches=1):
        """All stages for the clips of one batch.  Everything -- the cpx kernels and the few torch ops between them --
        is enqueued on the handle's HIP stream (torch's current stream is switched to it for the duration), so stream
        order replaces host synchronisation: the host blocks exactly once, to read the per-clip work counts that size
        the classification buffers.  sub_batches > 1 (and a network living on a second engine, i.e. a second HIP
        stream): the batch is cut into that many groups of clips and the HBM-bound track stage of group k+1 runs
        while the MFMA-bound network works on group k."""
        if sub_batches > 1 and classify and self.net is not None and not keep_samples:
            return self._run_overlapped(frames_dev, clip_offsets, meta, outputs, sub_batches)
        eng, t = self.eng, self.eng.torch
        t.cuda.current_stream(eng.device).synchronize()  # the caller's inputs (made on its stream) are complete
        with t.cuda.stream(eng.torch_stream()):
            return self._run_on_stream(frames_dev, clip_offsets, meta, outputs, classify, keep_samples)

    def _run_on_stream(self, frames_dev, clip_offsets, meta, outputs, classify, keep_samples):
        eng, t = self.eng, self.eng.torch
        out = self._front(frames_dev, clip_offsets, meta, outputs, classify)
        if not classify or out.n_tracks == 0 or out.n_samples == 0:
            eng.synchronize()
        else:
            out = self.classify_front(out, frames_dev, keep_samples)
        if out.overflowed:
            out = self._regrow(out, frames_dev, clip_offsets, meta, classify, keep_samples)
        return out

    def _regrow(self, out, frames_dev, clip_offsets, meta, classify, keep_samples):
        """No recording is lost to a capacity: the clips whose frames held more components, or which had more tracks,
        than this pass's tables (reference: no limits, cliptrackextractor.py:236-247, cliptracker.py:202-247) reported
        no tracks; each is run again alone on tables grown to fit and its tracks join the batch's."""
        from .tracking import FilterParams

        eng, t = self.eng, self.eng.torch
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        meta = np.ascontiguousarray(meta)
        parts = [out]
        for b in out.overflowed:
            f0, f1 = int(offs[b]), int(offs[b + 1])
            fr = frames_dev[f0:f1]
            eng_k, res_k, assoc_k, tp_k = eng.track_clip_grown(fr, meta[f0:f1], params=self.tp, want_filtered=True,
                                                               want_regions=self.want_regions)
            fp_k = FilterParams.from_buffer_copy(self.fp)
            fp_k.max_active_tracks, fp_k.max_tracks_per_clip = tp_k.max_active_tracks, tp_k.max_tracks
            sub = BatchPipeline(eng_k, self.net, n_labels=self.n_labels, fp_index=self.fp_index, frame_size=self.fs,
                                square_width=self.sq, track_params=tp_k, filter_params=fp_k, cnn_chunk=self.cnn_chunk,
                                want_regions=self.want_regions, limits_flags=self.limits_flags)
            with t.cuda.stream(eng_k.torch_stream()):
                part = sub._front(fr, np.array([0, f1 - f0], np.int32), meta[f0:f1], None, classify, pre=(res_k, assoc_k))
                if classify and part.n_tracks and part.n_samples:
                    part = sub.classify_front(part, fr, keep_samples)
                else:
                    eng_k.synchronize()
            part.clip0 = b
            out.regrown.append((b, eng_k.cap, tp_k.max_active_tracks, tp_k.max_tracks))
            parts.append(part)
        live = [p for p in parts if p.n_tracks and p.track_clip is not None]
        out.parts = parts
        out.n_tracks = sum(p.n_tracks for p in parts)
        out.n_samples = sum(p.n_samples for p in parts)
        if live:
            tcs = []
            for p in live:
                tc = p.track_clip.clone()
                tc[:, 0] += p.clip0
                tcs.append(tc)
            out.track_clip = t.cat(tcs)
            for name in ("scores", "best", "probs", "logits"):
                vals = [getattr(p, name, None) for p in live]
                if all(v is not None for v in vals):
                    setattr(out, name, t.cat(vals))
        return out

    def classify_front(self, out, frames_dev, keep_samples=False):
        """Stages 5b-7 for a _front result: crop / tile, network, aggregation (on the current stream = the handle's).
        Callable again on the same `out` from a pipeline with another network / frame size (one per model)."""
        eng, t = self.eng, self.eng.torch
        lib, h = eng.lib, eng.h
        dev = eng.device
        n_tracks, n_samples = out.n_tracks, out.n_samples
        reqs, limits, per = out.reqs_dev, out.limits_dev, self.sq * self.sq
        side = self.sq * self.fs
        probs = t.empty((n_samples, self.n_labels), dtype=t.float32, device=dev)
        logits = t.empty((n_samples, self.n_labels), dtype=t.float32, device=dev)
        # equal chunks of at most cnn_chunk samples: a short last chunk runs the network's persistent kernels on a sliver of
        # the chip (92 samples behind three chunks of 2048 cost 4.3 ms of a 388 ms step, profiles/r06_step_timeline.txt)
        n_chunks = -(-n_samples // max(1, self.cnn_chunk))
        chunk = -(-n_samples // n_chunks)
        if keep_samples:
            out.samples_dev = t.empty((n_samples, side, side, 2), dtype=t.float32, device=dev)
        elif self._sample_buf is None or self._sample_buf.shape[0] < chunk or self._sample_buf.shape[1] != side:
            self._sample_buf = t.empty((chunk, side, side, 2), dtype=t.float32, device=dev)
        sample_bytes = side * side * 2 * 4
        # a network living on a second engine (= a second HIP stream; the set-up of sub_batches > 1) is ordered
        # against this engine's stream by events: its forward waits for the crop that fills its input, the next crop
        # into the shared buffer waits for that forward, and the aggregation waits for the last forward
        two_streams = self.net is not None and self.net.eng is not eng
        s_eng = eng.torch_stream()
        s_net = self.net.eng.torch_stream() if two_streams else None
        for s0 in range(0, n_samples, chunk):
            s1 = min(s0 + chunk, n_samples)
            buf = out.samples_dev[s0:s1] if keep_samples else self._sample_buf[: s1 - s0]
            # the chunk's requests carry absolute sample indices: hand the kernel the address sample 0 WOULD have
            # (it only writes the samples of these requests), instead of rewriting the requests
            self._check(lib.cpx_crop_tile(
                h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(out.track.filtered_dev.data_ptr()),
                C.c_void_p(out.track.info_dev.data_ptr()), C.c_void_p(reqs.data_ptr() + s0 * per * 32), (s1 - s0) * per,
                C.c_void_p(limits.data_ptr()), self.fs, self.sq, C.c_void_p(buf.data_ptr() - s0 * sample_bytes)))
            if self.net is not None:
                if two_streams:
                    ev = t.cuda.Event()
                    ev.record(s_eng)
                    s_net.wait_event(ev)
                self.net.forward_async(buf, logits[s0:s1], probs[s0:s1])
                if os.environ.get("CPX_CNN_DEBUG_OVF") and self.net.eng.cnn_last_overflow():
                    per = buf.reshape(buf.shape[0], -1)
                    mx = per.abs().amax(dim=1)
                    print("overflow forward: samples %d..%d, input max %.1f min %.1f nan %d; per-sample max: top %s; "
                          "logits max %.2f" % (s0, s1, float(mx.max()), float(per.min()), int(t.isnan(per).sum()),
                                               [round(float(v), 1) for v in mx.topk(min(5, mx.numel())).values],
                                               float(logits[s0:s1].abs().max())), flush=True)
                if two_streams:
                    ev = t.cuda.Event()
                    ev.record(s_net)
                    s_eng.wait_event(ev)
        if self.net is None:
            eng.synchronize()
            return out
        # ---- 7. per-track aggregation ----
        out.scores = t.empty((n_tracks, self.n_labels), dtype=t.float32, device=dev)
        out.best = t.empty(n_tracks, dtype=t.int32, device=dev)
        self._check(lib.cpx_aggregate_predictions(
            h, C.c_void_p(probs.data_ptr()), C.c_void_p(out.sample_track_dev.data_ptr()), n_samples,
            C.c_void_p(reqs.data_ptr()), n_tracks, self.n_labels, self.fp_index, self.sq,
            C.c_void_p(out.scores.data_ptr()), C.c_void_p(out.best.data_ptr())))
        out.probs = probs
        out.logits = logits
        eng.synchronize()  # results are complete when run() returns (callers read them from any stream)
        if two_streams:
            self.net.eng.synchronize()
        return out

    def _front(self, frames_dev, clip_offsets, meta, outputs, classify=True, pre=None):
        """Stages 1-5a on the track engine: track, association, end-of-clip filtering, segment plan, limits.
        pre: (TrackBatchResult, AssocBatchResult) of these clips when stages 1-2 have run already (_regrow)."""
        eng, t = self.eng, self.eng.torch
        lib, h = eng.lib, eng.h
        dev = eng.device
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        meta = np.ascontiguousarray(meta)
        B = offs.size - 1
        out = BatchResult()
        # ---- 1. track stage (one launch per time step), 2. association ----
        if pre is not None:
            out.track, out.assoc = pre
        else:
            # (the per-frame medians run on the handle's second stream beside the association, finalisation and plan below,
            # none of which reads them; cpx_track_limits_batch_ex / cpx_crop_tile / engine.synchronize() wait for them)
            out.track = eng.track_batch(frames_dev, offs, meta, want_filtered=True, outputs=outputs,
                                        flags=TRACK_DEFER_MEDIANS)
            out.assoc = eng.associate_batch(out.track, offs, meta, params=self.tp, want_regions=self.want_regions)
        # ---- 3. end of clip: trim / stats / rejects / plan sizes ----
        mt = self.tp.max_tracks
        summ = t.zeros(B * mt * 30, dtype=t.int32, device=dev)
        counts = t.zeros((B, 4), dtype=t.int32, device=dev)
        offs_p = offs.ctypes.data_as(C.POINTER(C.c_int32))
        eng.sync_inputs()
        self._check(lib.cpx_finalize_tracks(
            h, C.byref(self.fp), offs_p, C.c_void_p(meta.ctypes.data), B, C.c_void_p(out.assoc.pool_dev.data_ptr()),
            C.c_void_p(out.assoc.tracks_dev.data_ptr()), C.c_void_p(out.assoc.ntracks_dev.data_ptr()),
            C.c_void_p(summ.data_ptr()), C.c_void_p(counts.data_ptr())))
        out.summaries_dev = summ
        on_stream = int(t.cuda.current_stream(dev).cuda_stream) == int(lib.cpx_stream(h) or 0)
        if not on_stream:
            eng.synchronize()  # torch runs on another stream (the overlapped form): the kernels must have finished
        # exclusive prefix sums of the counts on the handle's stream (cpx_counts_prefix: the step launches no torch kernel)
        prefix = t.empty((B + 1, 4), dtype=t.int32, device=dev)
        self._check(lib.cpx_counts_prefix(h, C.c_void_p(counts.data_ptr()), B, C.c_void_p(prefix.data_ptr())))
        if not on_stream:
            eng.synchronize()
        out.counts = counts.cpu().numpy()   # the one host wait of a run: the work counts size what follows
        # (with it: which clips outgrew the tables -- a frame with more components than max_components, more tracks than
        # the association's slots; such a clip reported no tracks above and run() tracks it again on larger tables)
        out.overflowed = [int(b) for b in np.nonzero(out.assoc.status_dev.cpu().numpy())[0]]
        totals = out.counts.sum(axis=0)
        out.track_timing = eng.last_kernel_timing()  # (ms, launches) of this group's frame-kernel launches
        n_tracks, n_refs, n_samples = int(totals[0]), int(totals[1]), int(totals[2])
        out.n_tracks, out.n_samples = n_tracks, n_samples
        if not classify or n_tracks == 0 or n_samples == 0:
            return out
        # ---- 4. segment plan ----
        per = self.sq * self.sq
        # (every entry is written by the plan pass, the closing track offset included: no fills)
        refs = t.empty(max(n_refs, 1) * 6, dtype=t.int32, device=dev)
        toffs = t.empty(n_tracks + 1, dtype=t.int32, device=dev)
        reqs = t.empty(n_samples * per * 8, dtype=t.int32, device=dev)
        sample_track = t.empty(n_samples, dtype=t.int32, device=dev)
        track_clip = t.empty((n_tracks, 2), dtype=t.int32, device=dev)
        eng.sync_inputs()
        self._check(lib.cpx_plan_segments(
            h, C.byref(self.fp), offs_p, C.c_void_p(meta.ctypes.data), B, C.c_void_p(out.assoc.pool_dev.data_ptr()),
            C.c_void_p(summ.data_ptr()), C.c_void_p(out.assoc.ntracks_dev.data_ptr()), C.c_void_p(prefix.data_ptr()),
            self.sq, C.c_void_p(refs.data_ptr()), C.c_void_p(toffs.data_ptr()), C.c_void_p(reqs.data_ptr()),
            C.c_void_p(sample_track.data_ptr()), C.c_void_p(track_clip.data_ptr())))
        if not on_stream:
            eng.synchronize()
        out.track_clip, out.reqs_dev, out.sample_track_dev = track_clip, reqs, sample_track
        # ---- 5a. per-track limits ----
        limits = t.zeros(n_tracks * 8, dtype=t.int32, device=dev)
        eng.sync_inputs()
        self._check(lib.cpx_track_limits_batch_ex(
            h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(out.track.filtered_dev.data_ptr()),
            C.c_void_p(out.track.info_dev.data_ptr()), C.c_void_p(refs.data_ptr()), C.c_void_p(toffs.data_ptr()),
            n_tracks, C.c_void_p(limits.data_ptr()), int(self.limits_flags)))
        out.limits_dev = limits
        out._keep = (refs, toffs, prefix)
        return out

    def _run_overlapped(self, frames_dev, clip_offsets, meta, outputs, n_sub):
        eng, ceng, t = self.eng, self.net.eng, self.eng.torch
        lib, dev = eng.lib, eng.device
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        meta = np.ascontiguousarray(meta)
        B = offs.size - 1
        n_sub = max(1, min(n_sub, B))
        bounds = [int(round(B * k / n_sub)) for k in range(n_sub + 1)]
        s_track, s_cnn = eng.torch_stream(), ceng.torch_stream()
        per, side, cap = self.sq * self.sq, self.sq * self.fs, eng.cap
        parts = []
        for k in range(n_sub):
            b0, b1 = bounds[k], bounds[k + 1]
            if b1 <= b0:
                continue
            f0, f1 = int(offs[b0]), int(offs[b1])
            sub_out = None
            if outputs is not None:
                comps, info, labels, filt, bgo = outputs
                sub_out = (comps[f0 * cap * 8:f1 * cap * 8], info[f0 * 20:f1 * 20],
                           None if labels is None else labels[f0:f1], None if filt is None else filt[f0:f1],
                           None if bgo is None else bgo[b0:b1])
            fr = frames_dev[f0:f1]
            part = self._front(fr, offs[b0:b1 + 1] - f0, meta[f0:f1], sub_out)
            part.clip0 = b0
            parts.append(part)
            if part.n_tracks == 0 or part.n_samples == 0:
                continue
            ns, nt = part.n_samples, part.n_tracks
            # ---- 5b. crop / tile every segment of the group (track stream) ----
            samples = t.empty((ns, side, side, 2), dtype=t.float32, device=dev)
            part.probs = t.empty((ns, self.n_labels), dtype=t.float32, device=dev)
            logits = t.empty((ns, self.n_labels), dtype=t.float32, device=dev)
            part.scores = t.empty((nt, self.n_labels), dtype=t.float32, device=dev)
            part.best = t.empty(nt, dtype=t.int32, device=dev)
            eng.sync_inputs()
            self._check(lib.cpx_crop_tile(
                eng.h, C.c_void_p(fr.data_ptr()), C.c_void_p(part.track.filtered_dev.data_ptr()),
                C.c_void_p(part.track.info_dev.data_ptr()), C.c_void_p(part.reqs_dev.data_ptr()), ns * per,
                C.c_void_p(part.limits_dev.data_ptr()), self.fs, self.sq, C.c_void_p(samples.data_ptr())))
            ev = t.cuda.Event()
            ev.record(s_track)
            s_cnn.wait_event(ev)
            # ---- 6. network + 7. aggregation (network stream; nothing below blocks the host) ----
            n_chunks = -(-ns // max(1, self.cnn_chunk))
            chunk = -(-ns // n_chunks)           # equal chunks (classify_front)
            for s0 in range(0, ns, chunk):
                s1 = min(s0 + chunk, ns)
                self.net.forward_async(samples[s0:s1], logits[s0:s1], part.probs[s0:s1])
            rc = lib.cpx_aggregate_predictions(
                ceng.h, C.c_void_p(part.probs.data_ptr()), C.c_void_p(part.sample_track_dev.data_ptr()), ns,
                C.c_void_p(part.reqs_dev.data_ptr()), nt, self.n_labels, self.fp_index, self.sq,
                C.c_void_p(part.scores.data_ptr()), C.c_void_p(part.best.data_ptr()))
            if rc != 0:
                raise CpxError(rc, ceng._err())
            part._keep2 = (samples, logits, ev)
        eng.synchronize()
        ceng.synchronize()
        # ---- merge the groups ----
        out = BatchResult()
        out.parts = parts
        out.track = parts[0].track if len(parts) == 1 else _MergedCheck([p.track for p in parts])
        out.assoc = parts[0].assoc if len(parts) == 1 else _MergedCheck([p.assoc for p in parts])
        out.counts = np.concatenate([p.counts for p in parts])
        out.track_timing = (sum(p.track_timing[0] for p in parts), sum(p.track_timing[1] for p in parts))
        out.n_tracks = sum(p.n_tracks for p in parts)
        out.n_samples = sum(p.n_samples for p in parts)
        live = [p for p in parts if p.n_tracks and p.n_samples]
        if live:
            tcs = []
            for p in live:
                tc = p.track_clip.clone()
                tc[:, 0] += p.clip0
                tcs.append(tc)
            out.track_clip = t.cat(tcs)
            out.scores = t.cat([p.scores for p in live])
            out.best = t.cat([p.best for p in live])
            out.probs = t.cat([p.probs for p in live])
        # a clip that outgrew its group's tables reported no tracks above (cpx_assoc_kernel: n_tracks = 0): as in the
        # one-stream form it is tracked again alone on grown tables and its tracks join the batch's
        out.overflowed = sorted(p.clip0 + b for p in parts for b in p.overflowed)
        if out.overflowed:
            with t.cuda.stream(eng.torch_stream()):
                out = self._regrow(out, frames_dev, clip_offsets, meta, True, False)
        return out


class _MergedCheck:
    """check() over the per-group results of an overlapped run."""

    def __init__(self, items):
        self.items = items

    def check(self):
        for it in self.items:
            it.check()
