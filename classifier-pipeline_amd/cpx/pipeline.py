"""Batched extract-and-classify pipeline: every stage of the hot path for B clips with no host
round trip of image data -- track stage, association, end-of-clip filtering, segment planning,
crop / tile, CNN forward, per-track aggregation.  This is what bench.py times; the per-clip
drop-in classes (cpx.track / cpx.classify) run the same kernels one clip at a time."""

import ctypes as C

import numpy as np

from ._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE, CpxError
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

    def summaries(self, max_tracks):
        return self.summaries_dev.cpu().numpy().view(TRACK_SUMMARY_DTYPE).reshape(-1, max_tracks)


class BatchPipeline:
    def __init__(self, engine, network=None, n_labels=17, fp_index=-1, frame_size=32, square_width=5,
                 track_params=None, filter_params=None, cnn_chunk=512):
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

    def run(self, frames_dev, clip_offsets, meta, outputs=None, classify=True, keep_samples=False):
        eng, t = self.eng, self.eng.torch
        lib, h = eng.lib, eng.h
        dev = eng.device
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        B = offs.size - 1
        out = BatchResult()
        # ---- 1. track stage (one launch per time step), 2. association ----
        out.track = eng.track_batch(frames_dev, offs, meta, want_filtered=True, outputs=outputs)
        out.assoc = eng.associate_batch(out.track, offs, meta, params=self.tp, want_regions=False)
        # ---- 3. end of clip: trim / stats / rejects / plan sizes ----
        mt = self.tp.max_tracks
        summ = t.zeros(B * mt * 30, dtype=t.int32, device=dev)
        counts = t.zeros((B, 4), dtype=t.int32, device=dev)
        offs_p = offs.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(lib.cpx_finalize_tracks(
            h, C.byref(self.fp), offs_p, C.c_void_p(meta.ctypes.data), B, C.c_void_p(out.assoc.pool_dev.data_ptr()),
            C.c_void_p(out.assoc.tracks_dev.data_ptr()), C.c_void_p(out.assoc.ntracks_dev.data_ptr()),
            C.c_void_p(summ.data_ptr()), C.c_void_p(counts.data_ptr())))
        eng.synchronize()
        out.summaries_dev = summ
        prefix = (t.cumsum(counts, dim=0) - counts).to(t.int32).contiguous()
        totals = counts.sum(dim=0).cpu().numpy()
        out.counts = counts.cpu().numpy()
        n_tracks, n_refs, n_samples = int(totals[0]), int(totals[1]), int(totals[2])
        out.n_tracks, out.n_samples = n_tracks, n_samples
        if not classify or n_tracks == 0 or n_samples == 0:
            return out
        # ---- 4. segment plan ----
        per = self.sq * self.sq
        refs = t.zeros(max(n_refs, 1) * 6, dtype=t.int32, device=dev)
        toffs = t.zeros(n_tracks + 1, dtype=t.int32, device=dev)
        reqs = t.zeros(n_samples * per * 8, dtype=t.int32, device=dev)
        sample_track = t.zeros(n_samples, dtype=t.int32, device=dev)
        track_clip = t.zeros((n_tracks, 2), dtype=t.int32, device=dev)
        t.cuda.current_stream(dev).synchronize()
        self._check(lib.cpx_plan_segments(
            h, C.byref(self.fp), offs_p, C.c_void_p(meta.ctypes.data), B, C.c_void_p(out.assoc.pool_dev.data_ptr()),
            C.c_void_p(summ.data_ptr()), C.c_void_p(out.assoc.ntracks_dev.data_ptr()), C.c_void_p(prefix.data_ptr()),
            self.sq, C.c_void_p(refs.data_ptr()), C.c_void_p(toffs.data_ptr()), C.c_void_p(reqs.data_ptr()),
            C.c_void_p(sample_track.data_ptr()), C.c_void_p(track_clip.data_ptr())))
        eng.synchronize()
        toffs[n_tracks] = n_refs
        out.track_clip, out.reqs_dev, out.sample_track_dev = track_clip, reqs, sample_track
        # ---- 5. limits, then crop / tile + 6. CNN in chunks of samples ----
        limits = t.zeros(n_tracks * 4, dtype=t.int32, device=dev)
        t.cuda.current_stream(dev).synchronize()
        self._check(lib.cpx_track_limits_batch(
            h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(out.track.filtered_dev.data_ptr()),
            C.c_void_p(out.track.info_dev.data_ptr()), C.c_void_p(refs.data_ptr()), C.c_void_p(toffs.data_ptr()),
            n_tracks, C.c_void_p(limits.data_ptr())))
        side = self.sq * self.fs
        probs = t.empty((n_samples, self.n_labels), dtype=t.float32, device=dev)
        chunk = min(self.cnn_chunk, n_samples)
        if keep_samples:
            out.samples_dev = t.empty((n_samples, side, side, 2), dtype=t.float32, device=dev)
        elif self._sample_buf is None or self._sample_buf.shape[0] < chunk or self._sample_buf.shape[1] != side:
            self._sample_buf = t.empty((chunk, side, side, 2), dtype=t.float32, device=dev)
        for s0 in range(0, n_samples, chunk):
            s1 = min(s0 + chunk, n_samples)
            buf = out.samples_dev[s0:s1] if keep_samples else self._sample_buf[: s1 - s0]
            # requests of the chunk address samples relative to s0
            rq = reqs.view(-1, 8)[s0 * per : s1 * per].clone()
            rq[:, 6] -= s0
            self._check(lib.cpx_crop_tile(
                h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(out.track.filtered_dev.data_ptr()),
                C.c_void_p(out.track.info_dev.data_ptr()), C.c_void_p(rq.data_ptr()), (s1 - s0) * per,
                C.c_void_p(limits.data_ptr()), self.fs, self.sq, C.c_void_p(buf.data_ptr())))
            eng.synchronize()
            if self.net is not None:
                _, p = self.net.forward(buf)
                probs[s0:s1] = p
        if self.net is None:
            return out
        # ---- 7. per-track aggregation ----
        out.scores = t.empty((n_tracks, self.n_labels), dtype=t.float32, device=dev)
        out.best = t.empty(n_tracks, dtype=t.int32, device=dev)
        t.cuda.current_stream(dev).synchronize()
        self._check(lib.cpx_aggregate_predictions(
            h, C.c_void_p(probs.data_ptr()), C.c_void_p(sample_track.data_ptr()), n_samples,
            C.c_void_p(reqs.data_ptr()), n_tracks, self.n_labels, self.fp_index, self.sq,
            C.c_void_p(out.scores.data_ptr()), C.c_void_p(out.best.data_ptr())))
        eng.synchronize()
        out.probs = probs
        return out
