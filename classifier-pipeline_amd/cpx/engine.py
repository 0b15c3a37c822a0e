"""Device engine: owns a cpx handle and runs the track stage for batches of
clips.  PyTorch is used only as the device allocator / copy engine."""

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (COMPONENT_DTYPE, CROP_REQ_DTYPE, FRAME_INFO_DTYPE, FRAME_META_DTYPE, REGION_REF_DTYPE,
                   THUMB_STAT_DTYPE, TRACK_LIMITS_DTYPE, CpxError)
from .tracking import REGION_DTYPE, TRACK_RECORD_DTYPE, make_track_params, track_regions


def thresholds_for_model(model):
    """(background_thresh, weight_add) -- config/trackingmotionconfig.py:24-59,
    track/cliptrackextractor.py:124-127."""
    if model == "lepton3.5":
        return 50.0, 1.0
    return 20.0, 0.1


class TrackBatchResult:
    """Outputs of one cpx_track_batch call (host copies are made lazily).

    Lifetimes are the allocator's business, not the caller's: the engine allocates every output under its own stream
    (`TrackEngine._own_stream`), so a result dropped while its kernels are still running hands its blocks back to THAT
    stream's pool, where the next allocation is ordered behind those kernels; allocations on any other stream never see
    the blocks.  A result also keeps its engine's handle -- hence the stream -- alive: `TrackEngine.close()` on an
    engine with live results (a grown / sibling engine the parent closes) is deferred until the last of them dies.
    What stays the consumer's duty is torch's ordinary multi-stream rule: read the device tensors on the engine's
    stream or after `engine.synchronize()` (every host accessor here synchronises first)."""

    def __init__(self, engine, total, cap, comps, info, labels, filtered, background):
        self.engine, self.total, self.cap = engine, total, cap
        self.comps_dev, self.info_dev = comps, info
        self.labels_dev, self.filtered_dev, self.background_dev = labels, filtered, background
        self._info = self._comps = None
        engine._retain(self)

    @property
    def info(self):
        if self._info is None:
            self.engine.synchronize()
            self._info = self.info_dev.cpu().numpy().view(FRAME_INFO_DTYPE).reshape(-1)
        return self._info

    @property
    def comps(self):
        if self._comps is None:
            self.engine.synchronize()
            self._comps = self.comps_dev.cpu().numpy().view(COMPONENT_DTYPE).reshape(self.total, self.cap)
        return self._comps

    def check(self):
        bad = np.nonzero((self.info["frame_number"] >= 0) & (self.info["status"] != 0))[0]
        if bad.size:
            raise CpxError(int(self.info["status"][bad[0]]), "frame %d: %d components exceed capacity"
                           % (int(bad[0]), int(self.info["n_components"][bad[0]])))

    def overflowed(self, clip_offsets):
        """{clip index: components its fullest refused frame needs} for the clips with a frame beyond the engine's
        max_components (cpx_frame_info.status = CPX_ERR_OVERFLOW; n_components then holds the count found)."""
        info = self.info
        bad = np.nonzero((info["frame_number"] >= 0) & (info["status"] != 0))[0]
        out = {}
        if bad.size:
            offs = np.asarray(clip_offsets)
            for f in bad:
                b = int(np.searchsorted(offs, f, side="right") - 1)
                out[b] = max(out.get(b, 0), int(info["n_components"][f]))
        return out

    def components(self, f):
        n = int(self.info["n_components"][f])
        return self.comps[f, :n]

    def labels(self):
        self.engine.synchronize()
        return None if self.labels_dev is None else self.labels_dev.cpu().numpy()

    def filtered(self):
        self.engine.synchronize()
        return None if self.filtered_dev is None else self.filtered_dev.cpu().numpy()

    def background(self):
        self.engine.synchronize()
        return None if self.background_dev is None else self.background_dev.cpu().numpy()


class AssocBatchResult:
    """Outputs of cpx_associate_batch: every track ever created per clip (untrimmed,
    unfiltered -- the end-of-clip statistics are host work) + per-frame region lists."""

    def __init__(self, engine, offs, params, pool, tracks, ntracks, status, regions, rcounts):
        self.engine, self.offs, self.params = engine, offs, params
        self.pool_dev, self.tracks_dev, self.ntracks_dev, self.status_dev = pool, tracks, ntracks, status
        self.regions_dev, self.rcounts_dev = regions, rcounts
        self._host = None
        engine._retain(self)

    def _fetch(self):
        if self._host is None:
            self.engine.synchronize()
            ma, mt = self.params.max_active_tracks, self.params.max_tracks
            pool = self.pool_dev.cpu().numpy().view(REGION_DTYPE).reshape(-1, ma)
            tracks = self.tracks_dev.cpu().numpy().view(TRACK_RECORD_DTYPE).reshape(-1, mt)
            ntr = self.ntracks_dev.cpu().numpy()
            status = self.status_dev.cpu().numpy()
            regions = rc = None
            if self.regions_dev is not None:
                regions = self.regions_dev.cpu().numpy().view(REGION_DTYPE).reshape(-1, self.engine.cap)
                rc = self.rcounts_dev.cpu().numpy()
            self._host = (pool, tracks, ntr, status, regions, rc)
        return self._host

    def check(self, clip=None):
        """Raise when the association of any clip (or of `clip` only) ran out of track slots."""
        status = self._fetch()[3]
        bad = np.nonzero(status != 0)[0] if clip is None else (np.array([clip]) if status[clip] != 0 else np.array([], int))
        if bad.size:
            raise CpxError(int(status[bad[0]]), "clip %d: track capacity exceeded" % int(bad[0]))

    def overflowed(self):
        """Clips whose association ran out of simultaneous (max_active_tracks) or total (max_tracks) track slots."""
        return [int(b) for b in np.nonzero(self._fetch()[3] != 0)[0]]

    def clip_tracks(self, b):
        """-> list of (track_record, regions[n_frames]) for clip b, in creation (id) order."""
        pool, tracks, ntr, _, _, _ = self._fetch()
        ma = self.params.max_active_tracks
        f0, f1 = int(self.offs[b]), int(self.offs[b + 1])
        cpool = pool[f0:f1].reshape(-1)
        return [(tracks[b, i], track_regions(cpool, tracks[b, i], ma)) for i in range(int(ntr[b]))]

    def frame_regions(self, f):
        _, _, _, _, regions, rc = self._fetch()
        return regions[f, : int(rc[f])]


class TrackStream:
    """One clip tracked incrementally.  Owns the clip's device buffers (frames, per-frame outputs, region pool,
    track records); append() uploads a frame and runs the track (+ association) kernels for it."""

    def __init__(self, engine, capacity, params, want_labels):
        t = engine.torch
        self.engine, self.cap_frames, self.params = engine, int(capacity), params
        if self.cap_frames > engine.cfg.max_frames + 1:
            raise ValueError("stream capacity %d exceeds the engine's max_frames %d"
                             % (self.cap_frames, engine.cfg.max_frames))
        dev, H, W, cap = engine.device, engine.height, engine.width, engine.cap
        n = self.cap_frames
        self.frames_dev = t.zeros((n, H, W), dtype=t.int16, device=dev)
        self.comps = t.zeros(n * cap * 8, dtype=t.int32, device=dev)
        self.info = t.zeros(n * 20, dtype=t.int32, device=dev)
        self.labels = t.zeros((n, H, W), dtype=t.int32, device=dev) if want_labels else None
        self.filtered = t.zeros((n, H, W), dtype=t.float32, device=dev)
        self.background = t.zeros((1, H, W), dtype=t.float32, device=dev)
        ma, mt = params.max_active_tracks, params.max_tracks
        self.pool = t.zeros(n * ma * 14, dtype=t.int32, device=dev)
        self.tracks = t.zeros(mt * 8, dtype=t.int32, device=dev)
        self.ntracks = t.zeros(1, dtype=t.int32, device=dev)
        self.status = t.zeros(1, dtype=t.int32, device=dev)
        self.regions = t.zeros(n * cap * 14, dtype=t.int32, device=dev)
        self.rcounts = t.zeros(n, dtype=t.int32, device=dev)
        self.meta = np.zeros(n, dtype=FRAME_META_DTYPE)
        self.n = 0              # frames in the buffer
        self.n_tracked = 0      # frames the track kernels have consumed
        self.result = TrackBatchResult(engine, n, cap, self.comps, self.info, self.labels, self.filtered,
                                       self.background)

    def _p(self, tensor):
        return C.c_void_p(tensor.data_ptr() if tensor is not None else None)

    def append(self, pix, time_on=None, last_ffc=None, init_only=False, associate=True, flags=0):
        """Upload one frame and process it.  init_only: the frame only initialises the background
        (a CPTV background frame).  flags: _lib.TRACK_* (who owns the background, include/cpx.h).
        Returns the frame's index in the stream's arrays."""
        eng, t = self.engine, self.engine.torch
        if self.n >= self.cap_frames:
            raise CpxError(-1, "stream is full (%d frames): open the extractor with a larger max_frames"
                           % self.cap_frames)
        f = self.n
        a = np.ascontiguousarray(pix, dtype=np.uint16)
        self.frames_dev[f].copy_(t.from_numpy(a.view(np.int16)))
        m = self.meta[f]
        m["background_frame"] = 1 if init_only else 0
        if time_on is not None and last_ffc is not None:
            m["time_on_ms"], m["last_ffc_ms"], m["has_times"] = time_on, last_ffc, 1
        self.n = f + 1
        if init_only and f == 0:
            return f  # consumed together with the first real frame
        eng.sync_inputs()
        eng.track_calls = getattr(eng, "track_calls", 0) + 1
        mp = C.c_void_p(self.meta.ctypes.data)
        rc = eng.lib.cpx_track_frame_ex(eng.h, self._p(self.frames_dev), mp, self.n_tracked, self.n,
                                        self._p(self.comps), self._p(self.info), self._p(self.labels),
                                        self._p(self.filtered), self._p(self.background), int(flags))
        if rc != 0:
            raise CpxError(rc, eng._err())
        n_prev = self.n_tracked
        self.n_tracked = self.n
        if associate and not init_only:
            rc = eng.lib.cpx_associate_frame(eng.h, C.byref(self.params), mp, n_prev, self.n, self._p(self.comps),
                                             self._p(self.info), self._p(self.pool), self._p(self.tracks),
                                             self._p(self.ntracks), self._p(self.status), self._p(self.regions),
                                             self._p(self.rcounts))
            if rc != 0:
                raise CpxError(rc, eng._err())
        eng.synchronize()
        return f

    def replay(self, old, flags=0, associate=True):
        """Take over the frames another stream of the same clip has consumed (a stream whose capacities turned out too
        small) and run them again here, in one call per stage: this stream then stands where the old one stood."""
        eng, n = self.engine, old.n
        if n > self.cap_frames:
            raise CpxError(-1, "replay of %d frames into a stream of %d" % (n, self.cap_frames))
        self.frames_dev[:n].copy_(old.frames_dev[:n])
        self.meta[:n] = old.meta[:n]
        self.n = n
        if n == 0 or (n == 1 and self.meta[0]["background_frame"]):
            return
        eng.torch.cuda.current_stream(eng.device).synchronize()
        eng.track_calls = getattr(eng, "track_calls", 0) + 1
        mp = C.c_void_p(self.meta.ctypes.data)
        rc = eng.lib.cpx_track_frame_ex(eng.h, self._p(self.frames_dev), mp, 0, n, self._p(self.comps), self._p(self.info),
                                        self._p(self.labels), self._p(self.filtered), self._p(self.background), int(flags))
        if rc != 0:
            raise CpxError(rc, eng._err())
        self.n_tracked = n
        if associate:
            rc = eng.lib.cpx_associate_frame(eng.h, C.byref(self.params), mp, 0, n, self._p(self.comps), self._p(self.info),
                                             self._p(self.pool), self._p(self.tracks), self._p(self.ntracks),
                                             self._p(self.status), self._p(self.regions), self._p(self.rcounts))
            if rc != 0:
                raise CpxError(rc, eng._err())
        eng.synchronize()

    def overflow(self):
        """(components the fullest refused frame needs or 0, association out of track slots) over the frames so far."""
        info = self.info[: self.n * 20].cpu().numpy().view(FRAME_INFO_DTYPE)
        bad = (info["frame_number"] >= 0) & (info["status"] != 0)
        need = int(info["n_components"][bad].max()) if bad.any() else 0
        return need, int(self.status.item()) != 0

    def frame_info(self, f):
        return self.info[f * 20:(f + 1) * 20].cpu().numpy().view(FRAME_INFO_DTYPE)[0]

    def frame_regions(self, f):
        n = int(self.rcounts[f].item())
        cap = self.engine.cap
        return self.regions[f * cap * 14:(f * cap + n) * 14].cpu().numpy().view(REGION_DTYPE)

    def track_records(self):
        n = int(self.ntracks.item())
        st = int(self.status.item())
        if st != 0:
            raise CpxError(st, "track capacity exceeded")
        return self.tracks[: n * 8].cpu().numpy().view(TRACK_RECORD_DTYPE)

    def pool_row(self, q):
        """Regions written for processed frame number q, one per active-track slot."""
        ma = self.params.max_active_tracks
        return self.pool[q * ma * 14:(q + 1) * ma * 14].cpu().numpy().view(REGION_DTYPE)


class ComponentStream:
    """Association of ONE clip frame by frame from component lists the caller supplies (the IR tracker: its
    components come out of the background subtractor + cpx_ir_detect + the host's fragment merge, not out of
    cpx_track_frame).  Same device core and outputs as TrackStream: cpx_associate_frame."""

    def __init__(self, engine, capacity, params):
        t = engine.torch
        self.engine, self.cap_frames, self.params = engine, int(capacity), params
        if self.cap_frames > engine.cfg.max_frames:
            raise ValueError("stream capacity %d exceeds the engine's max_frames %d" % (capacity, engine.cfg.max_frames))
        dev, cap, n = engine.device, engine.cap, self.cap_frames
        ma, mt = params.max_active_tracks, params.max_tracks
        self.comps = t.zeros(n * cap * 8, dtype=t.int32, device=dev)
        self.info = t.zeros(n * 20, dtype=t.int32, device=dev)
        self.pool = t.zeros(n * ma * 14, dtype=t.int32, device=dev)
        self.tracks = t.zeros(mt * 8, dtype=t.int32, device=dev)
        self.ntracks = t.zeros(1, dtype=t.int32, device=dev)
        self.status = t.zeros(1, dtype=t.int32, device=dev)
        self.regions = t.zeros(n * cap * 14, dtype=t.int32, device=dev)
        self.rcounts = t.zeros(n, dtype=t.int32, device=dev)
        self.meta = np.zeros(n, dtype=FRAME_META_DTYPE)
        self.n = 0

    def append(self, components, ffc_affected=False):
        """components: COMPONENT_DTYPE rows of this frame (label order = region ids).  -> the frame's index."""
        eng, t = self.engine, self.engine.torch
        if self.n >= self.cap_frames:
            raise CpxError(-1, "stream is full (%d frames)" % self.cap_frames)
        comps = np.ascontiguousarray(components, dtype=COMPONENT_DTYPE)
        if len(comps) > eng.cap:
            raise CpxError(-5, "%d components exceed the engine's max_components %d" % (len(comps), eng.cap))
        f = self.n
        if len(comps):
            self.comps[f * eng.cap * 8:(f * eng.cap + len(comps)) * 8] = t.from_numpy(comps.view(np.int32).reshape(-1).copy()).to(eng.device)
        fi = np.zeros(1, FRAME_INFO_DTYPE)
        fi["frame_number"], fi["n_components"], fi["ffc_affected"] = f, len(comps), 1 if ffc_affected else 0
        self.info[f * 20:(f + 1) * 20] = t.from_numpy(fi.view(np.int32).copy()).to(eng.device)
        if ffc_affected:  # the schedule derives the flag from the times: less than 9 ms apart
            self.meta[f]["time_on_ms"], self.meta[f]["last_ffc_ms"], self.meta[f]["has_times"] = 100, 100, 1
        self.n = f + 1
        eng.sync_inputs()
        p = lambda x: C.c_void_p(x.data_ptr())
        rc = eng.lib.cpx_associate_frame(eng.h, C.byref(self.params), C.c_void_p(self.meta.ctypes.data), f, self.n,
                                         p(self.comps), p(self.info), p(self.pool), p(self.tracks), p(self.ntracks),
                                         p(self.status), p(self.regions), p(self.rcounts))
        if rc != 0:
            raise CpxError(rc, eng._err())
        eng.synchronize()
        return f

    frame_regions = TrackStream.frame_regions
    track_records = TrackStream.track_records
    pool_row = TrackStream.pool_row


class TrackEngine:
    def __init__(self, width=160, height=120, model="lepton3", device=0, edge_pixels=1, window=45,
                 max_components=64, max_frames=4096, background_thresh=None, weight_add=None, denoise=False):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cpx.TrackEngine needs a HIP device (no CPU fallback)")
        bt, wa = thresholds_for_model(model)
        self.cfg = _lib.Config(width, height, edge_pixels, window,
                               float(bt if background_thresh is None else background_thresh),
                               float(wa if weight_add is None else weight_add),
                               max_components, max_frames, 1 if denoise else 0, 0)
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        rc = self.lib.cpx_create(device, C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise CpxError(rc, "cpx_create")
        self.width, self.height, self.cap = width, height, max_components
        self._ctor = dict(width=width, height=height, model=model, device=device, edge_pixels=edge_pixels,
                          background_thresh=self.cfg.background_thresh, weight_add=self.cfg.weight_add,
                          max_frames=max_frames, denoise=denoise)
        self._grown = {}
        self._live_results = 0      # results whose device tensors live in this handle's stream pool
        self._close_deferred = False

    def close(self):
        """Destroys the handle (and its HIP stream) -- at once when no result of this engine is alive, else when the last
        one dies: their tensors were allocated under the stream and must be handed back to a stream that still exists."""
        for eng in getattr(self, "_grown", {}).values():
            eng.close()
        self._grown = {}
        if getattr(self, "_live_results", 0) > 0:
            # the stream stays for the results' buffers; what the handle itself holds on the device (the network's activation
            # arena -- tens of GB --, workspaces) goes now: a closed engine that lingers must not cost the next one its memory
            self._close_deferred = True
            if self.h:
                self.lib.cpx_release_memory(self.h)
            return
        self._destroy()

    def _destroy(self):
        if self.h:
            self.lib.cpx_destroy(self.h)
            self.h = C.c_void_p()
        self._torch_stream = None

    def _retain(self, result):
        """`result` holds device tensors of this engine's stream: keep the stream until it is collected."""
        import weakref

        self._live_results += 1
        weakref.finalize(result, self._release).atexit = False   # (the bound method keeps the engine alive until then)

    def _release(self):
        self._live_results -= 1
        # a result of a TRACK_DEFER_MEDIANS call dropped before anything read its medians: its blocks return to this stream's
        # pool while the median kernel may still write them from the second stream -- order the stream behind it (a stream
        # wait; nothing pending: a no-op)
        if self.h:
            self.lib.cpx_join_medians(self.h)
        if self._live_results == 0 and self._close_deferred:
            self._close_deferred = False
            try:
                self._destroy()
            except Exception:
                pass

    def _own_stream(self):
        """Context for allocating a call's outputs: order the handle's stream behind what the caller has enqueued on
        torch's current stream (sync_inputs), then make the handle's stream current, so that the outputs' blocks belong
        to its pool and zero-fills run on it."""
        self.sync_inputs()
        return self.torch.cuda.stream(self.torch_stream())

    # ---- no recording is lost to a capacity -------------------------------------------------------------------
    # The reference has no limit on components per frame (cliptrackextractor.py:236-247), simultaneous tracks or tracks
    # per clip (cliptracker.py:202-247).  The kernels work on caller-sized tables and report CPX_ERR_OVERFLOW with the
    # count they found; the host layer owns the sizes, so it is the one that grows them: the clip that did not fit is
    # run again, alone, on a sibling handle with the capacities it needs (doubled until nothing overflows).
    def MAX_COMPONENTS_EVER(self):
        return ((self.height + 1) // 2) * ((self.width + 1) // 2)  # at most one 8-connected component per 2 x 2 block

    def grown(self, max_components, max_frames=None):
        """A sibling engine (same geometry / thresholds / denoise) whose per-frame component capacity is the power of
        two >= max_components (capped at one component per 2 x 2 block); cached, closed with this engine."""
        cap = self.grown_capacity(max_components)
        eng = self._grown.get(cap)
        if eng is None or (max_frames and eng.cfg.max_frames < max_frames):
            if eng is not None:
                eng.close()
            eng = self.sibling(cap, max_frames)
            self._grown[cap] = eng
        return eng

    def grown_capacity(self, max_components):
        cap = 128
        while cap < max_components:
            cap *= 2
        return max(min(cap, self.MAX_COMPONENTS_EVER()), int(max_components))

    def sibling(self, max_components, max_frames=None):
        """An independent engine (a handle of its own) with this one's geometry / thresholds / denoise."""
        args = dict(self._ctor)
        if max_frames:
            args["max_frames"] = max(int(max_frames), args["max_frames"])
        return TrackEngine(max_components=int(max_components), **args)

    def track_clip_grown(self, frames_dev, meta, params=None, need_components=0, want_labels=False, want_filtered=True,
                         want_background=False, flags=0, want_regions=True, associate=True):
        """One clip (device frames [n, H, W], its cpx_frame_meta rows) tracked -- and, with `associate`, associated
        with track capacities of `params` -- on engines grown until neither stage reports CPX_ERR_OVERFLOW.
        -> (engine used, TrackBatchResult, AssocBatchResult or None, params used)."""
        from .tracking import TrackParams

        n = int(frames_dev.shape[0])
        offs = np.array([0, n], np.int32)
        need = max(int(need_components), self.cap)
        params = params or make_track_params(self.width, self.height, self.cfg.edge_pixels)
        while True:
            eng = self if need <= self.cap else self.grown(need, max_frames=n)
            res = eng.track_batch(frames_dev, offs, meta, want_labels=want_labels, want_filtered=want_filtered,
                                  want_background=want_background, flags=flags)
            over = res.overflowed(offs)
            if over:
                if over[0] <= eng.cap:
                    raise CpxError(-5, "a frame reports %d components and still does not fit %d" % (over[0], eng.cap))
                need = over[0]
                continue
            if not associate:
                return eng, res, None, params
            while True:
                assoc = eng.associate_batch(res, offs, meta, params=params, want_regions=want_regions)
                if not assoc.overflowed():
                    return eng, res, assoc, params
                if params.max_active_tracks >= 65536:
                    raise CpxError(-5, "track capacity exceeded at 65536 simultaneous tracks")
                grown = TrackParams.from_buffer_copy(params)
                grown.max_active_tracks = params.max_active_tracks * 2
                grown.max_tracks = params.max_tracks * 2
                params = grown

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return (self.lib.cpx_last_error(self.h) or b"").decode()

    def torch_stream(self):
        """The handle's HIP stream as a torch stream (event record / wait between two engines)."""
        if getattr(self, "_torch_stream", None) is None:
            self._torch_stream = self.torch.cuda.ExternalStream(int(self.lib.cpx_stream(self.h)), device=self.device)
        return self._torch_stream

    def synchronize(self):
        rc = self.lib.cpx_synchronize(self.h)
        if rc != 0:
            raise CpxError(rc, self._err())

    def sync_inputs(self):
        """Order this handle's next kernels behind what torch has enqueued: a host wait on torch's current stream --
        unless that stream IS the handle's (``with torch.cuda.stream(engine.torch_stream())``), where stream order
        already does it and nothing blocks the host (cpx.pipeline runs that way)."""
        cur = self.torch.cuda.current_stream(self.device)
        if int(cur.cuda_stream) != int(self.lib.cpx_stream(self.h) or 0):
            cur.synchronize()

    def upload_frames(self, frames):
        """uint16 [N,H,W] numpy -> device tensor (stored as int16 bits)."""
        t = self.torch
        a = np.ascontiguousarray(frames, dtype=np.uint16)
        return t.from_numpy(a.view(np.int16)).to(self.device)

    def cptv_unpack(self, payload, frame_offsets, bit_widths, clip_offsets):
        """Inflated CPTV bytes + section index (cpx.cptv.CptvReader.scan) -> device frames [total,H,W]
        (uint16 bits in an int16 tensor), decoded by cpx_cptv_unpack."""
        t = self.torch
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        total = int(offs[-1])
        fo = np.ascontiguousarray(frame_offsets, dtype=np.int64)
        bw = np.ascontiguousarray(bit_widths, dtype=np.int32)
        assert fo.size == total and bw.size == total
        pay = t.from_numpy(np.array(payload, dtype=np.uint8, copy=True)).to(self.device)
        fo_d, bw_d, offs_d = (t.from_numpy(a).to(self.device) for a in (fo, bw, offs))
        out = t.empty((total, self.height, self.width), dtype=t.int16, device=self.device)
        self.sync_inputs()
        rc = self.lib.cpx_cptv_unpack(self.h, C.c_void_p(pay.data_ptr()), C.c_void_p(fo_d.data_ptr()),
                                      C.c_void_p(bw_d.data_ptr()), C.c_void_p(offs_d.data_ptr()), offs.size - 1,
                                      C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise CpxError(rc, self._err())
        self.synchronize()  # the staging tensors go out of scope here
        return out

    # ---- incremental tracking (one clip, frame by frame) ---------------------------------------------------
    def open_stream(self, capacity, params=None, want_labels=True):
        """Device buffers of a frame-by-frame tracked clip (cpx_track_frame / cpx_associate_frame)."""
        return TrackStream(self, capacity, params or make_track_params(self.width, self.height, self.cfg.edge_pixels),
                           want_labels)

    def thumb_stats(self, frames_dev, track_result, refs):
        """cpx_thumb_stats over REGION_REF_DTYPE refs -> THUMB_STAT_DTYPE array (host)."""
        t = self.torch
        if track_result.labels_dev is None:
            raise ValueError("thumb_stats needs the labels output of track_batch (want_labels=True)")
        n = len(refs)
        if n == 0:
            return np.zeros(0, THUMB_STAT_DTYPE)
        refs = np.ascontiguousarray(refs, dtype=REGION_REF_DTYPE)
        refs_dev = self._to_dev(refs)
        out = t.zeros(n * 4, dtype=t.int32, device=self.device)
        self.sync_inputs()
        args = (C.c_void_p(frames_dev.data_ptr()), C.c_void_p(track_result.labels_dev.data_ptr()),
                C.c_void_p(track_result.info_dev.data_ptr()))
        # Tiers: the kernel keeps a region's mask and border chain in LDS, one wavefront per region, and a CU works on
        # as many regions as fit its LDS -- sized for ordinary regions (<= 46 x 46, the animals of the path) seven run
        # per CU, at 80 x 80 three, in the whole-frame form one.  A region goes to the first tier that holds it; one
        # whose border outgrows a tier's chain is marked by the kernel and moves on to the next.
        got = np.zeros(n, THUMB_STAT_DTYPE)
        pending = np.arange(n)
        for side, chain in self.THUMB_TIERS + ((None, None),):
            if pending.size == 0:
                break
            if side is None:
                take = pending
            else:
                fits = (refs["width"][pending] <= side) & (refs["height"][pending] <= side)
                take = pending[fits]
            if take.size == 0:
                continue
            whole = take.size == n
            sub_dev = refs_dev if whole else self._to_dev(np.ascontiguousarray(refs[take]))
            sub_out = out if whole else t.zeros(take.size * 4, dtype=t.int32, device=self.device)
            self.sync_inputs()
            if side is None:
                rc = self.lib.cpx_thumb_stats(self.h, *args, C.c_void_p(sub_dev.data_ptr()), int(take.size),
                                              C.c_void_p(sub_out.data_ptr()))
            else:
                rc = self.lib.cpx_thumb_stats_ex(self.h, *args, C.c_void_p(sub_dev.data_ptr()), int(take.size),
                                                 C.c_void_p(sub_out.data_ptr()), side, side, chain)
            if rc != 0:
                raise CpxError(rc, self._err())
            self.synchronize()
            res = sub_out.cpu().numpy().view(THUMB_STAT_DTYPE)
            got[take] = res
            done = take[res["status"] == 0] if side is not None else take
            pending = np.setdiff1d(pending, done, assume_unique=True)
        bad = np.nonzero(got["status"] != 0)[0]
        if bad.size:
            raise CpxError(int(got["status"][bad[0]]), "region %d: contour longer than the kernel's chain capacity"
                           % int(bad[0]))
        return got

    THUMB_TIERS = ((46, 2304), (80, 4608))

    CNN_MATH = {"f32": 0, "bf16x3": 1, "bf16x2": 2, "fp16x2": 3}
    DEFAULT_CNN_MATH = "fp16x2"  # what cpx_create sets unless CPX_CNN_MATH says otherwise

    def set_cnn_math(self, mode):
        """"f32": v_mfma_f32_32x32x2_f32; "bf16x3": exact three-way bf16 split of the float32 operands on the bf16
        matrix pipe (six products); "bf16x2" (opt-in): stages 2-4 on two rounded bf16 planes and three products -- not
        float32 per element (<= 3 x 2^-16), logits within the same bound; "fp16x2" (default): the same layers on two
        rounded fp16 planes (11 + 11 bits: 2^-22 per operand, the float32 kernels' own error level against a float64
        convolution), operands scaled by powers of two into fp16's range, a device-side rerun in bf16x3 if an
        activation leaves it (include/cpx.h: cpx_set_cnn_math).  Same inputs, outputs and logit tolerance."""
        rc = self.lib.cpx_set_cnn_math(self.h, self.CNN_MATH[mode])
        if rc != 0:
            raise CpxError(rc, self._err())

    def cnn_last_overflow(self):
        """fp16x2: True when the last forward / convolution on this engine fell back to the bf16x3 kernels because an
        activation left fp16's range (cpx_cnn_last_overflow; synchronises)."""
        out = C.c_int(0)
        rc = self.lib.cpx_cnn_last_overflow(self.h, C.byref(out))
        if rc != 0:
            raise CpxError(rc, self._err())
        return bool(out.value)

    def cnn_overflow_forwards(self, reset=False):
        """fp16x2: forwards on this engine that fell back to the bf16x3 kernels so far (cpx_cnn_overflow_forwards)."""
        out = C.c_int(0)
        rc = self.lib.cpx_cnn_overflow_forwards(self.h, C.byref(out), 1 if reset else 0)
        if rc != 0:
            raise CpxError(rc, self._err())
        return int(out.value)

    def get_cnn_math(self):
        return {v: k for k, v in self.CNN_MATH.items()}[self.lib.cpx_get_cnn_math(self.h)]

    def ir_detect(self, images_dev, threshold=0, max_components=1024, want_labels=False):
        """cpx_ir_detect over uint8 [n, H, W] device frames -> (counts int32[n], components COMPONENT_DTYPE
        [n, max(counts)] (host; entries past counts[i] are unspecified), labels_dev int32 [n, H, W] or None).
        Raises CpxError(CPX_ERR_OVERFLOW) when a frame has more than max_components components."""
        t = self.torch
        if images_dev.dtype != t.uint8 or images_dev.dim() != 3 or not images_dev.is_contiguous():
            raise ValueError("ir_detect wants a contiguous uint8 [n, H, W] device tensor")
        n, H, W = (int(v) for v in images_dev.shape)
        with self._own_stream():
            comps = t.empty((n, max_components, COMPONENT_DTYPE.itemsize // 4), dtype=t.int32, device=self.device)
            counts = t.zeros(n, dtype=t.int32, device=self.device)
            status = t.zeros(n, dtype=t.int32, device=self.device)
            labels = t.empty((n, H, W), dtype=t.int32, device=self.device) if want_labels else None
        rc = self.lib.cpx_ir_detect(self.h, C.c_void_p(images_dev.data_ptr()), n, W, H, int(threshold),
                                    int(max_components), C.c_void_p(comps.data_ptr()), C.c_void_p(counts.data_ptr()),
                                    C.c_void_p(status.data_ptr()),
                                    C.c_void_p(labels.data_ptr()) if want_labels else None)
        if rc != 0:
            raise CpxError(rc, self._err())
        self.synchronize()
        st = status.cpu().numpy()
        cnt = counts.cpu().numpy()
        bad = np.nonzero(st != 0)[0]
        if bad.size:
            raise CpxError(int(st[bad[0]]), "frame %d has %d components, max_components is %d"
                           % (int(bad[0]), int(cnt[bad[0]]), max_components))
        used = max(1, int(cnt.max()))  # only the filled part of the table crosses PCIe
        host = comps[:, :used].contiguous().cpu().numpy().view(COMPONENT_DTYPE).reshape(n, used)
        return cnt, host, labels

    def ir_resize_area(self, images_dev, factor, pad_width_to=1):
        """cpx_ir_resize_area: cv2.resize(..., interpolation=cv2.INTER_AREA) by an integer factor, uint8 [n, H, W] (or
        [H, W]) on the device -> uint8 [n, H / factor, W / factor]; pad_width_to = 64: zero columns on the right up
        to a multiple of 64 (what cpx_ir_detect's bit rows want).  Everything runs on the handle's stream."""
        t = self.torch
        single = images_dev.dim() == 2
        src = (images_dev[None] if single else images_dev).contiguous()
        n, H, W = (int(v) for v in src.shape)
        with self._own_stream():
            out = t.empty((n, H // factor, W // factor), dtype=t.uint8, device=self.device)
        rc = self.lib.cpx_ir_resize_area(self.h, C.c_void_p(src.data_ptr()), n, W, H, int(factor), C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise CpxError(rc, self._err())
        wo = W // factor
        if pad_width_to > 1 and wo % pad_width_to:
            with t.cuda.stream(self.torch_stream()):
                wide = t.zeros((n, H // factor, (wo + pad_width_to - 1) // pad_width_to * pad_width_to), dtype=t.uint8,
                               device=self.device)
                wide[:, :, :wo] = out
            out = wide
        return out[0] if single else out

    def ir_delta_variance(self, cur_dev, prev_dev, rects):
        """cpx_ir_delta_variance: np.var of the uint8-wrapping frame difference over each [x, y, w, h] box -> float64 [n]."""
        t = self.torch
        rects = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
        n = len(rects)
        if n == 0:
            return np.zeros(0, np.float64)
        H, W = (int(v) for v in cur_dev.shape)
        r_dev = t.from_numpy(rects).to(self.device)
        out = t.empty(n, dtype=t.float64, device=self.device)
        self.sync_inputs()
        rc = self.lib.cpx_ir_delta_variance(self.h, C.c_void_p(cur_dev.data_ptr()), C.c_void_p(prev_dev.data_ptr()), W, H,
                                            C.c_void_p(r_dev.data_ptr()), n, C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise CpxError(rc, self._err())
        self.synchronize()
        return out.cpu().numpy()

    def trackless_thumb(self, frames_dev, frame, background):
        """cpx_trackless_thumb -> (x, y) of the chosen 64x64 window."""
        t = self.torch
        out = t.zeros(2, dtype=t.int32, device=self.device)
        self.sync_inputs()
        rc = self.lib.cpx_trackless_thumb(self.h, C.c_void_p(frames_dev.data_ptr()), int(frame), int(background),
                                          C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise CpxError(rc, self._err())
        self.synchronize()
        x, y = out.cpu().numpy()
        return int(x), int(y)

    @staticmethod
    def make_meta(n, time_on=None, last_ffc=None, background=None):
        m = np.zeros(n, dtype=FRAME_META_DTYPE)
        if time_on is not None:
            for i in range(n):
                if time_on[i] is not None and last_ffc[i] is not None:
                    m["time_on_ms"][i] = time_on[i]
                    m["last_ffc_ms"][i] = last_ffc[i]
                    m["has_times"][i] = 1
        if background is not None:
            m["background_frame"] = np.asarray(background, dtype=np.int32)
        return m

    def set_background(self, clip, background, weights=None, average=None):
        """Stage a WeightedBackground state (host arrays: background [H,W], weights [H-2e,W-2e] or None, average) for
        clip `clip` of the next track call (cpx_set_background)."""
        bg = np.ascontiguousarray(background, dtype=np.float32)
        if bg.shape != (self.height, self.width):
            raise ValueError("background must be [%d, %d]" % (self.height, self.width))
        e = self.cfg.edge_pixels
        w = None
        if weights is not None:
            w = np.ascontiguousarray(weights, dtype=np.float64)
            if w.shape != (self.height - 2 * e, self.width - 2 * e):
                raise ValueError("weights must cover the cropped interior")
        if average is None:
            average = float(np.average(bg[e:self.height - e, e:self.width - e]))
        rc = self.lib.cpx_set_background(self.h, int(clip), C.c_void_p(bg.ctypes.data),
                                         C.c_void_p(w.ctypes.data if w is not None else None), float(average))
        if rc != 0:
            raise CpxError(rc, self._err())

    def get_background(self, clip=0):
        """-> (background float32 [H,W], weights float64 [H-2e,W-2e], average) the last track call left for `clip`."""
        e = self.cfg.edge_pixels
        bg = np.empty((self.height, self.width), np.float32)
        w = np.empty((self.height - 2 * e, self.width - 2 * e), np.float64)
        avg = C.c_double()
        rc = self.lib.cpx_get_background(self.h, int(clip), C.c_void_p(bg.ctypes.data), C.c_void_p(w.ctypes.data),
                                         C.byref(avg))
        if rc != 0:
            raise CpxError(rc, self._err())
        return bg, w, float(avg.value)

    def track_batch(self, frames_dev, clip_offsets, meta, want_labels=False, want_filtered=False,
                    want_background=False, outputs=None, flags=0):
        """frames_dev: device tensor [total,H,W] of uint16 bits; clip_offsets: int32 [B+1].
        flags: _lib.TRACK_* (who owns the background, include/cpx.h)."""
        t = self.torch
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        B = offs.size - 1
        total = int(offs[-1])
        meta = np.ascontiguousarray(meta, dtype=FRAME_META_DTYPE)
        assert meta.size == total and frames_dev.shape[0] >= total
        if outputs is None:
            with self._own_stream():
                comps = t.empty(total * self.cap * 8, dtype=t.int32, device=self.device)
                info = t.empty(total * 20, dtype=t.int32, device=self.device)
                labels = t.empty((total, self.height, self.width), dtype=t.int32, device=self.device) if want_labels else None
                filt = t.empty((total, self.height, self.width), dtype=t.float32, device=self.device) if want_filtered else None
                bgo = t.empty((B, self.height, self.width), dtype=t.float32, device=self.device) if want_background else None
        else:
            comps, info, labels, filt, bgo = outputs
            self.sync_inputs()  # inputs were produced on torch's stream
        self.track_calls = getattr(self, "track_calls", 0) + 1  # whose state cpx_get_background would read
        rc = self.lib.cpx_track_batch_ex(
            self.h, C.c_void_p(frames_dev.data_ptr()), offs.ctypes.data_as(C.POINTER(C.c_int32)),
            C.c_void_p(meta.ctypes.data), B, C.c_void_p(comps.data_ptr()), C.c_void_p(info.data_ptr()),
            C.c_void_p(labels.data_ptr() if labels is not None else None),
            C.c_void_p(filt.data_ptr() if filt is not None else None),
            C.c_void_p(bgo.data_ptr() if bgo is not None else None), int(flags))
        if rc != 0:
            raise CpxError(rc, self._err())
        return TrackBatchResult(self, total, self.cap, comps, info, labels, filt, bgo)

    def associate_batch(self, track_result, clip_offsets, meta, params=None, want_regions=True):
        """Region filter + matching + Kalman for the clips of a track_batch result."""
        t = self.torch
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        B = offs.size - 1
        total = int(offs[-1])
        meta = np.ascontiguousarray(meta, dtype=FRAME_META_DTYPE)
        params = params or make_track_params(self.width, self.height, self.cfg.edge_pixels)
        ma, mt = params.max_active_tracks, params.max_tracks
        with self._own_stream():
            pool = t.zeros(total * ma * 14, dtype=t.int32, device=self.device)
            tracks = t.zeros(B * mt * 8, dtype=t.int32, device=self.device)
            ntr = t.zeros(B, dtype=t.int32, device=self.device)
            status = t.zeros(B, dtype=t.int32, device=self.device)
            regions = t.zeros(total * self.cap * 14, dtype=t.int32, device=self.device) if want_regions else None
            rcounts = t.zeros(total, dtype=t.int32, device=self.device) if want_regions else None
        rc = self.lib.cpx_associate_batch(
            self.h, C.byref(params), offs.ctypes.data_as(C.POINTER(C.c_int32)), C.c_void_p(meta.ctypes.data), B,
            C.c_void_p(track_result.comps_dev.data_ptr()), C.c_void_p(track_result.info_dev.data_ptr()),
            C.c_void_p(pool.data_ptr()), C.c_void_p(tracks.data_ptr()), C.c_void_p(ntr.data_ptr()),
            C.c_void_p(status.data_ptr()),
            C.c_void_p(regions.data_ptr() if regions is not None else None),
            C.c_void_p(rcounts.data_ptr() if rcounts is not None else None))
        if rc != 0:
            raise CpxError(rc, self._err())
        return AssocBatchResult(self, offs, params, pool, tracks, ntr, status, regions, rcounts)

    def _to_dev(self, arr):
        """structured / plain numpy array -> device int32 tensor holding the same bytes."""
        a = np.ascontiguousarray(arr)
        return self.torch.from_numpy(a.view(np.int32).reshape(-1).copy()).to(self.device)

    def preprocess_segments(self, frames_dev, track_result, refs, track_offsets, reqs, n_samples, frame_size=32,
                            square_width=5, out=None, limits_flags=0):
        """get_limits + clip test per track, then crop / resize / normalise / tile every request.
        refs: REGION_REF_DTYPE [R]; track_offsets: int32 [n_tracks+1]; reqs: CROP_REQ_DTYPE [n].
        -> (device float tensor [n_samples, sq*fs, sq*fs, 2], limits host array)."""
        t = self.torch
        if track_result.filtered_dev is None:
            raise ValueError("preprocess_segments needs the filtered frames (track_batch(want_filtered=True))")
        refs = np.ascontiguousarray(refs, dtype=REGION_REF_DTYPE)
        reqs = np.ascontiguousarray(reqs, dtype=CROP_REQ_DTYPE)
        offs = np.ascontiguousarray(track_offsets, dtype=np.int32)
        n_tracks = offs.size - 1
        refs_dev = self._to_dev(refs) if refs.size else t.zeros(6, dtype=t.int32, device=self.device)
        offs_dev = t.from_numpy(offs).to(self.device)
        reqs_dev = self._to_dev(reqs) if reqs.size else t.zeros(8, dtype=t.int32, device=self.device)
        limits_dev = t.zeros(max(n_tracks, 1) * 8, dtype=t.int32, device=self.device)
        side = square_width * frame_size
        if out is None:
            out = t.empty((n_samples, side, side, 2), dtype=t.float32, device=self.device)
        self.sync_inputs()
        rc = self.lib.cpx_track_limits_batch_ex(
            self.h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(track_result.filtered_dev.data_ptr()),
            C.c_void_p(track_result.info_dev.data_ptr()), C.c_void_p(refs_dev.data_ptr()),
            C.c_void_p(offs_dev.data_ptr()), n_tracks, C.c_void_p(limits_dev.data_ptr()), int(limits_flags))
        if rc != 0:
            raise CpxError(rc, self._err())
        if reqs.size:  # limits only when there is nothing to crop
            rc = self.lib.cpx_crop_tile(
                self.h, C.c_void_p(frames_dev.data_ptr()), C.c_void_p(track_result.filtered_dev.data_ptr()),
                C.c_void_p(track_result.info_dev.data_ptr()), C.c_void_p(reqs_dev.data_ptr()), int(reqs.size),
                C.c_void_p(limits_dev.data_ptr()), frame_size, square_width, C.c_void_p(out.data_ptr()))
            if rc != 0:
                raise CpxError(rc, self._err())
        self.synchronize()
        limits = limits_dev.cpu().numpy().view(TRACK_LIMITS_DTYPE).reshape(-1)[:n_tracks]
        return out, limits

    def conv_timing(self, enable=None):
        """enable=True/False switches per-launch HIP-event timing of cpx_conv2d; None reports
        {key: (launches, total_ms, flops)} accumulated since it was enabled."""
        from ._lib import CONV_TIMING_DTYPE

        if enable is not None:
            rc = self.lib.cpx_conv_timing_enable(self.h, 1 if enable else 0)
            if rc != 0:
                raise CpxError(rc, self._err())
            return None
        out = np.zeros(32, CONV_TIMING_DTYPE)
        n = C.c_int(0)
        rc = self.lib.cpx_conv_timing_report(self.h, C.c_void_p(out.ctypes.data), 32, C.byref(n))
        if rc != 0:
            raise CpxError(rc, self._err())
        return {int(r["key"]): (int(r["launches"]), float(r["total_ms"]), float(r["flops"])) for r in out[: n.value]}

    def last_kernel_timing(self):
        ms = C.c_float()
        n = C.c_int()
        rc = self.lib.cpx_last_kernel_timing(self.h, C.byref(ms), C.byref(n))
        if rc != 0:
            raise CpxError(rc, self._err())
        return float(ms.value), int(n.value)
