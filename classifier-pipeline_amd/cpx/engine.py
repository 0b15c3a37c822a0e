"""Device engine: owns a cpx handle and runs the track stage for batches of
clips.  PyTorch is used only as the device allocator / copy engine."""

import ctypes as C

import numpy as np

from . import _lib
from ._lib import COMPONENT_DTYPE, FRAME_INFO_DTYPE, FRAME_META_DTYPE, CpxError


def thresholds_for_model(model):
    """(background_thresh, weight_add) -- config/trackingmotionconfig.py:24-59,
    track/cliptrackextractor.py:124-127."""
    if model == "lepton3.5":
        return 50.0, 1.0
    return 20.0, 0.1


class TrackBatchResult:
    """Outputs of one cpx_track_batch call (host copies are made lazily)."""

    def __init__(self, engine, total, cap, comps, info, labels, filtered, background):
        self.engine, self.total, self.cap = engine, total, cap
        self.comps_dev, self.info_dev = comps, info
        self.labels_dev, self.filtered_dev, self.background_dev = labels, filtered, background
        self._info = self._comps = None

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


class TrackEngine:
    def __init__(self, width=160, height=120, model="lepton3", device=0, edge_pixels=1, window=45,
                 max_components=64, max_frames=4096, background_thresh=None, weight_add=None):
        import torch

        self.torch = torch
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("cpx.TrackEngine needs a HIP device (no CPU fallback)")
        bt, wa = thresholds_for_model(model)
        self.cfg = _lib.Config(width, height, edge_pixels, window,
                               float(bt if background_thresh is None else background_thresh),
                               float(wa if weight_add is None else weight_add),
                               max_components, max_frames)
        self.device = torch.device("cuda", device)
        self.h = C.c_void_p()
        rc = self.lib.cpx_create(device, C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise CpxError(rc, "cpx_create")
        self.width, self.height, self.cap = width, height, max_components

    def close(self):
        if self.h:
            self.lib.cpx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self):
        return (self.lib.cpx_last_error(self.h) or b"").decode()

    def synchronize(self):
        rc = self.lib.cpx_synchronize(self.h)
        if rc != 0:
            raise CpxError(rc, self._err())

    def upload_frames(self, frames):
        """uint16 [N,H,W] numpy -> device tensor (stored as int16 bits)."""
        t = self.torch
        a = np.ascontiguousarray(frames, dtype=np.uint16)
        return t.from_numpy(a.view(np.int16)).to(self.device)

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

    def track_batch(self, frames_dev, clip_offsets, meta, want_labels=False, want_filtered=False,
                    want_background=False, outputs=None):
        """frames_dev: device tensor [total,H,W] of uint16 bits; clip_offsets: int32 [B+1]."""
        t = self.torch
        offs = np.ascontiguousarray(clip_offsets, dtype=np.int32)
        B = offs.size - 1
        total = int(offs[-1])
        meta = np.ascontiguousarray(meta, dtype=FRAME_META_DTYPE)
        assert meta.size == total and frames_dev.shape[0] >= total
        P = self.width * self.height
        if outputs is None:
            comps = t.empty(total * self.cap * 8, dtype=t.int32, device=self.device)
            info = t.empty(total * 20, dtype=t.int32, device=self.device)
            labels = t.empty((total, self.height, self.width), dtype=t.int32, device=self.device) if want_labels else None
            filt = t.empty((total, self.height, self.width), dtype=t.float32, device=self.device) if want_filtered else None
            bgo = t.empty((B, self.height, self.width), dtype=t.float32, device=self.device) if want_background else None
        else:
            comps, info, labels, filt, bgo = outputs
        t.cuda.current_stream(self.device).synchronize()  # inputs were produced on torch's stream
        rc = self.lib.cpx_track_batch(
            self.h, C.c_void_p(frames_dev.data_ptr()), offs.ctypes.data_as(C.POINTER(C.c_int32)),
            C.c_void_p(meta.ctypes.data), B, C.c_void_p(comps.data_ptr()), C.c_void_p(info.data_ptr()),
            C.c_void_p(labels.data_ptr() if labels is not None else None),
            C.c_void_p(filt.data_ptr() if filt is not None else None),
            C.c_void_p(bgo.data_ptr() if bgo is not None else None))
        if rc != 0:
            raise CpxError(rc, self._err())
        return TrackBatchResult(self, total, self.cap, comps, info, labels, filt, bgo)

    def last_kernel_timing(self):
        ms = C.c_float()
        n = C.c_int()
        rc = self.lib.cpx_last_kernel_timing(self.h, C.byref(ms), C.byref(n))
        if rc != 0:
            raise CpxError(rc, self._err())
        return float(ms.value), int(n.value)
