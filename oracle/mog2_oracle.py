"""TEST INFRASTRUCTURE ONLY: ctypes front of oracle/mog2_oracle.c (the MOG2 restatement; parity unpinned, see the
header of that file).  Mirrors how the reference's CVBackground drives cv2's object (track/cliptracker.py:561-613)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmog2_oracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE])
        lib = C.CDLL(path)
        lib.mog2_create.restype = C.c_void_p
        lib.mog2_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float]
        lib.mog2_destroy.argtypes = [C.c_void_p]
        lib.mog2_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        lib.mog2_background.argtypes = [C.c_void_p, C.c_void_p]
        lib.mog2_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        lib.mog2_rate.restype = C.c_double
        lib.mog2_rate.argtypes = [C.c_void_p, C.c_double]
        _LIB = lib
    return _LIB


class MOG2:
    """cv2.createBackgroundSubtractorMOG2(history, varThreshold, detectShadows=False) for uint8 [H, W] frames."""

    def __init__(self, width, height, history=1000, var_threshold=16.0):
        self.w, self.h = width, height
        self.m = _lib().mog2_create(width, height, history, var_threshold)

    def apply(self, frame, learning_rate=-1):
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        assert frame.shape == (self.h, self.w)
        mask = np.empty((self.h, self.w), np.uint8)
        _lib().mog2_apply(self.m, frame.ctypes.data, float(learning_rate), mask.ctypes.data)
        return mask

    def getBackgroundImage(self):
        out = np.empty((self.h, self.w), np.uint8)
        _lib().mog2_background(self.m, out.ctypes.data)
        return out

    def state(self):
        n = self.w * self.h
        w = np.empty((n, 5), np.float32)
        v = np.empty((n, 5), np.float32)
        mu = np.empty((n, 5), np.float32)
        k = np.empty(n, np.uint8)
        _lib().mog2_state(self.m, w.ctypes.data, v.ctypes.data, mu.ctypes.data, k.ctypes.data)
        return w, v, mu, k

    def close(self):
        if self.m:
            _lib().mog2_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass
