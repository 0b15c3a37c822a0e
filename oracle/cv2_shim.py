"""TEST INFRASTRUCTURE ONLY -- NumPy restatement of the handful of OpenCV calls
on the reference's track/classify path, injected as ``cv2`` when the reference
(``/root/reference/src``, Python) is imported in the build container to act as
the oracle and to generate ``tests/golden`` fixtures.

OpenCV is a third-party dependency of the reference that is NOT vendored under
/root/reference: ``opencv-python==4.8.0.76`` (pyproject.toml:43) /
``opencv-contrib-python-headless~=4.12.0.88`` (requirements.txt:4).  Each
function below restates the published algorithm of the OpenCV call at the cited
reference call site; the restatement is pinned by the reference's own golden
``tests/clips/possum.txt`` (see tests/test_possum_golden.py): boxes, masses,
frame ranges exact, pixel_variance to 2 dp, tracking_score to 1e-6.

Nothing in the product package imports this file.
"""

import numpy as np
from scipy import ndimage

# ---- constants the reference touches at import / call time -----------------
INTER_NEAREST = 0
INTER_LINEAR = 1
INTER_CUBIC = 2
INTER_AREA = 3
THRESH_BINARY = 0
THRESH_OTSU = 8
MORPH_ERODE = 0
MORPH_DILATE = 1
MORPH_OPEN = 2
MORPH_CLOSE = 3
RETR_EXTERNAL = 0
RETR_LIST = 1
CHAIN_APPROX_SIMPLE = 2
CHAIN_APPROX_NONE = 1
NORM_MINMAX = 32
INPAINT_TELEA = 1
COLOR_HSV2BGR = 54
COLOR_BGR2GRAY = 6
CC_STAT_LEFT, CC_STAT_TOP, CC_STAT_WIDTH, CC_STAT_HEIGHT, CC_STAT_AREA = range(5)
CV_32S = 4
__version__ = "4.8.0-shim"


def GaussianBlur(src, ksize, sigmaX, *args, **kwargs):
    """8-bit 5x5 Gaussian, sigma=0 (imageprocessing.py:242).

    OpenCV's 8U path uses the fixed-point binomial kernel [1,4,6,4,1]/16 per
    axis with BORDER_REFLECT_101 and a single rounding: (S + 128) >> 8.
    """
    src = np.asarray(src)
    if tuple(ksize) != (5, 5) or sigmaX != 0:
        raise NotImplementedError("shim: only GaussianBlur((5,5), 0)")
    if src.dtype != np.uint8:
        raise NotImplementedError("shim: only 8-bit GaussianBlur")
    k = np.array([1, 4, 6, 4, 1], dtype=np.int32)
    p = np.pad(src.astype(np.int32), 2, mode="reflect")
    h, w = src.shape
    acc = np.zeros((h + 4, w), dtype=np.int32)
    for j in range(5):
        acc += k[j] * p[:, j : j + w]
    out = np.zeros((h, w), dtype=np.int32)
    for i in range(5):
        out += k[i] * acc[i : i + h, :]
    return ((out + 128) >> 8).astype(np.uint8)


def threshold(src, thresh, maxval, type):
    """THRESH_BINARY on 8U (imageprocessing.py:246): dst = src > floor(thresh)."""
    src = np.asarray(src)
    if type != THRESH_BINARY:
        raise NotImplementedError("shim: only THRESH_BINARY")
    if src.dtype != np.uint8:
        raise NotImplementedError("shim: only 8-bit threshold")
    ithresh = int(np.floor(thresh))
    dst = np.where(src.astype(np.int32) > ithresh, np.uint8(maxval), np.uint8(0))
    return float(thresh), dst.astype(np.uint8)


def morphologyEx(src, op, kernel, *args, **kwargs):
    """MORPH_CLOSE / MORPH_OPEN with the reference's *tuple* kernel (imageprocessing.py:189,247).

    The binding turns the tuple (5, 5) or (15, 15) into a 2x1 CV_64F Mat: an all-set structuring element 1 px wide,
    2 px tall, anchor (0,1): dilate D[y] = max(I[y], I[y-1]); erode E[y] = min(I[y], I[y-1]); rows outside
    the image are ignored by both.  close = erode(dilate(I)) (pinned by the golden, F3), open = dilate(erode(I)) (the
    IR path, detect_objects_ir; same element, not covered by a golden).
    """
    src = np.asarray(src)
    if op not in (MORPH_CLOSE, MORPH_OPEN):
        raise NotImplementedError("shim: only MORPH_CLOSE / MORPH_OPEN")
    if not isinstance(kernel, tuple) or len(kernel) != 2:
        raise NotImplementedError("shim: only the tuple-kernel form")

    def dilate(a):
        d = a.copy()
        d[1:] = np.maximum(a[1:], a[:-1])
        return d

    def erode(a):
        e = a.copy()
        e[1:] = np.minimum(a[1:], a[:-1])
        return e

    return erode(dilate(src)) if op == MORPH_CLOSE else dilate(erode(src))


def connectedComponentsWithStats(image, *args, **kwargs):
    """8-connectivity labelling + stats (imageprocessing.py:248).

    stats[i] = (left, top, width, height, area) int32; centroids[i] = (mean x,
    mean y) float64; row 0 is the background.  Label numbering follows OpenCV's
    default 8-conn algorithm (2x2-block raster scan, union keeps the smaller
    provisional label, consecutive renumbering): components are numbered by the
    block-raster position of their first 2x2 block (SURVEY a7').
    """
    img = np.asarray(image)
    fg = img > 0
    h, w = fg.shape
    lab, n = ndimage.label(fg, structure=np.ones((3, 3), dtype=np.int32))
    labels = np.zeros((h, w), dtype=np.int32)
    stats = np.zeros((n + 1, 5), dtype=np.int32)
    cents = np.zeros((n + 1, 2), dtype=np.float64)
    ys, xs = np.nonzero(fg)
    bw = (w + 1) // 2
    if n > 0:
        key = (ys >> 1) * bw + (xs >> 1)
        comp = lab[ys, xs]
        first = np.full(n + 1, np.iinfo(np.int64).max, dtype=np.int64)
        np.minimum.at(first, comp, key)
        order = np.argsort(first[1:], kind="stable")  # old label-1 sorted by key
        remap = np.zeros(n + 1, dtype=np.int32)
        remap[order + 1] = np.arange(1, n + 1, dtype=np.int32)
        labels[ys, xs] = remap[comp]
        newc = remap[comp]
        big = np.iinfo(np.int64).max
        mnx = np.full(n + 1, big, np.int64)
        mny = np.full(n + 1, big, np.int64)
        mxx = np.full(n + 1, -1, np.int64)
        mxy = np.full(n + 1, -1, np.int64)
        np.minimum.at(mnx, newc, xs)
        np.minimum.at(mny, newc, ys)
        np.maximum.at(mxx, newc, xs)
        np.maximum.at(mxy, newc, ys)
        area = np.bincount(newc, minlength=n + 1)
        sx = np.bincount(newc, weights=xs, minlength=n + 1)  # exact: integer sums far below 2^53
        sy = np.bincount(newc, weights=ys, minlength=n + 1)
        for i in range(1, n + 1):
            stats[i] = (mnx[i], mny[i], mxx[i] - mnx[i] + 1, mxy[i] - mny[i] + 1, area[i])
            cents[i] = (sx[i] / area[i], sy[i] / area[i])
    bys, bxs = np.nonzero(~fg)
    if bys.size:
        stats[0] = (
            bxs.min(),
            bys.min(),
            bxs.max() - bxs.min() + 1,
            bys.max() - bys.min() + 1,
            bys.size,
        )
        cents[0] = (bxs.sum() / bxs.size, bys.sum() / bys.size)
    return n + 1, labels, stats, cents


_NLM_LUT = None


def _nlm_lut():
    global _NLM_LUT
    if _NLM_LUT is None:
        h = 3.0
        tsize = 49
        fixed_point_mult = (2**31 - 1) // (21 * 21 * 255)
        shift = 6  # smallest s with 2**s >= 49
        mult = (1 << shift) / tsize
        max_dist = 255 * 255
        n = int(max_dist / mult + 1)
        a = np.arange(n, dtype=np.float64)
        wv = np.exp(-(a * mult) / (h * h))
        w = np.rint(fixed_point_mult * wv)
        w[w < 0.001 * fixed_point_mult] = 0
        _NLM_LUT = (w.astype(np.int64), shift, fixed_point_mult)
    return _NLM_LUT


def fastNlMeansDenoising(src, dst=None, h=3, templateWindowSize=7, searchWindowSize=21):
    """Integer non-local means, defaults h=3 / 7 / 21 (cliptracker.py:117).

    SURVEY Appendix A.6: reflect-101 border of 13; per offset in [-10,10]^2 the
    7x7 summed squared difference, weight = LUT[dist >> 6], fixed-point
    accumulation, result (est + wsum//2) // wsum.
    """
    src = np.asarray(src)
    if src.dtype != np.uint8 or src.ndim != 2:
        raise NotImplementedError("shim: NLM only for 2-D uint8")
    if h != 3 or templateWindowSize != 7 or searchWindowSize != 21:
        raise NotImplementedError("shim: NLM defaults only")
    lut, shift, _ = _nlm_lut()
    t, s = 3, 10
    b = t + s
    H, W = src.shape
    ext = np.pad(src.astype(np.int64), b, mode="reflect")
    est = np.zeros((H, W), dtype=np.int64)
    wsum = np.zeros((H, W), dtype=np.int64)
    # region of centres including the template halo
    base = ext[b - t : b + H + t, b - t : b + W + t]
    for dy in range(-s, s + 1):
        for dx in range(-s, s + 1):
            sh = ext[b - t + dy : b + H + t + dy, b - t + dx : b + W + t + dx]
            d2 = (base - sh) ** 2
            c = np.cumsum(np.cumsum(d2, axis=0), axis=1)
            c = np.pad(c, ((1, 0), (1, 0)))
            k = 2 * t + 1
            dist = c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]
            wgt = lut[dist >> shift]
            est += wgt * ext[b + dy : b + dy + H, b + dx : b + dx + W]
            wsum += wgt
    out = (est + wsum // 2) // wsum
    return np.clip(out, 0, 255).astype(np.uint8)


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    """float32 bilinear / nearest resize (imageprocessing.py:78).

    INTER_LINEAR: source coordinate (d + 0.5) * (src/dst) - 0.5, floor +
    fractional weight in float32, indices clamped to the edge; horizontal pass
    then vertical pass, all float32.  INTER_NEAREST: min(floor(d*src/dst), src-1).
    Not pinned by the golden.
    """
    src = np.asarray(src)
    if src.ndim == 3:
        chans = [resize(src[:, :, c], dsize, interpolation=interpolation) for c in range(src.shape[2])]
        return np.stack(chans, axis=2)
    dw, dh = int(dsize[0]), int(dsize[1])
    sh, sw = src.shape
    if interpolation == INTER_NEAREST:
        xs = np.minimum(np.floor(np.arange(dw) * (sw / dw)).astype(np.int64), sw - 1)
        ys = np.minimum(np.floor(np.arange(dh) * (sh / dh)).astype(np.int64), sh - 1)
        return src[np.ix_(ys, xs)].copy()
    if interpolation == INTER_AREA:
        # integer-ratio area filter (resizeAreaFast_ in OpenCV's resize.cpp): the f x f block mean; 8-bit at f = 2 is
        # (sum + 2) >> 2 (ResizeAreaFastVec_SIMD_8u), otherwise saturate_cast<uchar>(sum * (1.f / (f * f))): round to
        # nearest, ties to even.  Used by the IR tracker's `scale` (irtrackextractor.py:445-451).  Not pinned (no cv2 here).
        if src.dtype != np.uint8 or sw % dw or sh % dh or sw // dw != sh // dh:
            raise NotImplementedError("shim: INTER_AREA for uint8 and one integer ratio on both axes only")
        f = sw // dw
        s = src.reshape(dh, f, dw, f).astype(np.int64).sum(axis=(1, 3))
        if f == 2:
            return ((s + 2) >> 2).astype(np.uint8)
        return np.clip(np.rint(s.astype(np.float32) * np.float32(1.0 / (f * f))), 0, 255).astype(np.uint8)
    if interpolation != INTER_LINEAR:
        raise NotImplementedError("shim: resize INTER_LINEAR / INTER_NEAREST / INTER_AREA (integer ratio) only")
    s32 = src.astype(np.float32)

    def coords(dn, sn):
        scale = sn / dn
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        a = (f - i0.astype(np.float32)).astype(np.float32)
        lo = i0 < 0
        a[lo] = 0
        i0[lo] = 0
        hi = i0 >= sn - 1
        a[hi] = 0
        i0[hi] = sn - 1
        i1 = np.minimum(i0 + 1, sn - 1)
        return i0, i1, a

    x0, x1, ax = coords(dw, sw)
    y0, y1, ay = coords(dh, sh)
    one = np.float32(1.0)
    hor = s32[:, x0] * (one - ax)[None, :] + s32[:, x1] * ax[None, :]
    hor = hor.astype(np.float32)
    out = hor[y0, :] * (one - ay)[:, None] + hor[y1, :] * ay[:, None]
    return out.astype(np.float32)


class KalmanFilter:
    """cv2.KalmanFilter(4, 2) in float32 (kalman.py:10).

    OpenCV evaluates every matrix product of predict()/correct() with its gemm,
    which for CV_32F accumulates in double and rounds to float once per output
    element (alpha*A*B + beta*C in one rounding).  The 2x2 solve (DECOMP_SVD in
    OpenCV) is defined here as Cramer's rule in double rounded to float; the
    filter output only feeds int()-truncated blank-frame boxes and a distance
    gate, and this definition reproduces the possum.txt golden.
    """

    def __init__(self, dynamParams, measureParams, controlParams=0, type=5):
        dp, mp = dynamParams, measureParams
        f = np.float32
        self.statePre = np.zeros((dp, 1), f)
        self.statePost = np.zeros((dp, 1), f)
        self.transitionMatrix = np.eye(dp, dtype=f)
        self.processNoiseCov = np.eye(dp, dtype=f)
        self.measurementMatrix = np.zeros((mp, dp), f)
        self.measurementNoiseCov = np.eye(mp, dtype=f)
        self.errorCovPre = np.zeros((dp, dp), f)
        self.errorCovPost = np.zeros((dp, dp), f)
        self.gain = np.zeros((dp, mp), f)

    @staticmethod
    def _gemm(a, b, alpha=1.0, c=None):
        """float32( alpha * sum_k a_ik b_kj + c_ij ), accumulated in double in k order."""
        a64 = np.asarray(a, dtype=np.float32).astype(np.float64)
        b64 = np.asarray(b, dtype=np.float32).astype(np.float64)
        n, kk = a64.shape
        m = b64.shape[1]
        out = np.zeros((n, m), dtype=np.float64)
        for k in range(kk):
            out += a64[:, k : k + 1] * b64[k : k + 1, :]
        out = alpha * out
        if c is not None:
            out = out + np.asarray(c, dtype=np.float32).astype(np.float64)
        return out.astype(np.float32)

    def predict(self, control=None):
        A = self.transitionMatrix
        self.statePre = self._gemm(A, self.statePost)
        temp1 = self._gemm(A, self.errorCovPost)
        self.errorCovPre = self._gemm(temp1, A.T, 1.0, self.processNoiseCov)
        self.statePost = self.statePre.copy()
        self.errorCovPost = self.errorCovPre.copy()
        return self.statePre.copy()

    def correct(self, measurement):
        f = np.float32
        H = self.measurementMatrix
        z = np.asarray(measurement, dtype=f).reshape(-1, 1)
        temp2 = self._gemm(H, self.errorCovPre)
        temp3 = self._gemm(temp2, H.T, 1.0, self.measurementNoiseCov)
        if temp3.shape != (2, 2):
            raise NotImplementedError("shim: KalmanFilter with 2 measurements only")
        a, b, c, d = (np.float64(temp3[0, 0]), np.float64(temp3[0, 1]),
                      np.float64(temp3[1, 0]), np.float64(temp3[1, 1]))
        det = a * d - b * c
        r0 = temp2[0].astype(np.float64)
        r1 = temp2[1].astype(np.float64)
        temp4 = np.stack([(d * r0 - b * r1) / det, (a * r1 - c * r0) / det]).astype(f)
        self.gain = temp4.T.copy()
        temp5 = self._gemm(H, self.statePre, -1.0, z)
        self.statePost = self._gemm(self.gain, temp5, 1.0, self.statePre)
        self.errorCovPost = self._gemm(self.gain, temp2, -1.0, self.errorCovPre)
        return self.statePost.copy()


CHAIN_APPROX_TC89_L1 = 3
CHAIN_APPROX_TC89_KCOS = 4
_CODE_DELTAS = ((1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1))  # code -> (dx, dy)
_ABS_DIFF = (1, 2, 3, 4, 3, 2, 1, 0, 1, 2, 3, 4, 3, 2, 1)


def _trace_border(img, x0, y0):
    """Suzuki-Abe border following of an outer border starting at (x0, y0) of the zero-padded int8
    image (OpenCV contours: icvFetchContour with nbd = 2).  Marks the border pixels (2, or -126 where
    the border has background on its right) and returns the Freeman chain codes."""
    def at(x, y):
        return img[y, x]

    s_end = s = 4
    while True:
        s = (s - 1) & 7
        x1, y1 = x0 + _CODE_DELTAS[s][0], y0 + _CODE_DELTAS[s][1]
        if at(x1, y1) != 0 or s == s_end:
            break
    if s == s_end:  # single pixel
        img[y0, x0] = -126
        return []
    chain = []
    x3, y3 = x0, y0
    while True:
        s_end = s
        while s < 15:
            s += 1
            d = _CODE_DELTAS[s & 7]
            x4, y4 = x3 + d[0], y3 + d[1]
            if at(x4, y4) != 0:
                break
        s &= 7
        if ((s - 1) & 0xFFFFFFFF) < s_end:  # unsigned compare as in the C source
            img[y3, x3] = -126
        elif img[y3, x3] == 1:
            img[y3, x3] = 2
        chain.append(s)
        if (x4, y4) == (x0, y0) and (x3, y3) == (x1, y1):
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return chain


def _approx_chain(chain, origin, method):
    """Chain codes -> polygon points (CHAIN_APPROX_NONE / SIMPLE / TC89_L1), following OpenCV's
    Teh-Chin implementation pass by pass (1-curvature, support regions, non-maxima suppression,
    1-length support removal, pair cleaning)."""
    n = len(chain)
    if n == 0:
        return [origin]
    pts, svals = [], []
    x, y = origin
    for i in range(n):
        prev_code = chain[i - 1]
        pts.append((x, y))
        svals.append(_ABS_DIFF[chain[i] - prev_code + 7])
        x += _CODE_DELTAS[chain[i]][0]
        y += _CODE_DELTAS[chain[i]][1]
    if method == CHAIN_APPROX_NONE:
        return pts
    if method == CHAIN_APPROX_SIMPLE:
        return [p for p, sv in zip(pts, svals) if sv != 0]
    if method != CHAIN_APPROX_TC89_L1:
        raise NotImplementedError("contour approximation %r" % method)
    length = n
    s = list(svals) + [0]
    k = [0] * (n + 1)
    nxt = [None] * (n + 1)           # linked list over array indices; -1 is the list head `temp`
    pts = pts + [None]
    head = {"next": None}
    order = [i for i in range(n) if s[i] != 0]
    assert order, "closed chain without a corner"

    def link(seq):
        head["next"] = seq[0] if seq else None
        for a, b in zip(seq, seq[1:]):
            nxt[a] = b
        if seq:
            nxt[seq[-1]] = None

    link(order)
    # Pass 1: support region of every remaining point
    cur = head["next"]
    while cur is not None:
        i = cur
        x0, y0 = pts[i]
        l = 0
        d_num = 0
        kk = 1
        while True:
            assert kk <= length
            i1 = i - kk
            i1 += length if i1 < 0 else 0
            i2 = i + kk
            i2 -= length if i2 >= length else 0
            dx = pts[i2][0] - pts[i1][0]
            dy = pts[i2][1] - pts[i1][1]
            lk = dx * dx + dy * dy
            dk_num = (x0 - pts[i1][0]) * dy - (y0 - pts[i1][1]) * dx
            d = float(np.float32(float(d_num) * lk - float(dk_num) * l))
            if kk > 1 and (l >= lk or (d_num > 0 and d <= 0) or (d_num < 0 and d >= 0)):
                break
            d_num = dk_num
            l = lk
            kk += 1
        k[cur] = kk - 1
        cur = nxt[cur]
    # Pass 2: non-maxima suppression
    prev = None
    cur = head["next"]

    def unlink(prev, cur):
        if prev is None:
            head["next"] = nxt[cur]
        else:
            nxt[prev] = nxt[cur]

    while cur is not None:
        k2 = k[cur] >> 1
        sv = s[cur]
        i = cur
        j = 1
        while j <= k2:
            i2 = i - j
            i2 += length if i2 < 0 else 0
            if s[i2] > sv:
                break
            i2 = i + j
            i2 -= length if i2 >= length else 0
            if s[i2] > sv:
                break
            j += 1
        if j <= k2:
            unlink(prev, cur)
            s[cur] = 0
        else:
            prev = cur
        cur = nxt[cur]
    # Pass 3: non-dominant points with a 1-length support region
    prev = None
    cur = head["next"]
    assert cur is not None
    while cur is not None:
        if k[cur] == 1:
            sv = s[cur]
            i = cur
            i1 = i - 1
            i1 += length if i1 < 0 else 0
            i2 = i + 1
            i2 -= length if i2 >= length else 0
            if sv <= s[i1] or sv <= s[i2]:
                unlink(prev, cur)
                s[cur] = 0
            else:
                prev = cur
        else:
            prev = cur
        cur = nxt[cur]
    # Pass 4: clean the remaining couples of neighbouring points
    assert head["next"] is not None
    all_survived = False
    if s[0] != 0 and s[length - 1] != 0:  # a run of points wraps around the array end
        i1 = 1
        while i1 < length and s[i1] != 0:
            s[i1 - 1] = 0
            i1 += 1
        if i1 == length:
            all_survived = True
        else:
            i1 -= 1
            i2 = length - 2
            while i2 > 0 and s[i2] != 0:
                nxt[i2] = None
                s[i2 + 1] = 0
                i2 -= 1
            i2 += 1
            if i1 == 0 and i2 == length - 1:  # only two points
                i1 = nxt[0]
                pts[length] = pts[0]
                s[length] = s[0]
                k[length] = k[0]
                nxt[length] = None
                nxt[length - 1] = length
            head["next"] = i1
    if not all_survived:
        cur = head["next"]
        first = prev = None           # None stands for the list head
        count = 1

        def set_next(node, val):
            if node is None:
                head["next"] = val
            else:
                nxt[node] = val

        def get_next(node):
            return head["next"] if node is None else nxt[node]

        while cur is not None:
            if nxt[cur] is None or nxt[cur] - cur != 1:
                if count >= 2:
                    if count == 2:
                        s1 = 0 if prev is None else s[prev]
                        s2 = s[cur]
                        k1 = 0 if prev is None else k[prev]
                        if s1 > s2 or (s1 == s2 and k1 <= k[cur]):
                            set_next(prev, nxt[cur])      # remove the second
                        else:
                            set_next(first, cur)          # remove the first
                    else:
                        set_next(get_next(first), cur)
                first = cur
                count = 1
            else:
                count += 1
            prev = cur
            cur = nxt[cur]
    out = []
    cur = head["next"]
    assert cur is not None
    while cur is not None:
        out.append(pts[cur])
        cur = nxt[cur]
    return out


def findContours(image, mode, method, *args, **kwargs):
    """cv2.findContours(u8, RETR_EXTERNAL, method): raster scan for outer-border starts (pixel 1 with
    background on its left) that are not nested inside an already traced outline, Suzuki-Abe border
    following, then the requested chain approximation.  Returns ([int32 [n,1,2] ...], None)."""
    if mode != RETR_EXTERNAL:
        raise NotImplementedError("only RETR_EXTERNAL is on the path (classify/thumbnail.py:91)")
    src = np.asarray(image)
    H, W = src.shape
    img = np.zeros((H + 2, W + 2), np.int8)
    img[1:-1, 1:-1] = src != 0
    contours = []
    for y in range(1, H + 1):
        prev = 0
        lnbd_x = 0
        row = img[y]
        for x in range(1, W + 1):
            p = int(row[x])
            if p == prev:
                continue
            if prev == 0 and p == 1 and not img[y, lnbd_x] > 0:
                chain = _trace_border(img, x, y)
                pts = _approx_chain(chain, (x - 1, y - 1), method)
                contours.append(np.asarray(pts, np.int32).reshape(-1, 1, 2))
                p = int(row[x])
            prev = p
            if prev & -2:
                lnbd_x = x
    return tuple(contours), None


def contourArea(contour):
    return float(len(contour))


def __getattr__(name):  # pragma: no cover - any other cv2 symbol is off-path
    if name.isupper():
        return 0
    raise AttributeError("cv2 shim has no attribute %r (off the hot path)" % name)
