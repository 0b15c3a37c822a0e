"""CPTV v2 container decoder (host side, row a1 / f1 of SURVEY.md §8).

Replaces the un-vendored Rust reader ``cptv_rs_python_bindings.CptvReader``
(python-cptv==0.0.8) at the call sites reference
``src/track/cliptrackextractor.py:108-129,160-162`` and
``src/classify/clipclassifier.py:460-469``.

Format (restated from the published CPTV v2 layout, verified on the two
fixture clips): gzip stream -> magic ``CPTV`` + version byte 2 -> section
``H`` (u8 field count; each field = u8 length, u8 code, data) -> repeated
``F`` sections (same field encoding; field ``f`` = payload byte count, ``w`` =
bits per delta) each followed by the payload: first pixel delta as i32 LE, then
W*H-1 deltas of ``w`` bits, MSB first, two's complement.  The running sum of the
deltas, laid out in snake order (odd rows right-to-left), is the inter-frame
difference; frame = previous frame + difference (previous = 0 for the first).

Reader contract used by the tracker: ``get_header()`` -> object with
``x_resolution, y_resolution, model, brand, timestamp`` (us);
``next_frame()`` -> object with ``pix uint16[H,W]``, ``time_on`` /
``last_ffc_time`` as *int* milliseconds, ``background_frame``,
``temp_c``, ``last_ffc_temp_c``; ``None`` at end of file.
"""

import struct
import zlib

import numpy as np


class CptvHeader:
    def __init__(self):
        self.timestamp = None
        self.x_resolution = 0
        self.y_resolution = 0
        self.compression = None
        self.device_name = None
        self.model = None
        self.brand = None
        self.fps = None
        self.device_id = None
        self.preview_secs = None
        self.motion_config = None
        self.latitude = None
        self.longitude = None
        self.loc_timestamp = None
        self.altitude = None
        self.accuracy = None
        self.firmware = None
        self.serial = None
        self.has_background_frame = False


class CptvFrame:
    __slots__ = (
        "pix",
        "time_on",
        "last_ffc_time",
        "temp_c",
        "last_ffc_temp_c",
        "background_frame",
    )

    def __init__(self, pix, time_on, last_ffc_time, temp_c, last_ffc_temp_c, background_frame):
        self.pix = pix
        self.time_on = time_on
        self.last_ffc_time = last_ffc_time
        self.temp_c = temp_c
        self.last_ffc_temp_c = last_ffc_temp_c
        self.background_frame = background_frame


def _read_fields(buf, pos):
    count = buf[pos]
    pos += 1
    fields = {}
    for _ in range(count):
        ln = buf[pos]
        code = chr(buf[pos + 1])
        fields[code] = bytes(buf[pos + 2 : pos + 2 + ln])
        pos += 2 + ln
    return fields, pos


def _u32(b):
    return struct.unpack("<I", b)[0]


def _f32(b):
    return struct.unpack("<f", b)[0]


def _header_from_fields(fields):
    h = CptvHeader()
    if "T" in fields:
        h.timestamp = struct.unpack("<Q", fields["T"])[0]
    h.x_resolution = _u32(fields["X"])
    h.y_resolution = _u32(fields["Y"])
    h.compression = fields.get("C", b"\0")[0]
    for code, name in (("D", "device_name"), ("E", "model"), ("B", "brand"),
                       ("V", "firmware"), ("M", "motion_config")):
        if code in fields:
            setattr(h, name, fields[code].decode("utf-8", "replace"))
    if "Z" in fields:
        h.fps = fields["Z"][0]
    if "I" in fields:
        h.device_id = _u32(fields["I"])
    if "N" in fields:
        h.serial = _u32(fields["N"])
    if "P" in fields:
        h.preview_secs = fields["P"][0]
    if "L" in fields:
        h.latitude = _f32(fields["L"])
    if "O" in fields:
        h.longitude = _f32(fields["O"])
    if "S" in fields:
        h.loc_timestamp = struct.unpack("<Q", fields["S"])[0]
    if "A" in fields:
        h.altitude = _f32(fields["A"])
    if "U" in fields:
        h.accuracy = _f32(fields["U"])
    if "g" in fields:
        h.has_background_frame = fields["g"][0] != 0
    return h


def _unpack_deltas(payload, n, width):
    """n signed `width`-bit big-endian (MSB first) values from payload -> int64."""
    if width == 8:
        return np.frombuffer(payload, dtype=np.int8, count=n).astype(np.int64)
    if width == 16:
        return np.frombuffer(payload, dtype=">i2", count=n).astype(np.int64)
    if width == 32:
        return np.frombuffer(payload, dtype=">i4", count=n).astype(np.int64)
    bits = np.unpackbits(np.frombuffer(payload, dtype=np.uint8), count=n * width)
    bits = bits.reshape(n, width).astype(np.int64)
    weights = 1 << np.arange(width - 1, -1, -1, dtype=np.int64)
    vals = bits @ weights
    sign = bits[:, 0] != 0
    vals[sign] -= 1 << width
    return vals


class CptvReader:
    """Sequential CPTV v2 reader with the ``cptv_rs_python_bindings`` surface."""

    def __init__(self, path):
        """path: a file name, or the recording itself as bytes (run_files_bulk(blobs=...) retries of in-memory
        recordings the device decoder refused)."""
        if isinstance(path, (bytes, bytearray, memoryview)):
            raw = bytes(path)
            path = "<%d bytes in memory>" % len(raw)
        else:
            with open(str(path), "rb") as f:
                raw = f.read()
        if raw[:2] == b"\x1f\x8b":
            # whole-buffer inflate (zlib releases the GIL: decode_clips_on_device runs readers in threads);
            # concatenated gzip members are followed like gzip.open does
            out = []
            while raw:
                d = zlib.decompressobj(47)
                try:
                    out.append(d.decompress(raw))
                except zlib.error as e:
                    raise ValueError("corrupt CPTV gzip stream: %s (%s)" % (path, e))
                if not d.eof:
                    out.append(d.flush())
                    break  # truncated stream: the section parser reports it
                raw = d.unused_data
            self._buf = b"".join(out)
        else:
            self._buf = raw
        buf = self._buf
        if buf[:4] != b"CPTV":
            raise ValueError("not a CPTV file: %s" % path)
        if buf[4] != 2:
            raise ValueError("unsupported CPTV version %d" % buf[4])
        if buf[5:6] != b"H":
            raise ValueError("CPTV header section missing")
        fields, self._pos = _read_fields(buf, 6)
        h = _header_from_fields(fields)
        self._header = h
        self._w = h.x_resolution
        self._h = h.y_resolution
        self._prev = np.zeros((self._h, self._w), dtype=np.int64)
        # snake order: odd rows are stored right-to-left
        idx = np.arange(self._w * self._h).reshape(self._h, self._w)
        idx[1::2] = idx[1::2, ::-1].copy()
        self._snake = idx

    def get_header(self):
        return self._header

    def _next_section(self):
        """Parse the next frame section's fields without touching its payload:
        (fields, payload offset, payload bytes, bits per delta) or None at the end."""
        buf = self._buf
        pos = self._pos
        if pos >= len(buf):
            return None
        if buf[pos : pos + 1] != b"F":
            raise ValueError("expected frame section at %d" % pos)
        fields, pos = _read_fields(buf, pos + 1)
        width = fields["w"][0]
        nbytes = _u32(fields["f"])
        if pos + nbytes > len(buf):
            raise ValueError("truncated CPTV frame")
        need = 4 + ((self._w * self._h - 1) * width + 7) // 8
        if nbytes < need or not 1 <= width <= 32:
            raise ValueError("malformed CPTV frame section at %d" % pos)
        self._pos = pos + nbytes
        return fields, pos, nbytes, width

    def scan(self):
        """Index the remaining frame sections for the device decoder
        (``decode_clips_on_device``): per frame the metadata-only CptvFrame
        (``pix`` None), payload offset into ``inflated`` and delta width."""
        frames, offsets, widths = [], [], []
        while True:
            sec = self._next_section()
            if sec is None:
                break
            fields, pos, _, width = sec
            frames.append(self._frame_from_fields(fields, None))
            offsets.append(pos)
            widths.append(width)
        return frames, np.asarray(offsets, np.int64), np.asarray(widths, np.int32)

    @property
    def inflated(self):
        return self._buf

    @staticmethod
    def _frame_from_fields(fields, pix):
        time_on = _u32(fields["t"]) if "t" in fields else None
        last_ffc = _u32(fields["c"]) if "c" in fields else None
        temp_c = _f32(fields["a"]) if "a" in fields else 0.0
        ffc_temp = _f32(fields["b"]) if "b" in fields else 0.0
        bg = ("g" in fields) and fields["g"][0] != 0
        return CptvFrame(pix, time_on, last_ffc, temp_c, ffc_temp, bg)

    def next_frame(self):
        sec = self._next_section()
        if sec is None:
            return None
        fields, pos, nbytes, width = sec
        payload = self._buf[pos : pos + nbytes]
        n = self._w * self._h
        deltas = np.empty(n, dtype=np.int64)
        deltas[0] = struct.unpack("<i", payload[:4])[0]
        deltas[1:] = _unpack_deltas(payload[4:], n - 1, width)
        diff = np.cumsum(deltas)[self._snake]
        self._prev = self._prev + diff
        pix = self._prev.astype(np.uint16)
        return self._frame_from_fields(fields, pix)

    def read_all(self):
        frames = []
        while True:
            f = self.next_frame()
            if f is None:
                return frames
            frames.append(f)


def encode_cptv(frames, time_on=None, last_ffc=None, model=b"lepton3", background_first=False, timestamp=1600000000000000,
                level=1, device_name=b"synthetic"):
    """A CPTV v2 file (bytes) of `frames` uint16 [N, H, W]: the inverse of CptvReader, for synthetic recordings (bench.py
    from_files, tests).  Every frame is packed at the narrowest of 8 / 16 / 32 bits per delta that holds it (the
    byte-aligned widths: packing is then one astype per frame); gzip level 1 by default."""
    import gzip
    import io

    frames = np.ascontiguousarray(frames, dtype=np.uint16)
    N, H, W = frames.shape

    def field(code, data):
        return bytes([len(data)]) + code + data

    hdr = [field(b"T", struct.pack("<Q", int(timestamp))), field(b"X", struct.pack("<I", W)),
           field(b"Y", struct.pack("<I", H)), field(b"C", b"\x01"), field(b"D", device_name), field(b"Z", b"\x09")]
    if model:
        hdr.append(field(b"E", model if isinstance(model, bytes) else str(model).encode()))
    if background_first:
        hdr.append(field(b"g", b"\x01"))
    out = [b"CPTV\x02H", bytes([len(hdr)]), b"".join(hdr)]
    snake = frames.astype(np.int32)
    snake[:, 1::2] = snake[:, 1::2, ::-1]
    snake = snake.reshape(N, H * W)
    diff = np.diff(snake, axis=0, prepend=np.zeros((1, H * W), np.int32))   # inter-frame difference, scan order
    deltas = np.diff(diff, axis=1, prepend=np.zeros((N, 1), np.int32))      # its running differences
    lo, hi = deltas[:, 1:].min(axis=1), deltas[:, 1:].max(axis=1)
    for i in range(N):
        if lo[i] >= -128 and hi[i] <= 127:
            w, packed = 8, deltas[i, 1:].astype(np.int8).tobytes()
        elif lo[i] >= -32768 and hi[i] <= 32767:
            w, packed = 16, deltas[i, 1:].astype(">i2").tobytes()
        else:
            w, packed = 32, deltas[i, 1:].astype(">i4").tobytes()
        payload = struct.pack("<i", int(deltas[i, 0])) + packed
        fl = [field(b"w", bytes([w])), field(b"f", struct.pack("<I", len(payload)))]
        if time_on is not None:
            fl += [field(b"t", struct.pack("<I", int(time_on[i]))), field(b"c", struct.pack("<I", int(last_ffc[i])))]
        if background_first and i == 0:
            fl.append(field(b"g", b"\x01"))
        out += [b"F", bytes([len(fl)]), b"".join(fl), payload]
    buf = io.BytesIO()
    with gzip.GzipFile(fileobj=buf, mode="wb", compresslevel=level, mtime=0) as fh:
        fh.write(b"".join(out))
    return buf.getvalue()


def decode_clips_on_device(engine, paths, workers=8):
    """Decode whole CPTV files with ``cpx_cptv_unpack`` (include/cpx.h): the host
    inflates the gzip stream and indexes the sections, the GPU unpacks the
    frames.  Returns (headers, per-clip lists of metadata-only CptvFrame,
    frames_dev uint16 [total, H, W] as a torch int16 tensor, clip_offsets)."""
    import torch

    from concurrent.futures import ThreadPoolExecutor

    def index(path):
        reader = CptvReader(path)
        return (reader,) + reader.scan()

    paths = list(paths)
    if len(paths) > 1:
        with ThreadPoolExecutor(max_workers=min(len(paths), workers)) as pool:
            indexed = list(pool.map(index, paths))
    else:
        indexed = [index(p) for p in paths]
    headers, metas, chunks, offs, widths, clip_offsets = [], [], [], [], [], [0]
    base = 0
    for path, (reader, frames, o, w) in zip(paths, indexed):
        h = reader.get_header()
        if (h.x_resolution, h.y_resolution) != (engine.width, engine.height):
            raise ValueError("%s is %dx%d, the engine was created for %dx%d"
                             % (path, h.x_resolution, h.y_resolution, engine.width, engine.height))
        headers.append(h)
        metas.append(frames)
        chunks.append(np.frombuffer(reader.inflated, np.uint8))
        offs.append(o + base)
        widths.append(w)
        base += len(reader.inflated)
        clip_offsets.append(clip_offsets[-1] + len(frames))
    total = clip_offsets[-1]
    if total == 0:
        raise ValueError("no frames in %s" % (list(paths),))
    payload = np.concatenate(chunks + [np.zeros(16, np.uint8)])
    frames_dev = engine.cptv_unpack(payload, np.concatenate(offs), np.concatenate(widths),
                                    np.asarray(clip_offsets, np.int32))
    return headers, metas, frames_dev, np.asarray(clip_offsets, np.int32)


# ---- whole files on the device: gzip inflate + section index + frame unpack (cpx_cptv_inflate) -------------------

def parse_header_bytes(buf):
    """CptvHeader from the first bytes of an inflated file (magic, version, section H)."""
    if buf[:4] != b"CPTV" or buf[4] != 2 or buf[5:6] != b"H":
        raise ValueError("not a CPTV v2 header")
    fields, _ = _read_fields(buf, 6)
    return _header_from_fields(fields)


class DeviceFileBatch:
    """What inflate_files_on_device returns.  ``ok`` lists the indices (into the input list) of the files that
    decoded; headers / metas / clip_offsets are per those files, in that order; ``errors`` maps the index of every
    other file to a message."""

    def __init__(self):
        self.ok = []
        self.errors = {}
        self.headers = []
        self.slots = None         # CPTV_SLOT_DTYPE [total frames] (host)
        self.clip_offsets = None  # int32 [len(ok) + 1]
        self.frames_dev = None    # uint16 bits [total, H, W] (torch int16)
        self.results = None       # CPTV_RESULT_DTYPE per input file

    def frame_metas(self, k):
        """Metadata-only CptvFrame objects of the k-th decoded file (the reader API's per-frame fields)."""
        from ._lib import CPTV_BACKGROUND_FRAME, CPTV_HAS_LAST_FFC, CPTV_HAS_TIME_ON

        s = self.slots[self.clip_offsets[k]:self.clip_offsets[k + 1]]
        fl = s["flags"].tolist()
        return [CptvFrame(None, t if f & CPTV_HAS_TIME_ON else None, c if f & CPTV_HAS_LAST_FFC else None, a, b,
                          bool(f & CPTV_BACKGROUND_FRAME))
                for t, c, a, b, f in zip(s["time_on_ms"].tolist(), s["last_ffc_ms"].tolist(), s["temp_c"].tolist(),
                                         s["last_ffc_temp_c"].tolist(), fl)]


def inflate_files_on_device(engine, blobs, names=None):
    """The bytes of whole .cptv files -> frames on the device, without the host touching their content: upload,
    cpx_cptv_inflate (one wavefront per file: gzip + DEFLATE + section walk), cpx_cptv_gather_index,
    cpx_cptv_unpack.  A file that fails is reported in ``errors`` and leaves the others alone."""
    import ctypes as C

    from ._lib import (CPTV_FILE_DTYPE, CPTV_HEADER_BYTES, CPTV_RESULT_DTYPE, CPTV_SLOT_DTYPE, CPTV_STATUS, CpxError)

    t = engine.torch
    dev = engine.device
    n = len(blobs)
    out = DeviceFileBatch()
    if n == 0:
        return out
    P = engine.width * engine.height
    min_frame = 4 + (P - 1 + 7) // 8 + 8      # payload at one bit per delta + the smallest field list
    files = np.zeros(n, CPTV_FILE_DTYPE)
    sizes = np.array([len(b) for b in blobs], np.int64)
    in_off = np.zeros(n + 1, np.int64)
    np.cumsum((sizes + 15) & ~15, out=in_off[1:])
    isize = np.zeros(n, np.int64)
    for i, b in enumerate(blobs):
        if len(b) >= 18:
            isize[i] = int.from_bytes(bytes(b[-4:]), "little")
            # DEFLATE cannot expand by more than 1032 : 1; a trailer beyond that is not one (a truncated file's last
            # bytes): no capacity -> the file fails with "output" / "input exhausted" instead of reserving gigabytes
            if isize[i] > 1032 * len(b) + 64:
                isize[i] = 0
    files["in_offset"], files["in_bytes"] = in_off[:-1], sizes
    cap = (isize + 15) & ~15
    out_off = np.zeros(n + 1, np.int64)
    np.cumsum(cap + 16, out=out_off[1:])
    files["out_offset"], files["out_capacity"] = out_off[:-1], isize
    slot_cap = isize // min_frame + 1
    slot_off = np.zeros(n + 1, np.int64)
    np.cumsum(slot_cap, out=slot_off[1:])
    files["slot_offset"], files["slot_capacity"] = slot_off[:-1], slot_cap
    stage = t.empty(int(in_off[-1]) + 16, dtype=t.uint8, pin_memory=True)
    sv = stage.numpy()
    for i, b in enumerate(blobs):
        sv[in_off[i]:in_off[i] + sizes[i]] = np.frombuffer(b, np.uint8)
    in_dev = stage.to(dev, non_blocking=True)
    files_dev = engine._to_dev(files)
    out_dev = t.empty(int(out_off[-1]) + 16, dtype=t.uint8, device=dev)
    slots_dev = t.empty(max(int(slot_off[-1]), 1) * 8, dtype=t.int32, device=dev)
    header_dev = t.empty((n, CPTV_HEADER_BYTES), dtype=t.uint8, device=dev)
    results_dev = t.zeros(n * 10, dtype=t.int32, device=dev)
    engine.sync_inputs()
    p = lambda x: C.c_void_p(x.data_ptr())
    rc = engine.lib.cpx_cptv_inflate(engine.h, p(in_dev), p(files_dev), n, p(out_dev), p(slots_dev), p(header_dev),
                                     p(results_dev))
    if rc != 0:
        raise CpxError(rc, engine._err())
    engine.synchronize()
    res = results_dev.cpu().numpy().view(CPTV_RESULT_DTYPE).reshape(-1)
    out.results = res
    hdr = header_dev.cpu().numpy()
    for i in range(n):
        st = int(res["status"][i])
        name = names[i] if names else "file %d" % i
        if st != 0:
            out.errors[i] = "%s: %s" % (name, CPTV_STATUS.get(st, "status %d" % st))
        elif (int(res["width"][i]), int(res["height"][i])) != (engine.width, engine.height):
            out.errors[i] = "%s is %dx%d, the engine was created for %dx%d" % (
                name, res["width"][i], res["height"][i], engine.width, engine.height)
        elif int(res["header_bytes"][i]) > CPTV_HEADER_BYTES:
            out.errors[i] = "%s: header section of %d bytes" % (name, res["header_bytes"][i])
        else:
            out.ok.append(i)
    out.inflated_dev = out_dev          # kept for tests (the inflated bytes, file i at files["out_offset"][i])
    out.files = files
    if not out.ok:
        out.clip_offsets = np.zeros(1, np.int32)
        return out
    ok = np.asarray(out.ok)
    out.headers = [parse_header_bytes(hdr[i].tobytes()) for i in out.ok]
    offs = np.zeros(len(ok) + 1, np.int32)
    np.cumsum(res["n_frames"][ok], out=offs[1:])
    total = int(offs[-1])
    out.clip_offsets = offs
    offs_dev = t.from_numpy(offs).to(dev)
    so_dev = t.from_numpy(np.ascontiguousarray(slot_off[:-1][ok])).to(dev)
    fo_dev = t.empty(total, dtype=t.int64, device=dev)
    bw_dev = t.empty(total, dtype=t.int32, device=dev)
    dense_dev = t.empty(total * 8, dtype=t.int32, device=dev)
    frames_dev = t.empty((total, engine.height, engine.width), dtype=t.int16, device=dev)
    engine.sync_inputs()
    rc = engine.lib.cpx_cptv_gather_index(engine.h, p(slots_dev), p(so_dev), p(offs_dev), len(ok), p(fo_dev), p(bw_dev),
                                          p(dense_dev))
    if rc != 0:
        raise CpxError(rc, engine._err())
    rc = engine.lib.cpx_cptv_unpack(engine.h, p(out_dev), p(fo_dev), p(bw_dev), p(offs_dev), len(ok), p(frames_dev))
    if rc != 0:
        raise CpxError(rc, engine._err())
    engine.synchronize()
    out.slots = dense_dev.cpu().numpy().view(CPTV_SLOT_DTYPE).reshape(-1)
    out.frames_dev = frames_dev
    return out
