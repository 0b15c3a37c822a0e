"""The DEFLATE decoder of the GPU front end (classifier-pipeline_amd/csrc/cpx_inflate_core.h) compiled for the HOST and
checked against zlib: the two fixture recordings, every compression level / strategy (stored, fixed and dynamic blocks,
long matches, overlapping copies), a multi-block stream with sync flushes, and corrupt streams (which must end with an
error status, never hang or write past the capacity)."""
import ctypes as C
import os
import subprocess
import zlib

import numpy as np
import pytest

from helpers import GOLDEN

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("inflate") / "libinflate_host.so"
    subprocess.check_call([
        "g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(REPO, "classifier-pipeline_amd", "csrc"),
        os.path.join(REPO, "tests", "native", "inflate_host.cpp"), "-o", str(out)])
    lb = C.CDLL(str(out))
    lb.inflate_host_raw.argtypes = [C.c_char_p, C.c_long, C.c_void_p, C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lb.inflate_host_gzip_header.argtypes = [C.c_char_p, C.c_long]
    lb.inflate_host_gzip_header.restype = C.c_long
    return lb


def run(lib, raw, cap):
    out = np.zeros(cap + 64, np.uint8)
    out[cap:] = 0xA5  # canary
    n, used = C.c_long(0), C.c_long(0)
    rc = lib.inflate_host_raw(raw, len(raw), out.ctypes.data, cap, C.byref(n), C.byref(used))
    assert (out[cap:] == 0xA5).all(), "wrote past the capacity"
    return rc, out[: n.value].tobytes(), used.value


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=-15, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def corpus():
    rng = np.random.default_rng(7)
    noise = rng.integers(0, 256, 50000, dtype=np.uint8).tobytes()
    skew = np.clip(rng.normal(128, 6, 120000), 0, 255).astype(np.uint8).tobytes()   # literal-heavy, like packed deltas
    text = (b"the quick brown fox jumps over the lazy dog. " * 3000)[:100000]
    runs = b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 400), rng.integers(1, 700, 400)))
    mixed = skew[:30000] + text[:30000] + noise[:5000] + runs[:40000]
    return {"empty": b"", "one": b"x", "noise": noise, "skew": skew, "text": text, "runs": runs, "mixed": mixed,
            "zeros": bytes(300000)}


@pytest.mark.parametrize("name", ["possum", "hedgehog"])
def test_fixture_recordings(lib, name):
    raw = open(os.path.join(GOLDEN, name + ".cptv"), "rb").read()
    want = zlib.decompress(raw, 47)
    start = lib.inflate_host_gzip_header(raw, len(raw))
    assert start == 10
    rc, got, used = run(lib, raw[start:], len(want))
    assert rc == 0 and got == want
    assert start + used + 8 == len(raw)                                        # crc32 + isize follow the stream
    assert int.from_bytes(raw[-4:], "little") == len(want) and zlib.crc32(got) == int.from_bytes(raw[-8:-4], "little")


def test_levels_and_strategies(lib):
    for name, data in corpus().items():
        for level in (0, 1, 3, 6, 9):
            for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                for memlevel in (1, 8):
                    raw = deflate(data, level, strategy, memlevel=memlevel)
                    rc, got, used = run(lib, raw, len(data))
                    assert rc == 0 and got == data and used == len(raw), (name, level, strategy, memlevel, rc)


def test_small_windows_and_flushes(lib):
    data = corpus()["mixed"]
    for wbits in (-9, -12, -15):
        raw = deflate(data, 9, wbits=wbits)
        rc, got, _ = run(lib, raw, len(data))
        assert rc == 0 and got == data
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = []
    for i in range(0, len(data), 7777):   # sync / full flushes: empty stored blocks in the middle of the stream
        parts.append(c.compress(data[i:i + 7777]))
        parts.append(c.flush(zlib.Z_SYNC_FLUSH if (i // 7777) % 2 else zlib.Z_FULL_FLUSH))
    parts.append(c.flush())
    raw = b"".join(parts)
    rc, got, used = run(lib, raw, len(data))
    assert rc == 0 and got == data and used == len(raw)


def test_capacity_and_truncation(lib):
    data = corpus()["mixed"]
    raw = deflate(data, 6)
    rc, got, _ = run(lib, raw, len(data) - 1)            # one byte short
    assert rc == 7 and data.startswith(got)
    for cut in (1, 2, 10, len(raw) // 2, len(raw) - 1):
        rc, got, _ = run(lib, raw[:cut], len(data))
        assert rc != 0, cut
        assert data.startswith(got)


def test_corrupt_streams_end_with_an_error(lib):
    """Flipped bits: the decoder must return (any status), stay inside the capacity, and when it reports OK the
    output must be what zlib makes of the same bytes."""
    rng = np.random.default_rng(11)
    data = corpus()["mixed"]
    raw = bytearray(deflate(data, 6))
    n_err = 0
    for _ in range(300):
        bad = bytearray(raw)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        rc, got, _ = run(lib, bytes(bad), len(data) + 1000)
        d = zlib.decompressobj(-15)
        try:
            want = d.decompress(bytes(bad), len(data) + 1000)
            z_ok = d.eof
        except zlib.error:
            want, z_ok = None, False
        if rc == 0:
            assert z_ok and got == want
        else:
            n_err += 1
            assert not z_ok or len(want) > len(data) + 999
    assert n_err > 20
    assert run(lib, bytes([0x07]), 100)[0] == 1          # reserved block type
    assert run(lib, bytes([0x01, 0x05, 0x00, 0x00, 0x00]), 100)[0] == 2   # stored: LEN / NLEN mismatch
