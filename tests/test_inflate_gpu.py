"""Whole .cptv files decoded on the GPU (cpx_cptv_inflate: gzip + DEFLATE + section walk, one wavefront per file;
cpx_cptv_gather_index; cpx_cptv_unpack) against zlib and the host reader: inflated bytes, section index, frames --
for the fixture recordings, the same content recompressed into every DEFLATE block type, synthetic recordings of
other delta widths, and batches with corrupt members, which must fail alone."""
import os
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, encode_cptv

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    yield eng
    eng.close()


def regzip(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, memlevel, strategy)
    return c.compress(data) + c.flush()


def host_view(blob):
    """(inflated bytes, frames uint16 [n, H, W], per-frame (time_on, last_ffc, temp, ffc temp, background)) by the host reader."""
    import tempfile

    from cpx.cptv import CptvReader

    with tempfile.NamedTemporaryFile(suffix=".cptv") as fh:
        fh.write(blob)
        fh.flush()
        r = CptvReader(fh.name)
        fr = r.read_all()
    return r.inflated, np.stack([f.pix for f in fr]), [(f.time_on, f.last_ffc_time, f.temp_c, f.last_ffc_temp_c,
                                                         f.background_frame) for f in fr], r.get_header()


def check(engine, blobs):
    from cpx.cptv import inflate_files_on_device

    got = inflate_files_on_device(engine, blobs)
    assert got.errors == {}, got.errors
    assert got.ok == list(range(len(blobs)))
    frames = got.frames_dev.cpu().numpy().view(np.uint16)
    raw = got.inflated_dev.cpu().numpy()
    for k, blob in enumerate(blobs):
        want_bytes, want_frames, want_meta, hdr = host_view(blob)
        o = int(got.files["out_offset"][k])
        assert int(got.results["out_bytes"][k]) == len(want_bytes)
        assert raw[o:o + len(want_bytes)].tobytes() == want_bytes, k
        f0, f1 = int(got.clip_offsets[k]), int(got.clip_offsets[k + 1])
        assert f1 - f0 == len(want_frames)
        assert np.array_equal(frames[f0:f1], want_frames), k
        metas = got.frame_metas(k)
        assert [(m.time_on, m.last_ffc_time, m.temp_c, m.last_ffc_temp_c, m.background_frame) for m in metas] == want_meta
        h = got.headers[k]
        assert (h.model, h.brand, h.timestamp, h.x_resolution, h.y_resolution, h.fps, h.device_name) == (
            hdr.model, hdr.brand, hdr.timestamp, hdr.x_resolution, hdr.y_resolution, hdr.fps, hdr.device_name)
    return got


def fixture(name):
    with open(os.path.join(GOLDEN, name + ".cptv"), "rb") as fh:
        return fh.read()


def test_fixture_recordings(engine):
    check(engine, [fixture("possum"), fixture("hedgehog")])


def test_every_block_type(engine):
    """The possum recording's content behind stored blocks (level 0), the fixed code, Huffman only, run-length matches
    and small / large hash tables: every decoding path of RFC 1951."""
    data = zlib.decompress(fixture("possum"), 47)[:600000]
    # cut at a frame boundary: keep whole sections only (the section walk rejects a truncated frame)
    from cpx.cptv import CptvReader

    r = CptvReader(os.path.join(GOLDEN, "possum.cptv"))
    _, offsets, widths = r.scan()
    P = 160 * 120
    ends = [int(o) + 4 + ((P - 1) * int(w) + 7) // 8 for o, w in zip(offsets, widths)]
    data = r.inflated[:[e for e in ends if e <= 600000][-1]]
    blobs = [regzip(data, 0), regzip(data, 1), regzip(data, 9), regzip(data, 6, zlib.Z_FIXED),
             regzip(data, 6, zlib.Z_HUFFMAN_ONLY), regzip(data, 6, zlib.Z_RLE), regzip(data, 9, memlevel=1)]
    check(engine, blobs)


def test_synthetic_widths_and_ragged_batch(engine, tmp_path):
    from cpx import synth

    rng = np.random.default_rng(5)
    blobs = []
    for k, (n, w) in enumerate(((3, 16), (40, 12), (17, 9), (1, 32), (64, 16))):
        clip = synth.make_clip(rng, n, max_blobs=2)
        p = tmp_path / ("s%d.cptv" % k)
        encode_cptv(p, clip, [w] * n, time_on=[1000 + 111 * i for i in range(n)], last_ffc=[7] * n,
                    background_first=(k % 2 == 0))
        blobs.append(p.read_bytes())
    check(engine, blobs)


def test_corrupt_members_fail_alone(engine):
    from cpx.cptv import inflate_files_on_device

    good = fixture("hedgehog")
    rng = np.random.default_rng(3)
    flipped = bytearray(good)
    for _ in range(40):
        flipped[int(rng.integers(100, len(flipped) - 8))] ^= 0xFF
    not_cptv = regzip(b"x" * 5000)
    two_members = good + regzip(b"tail")
    blobs = [good, good[: len(good) // 2], bytes(flipped), b"not gzip at all....................", not_cptv,
             two_members, fixture("possum"), b""]
    got = inflate_files_on_device(engine, blobs, names=["f%d" % i for i in range(len(blobs))])
    assert got.ok == [0, 6], (got.ok, got.errors)
    assert set(got.errors) == {1, 2, 3, 4, 5, 7}
    assert int(got.results["status"][3]) == 10 and int(got.results["status"][4]) == 20
    # a second gzip member: the trailer at the end of the file belongs to it, so the first member either overflows the
    # capacity taken from it (7) or ends before the end of the file (11) -- both send the file to the host reader
    assert int(got.results["status"][5]) in (7, 11)
    frames = got.frames_dev.cpu().numpy().view(np.uint16)
    for k, name in enumerate(("hedgehog", "possum")):
        _, want, _, _ = host_view(fixture(name))
        assert np.array_equal(frames[got.clip_offsets[k]:got.clip_offsets[k + 1]], want)


def test_match_shapes_against_zlib(engine):
    """The decoder's match paths one by one (the in-loop copy takes root-table codes of <= 64 bytes whose source lies
    before the pending literals; everything else leaves the loop): short matches in low-entropy noise, run-length
    chains (distance 1 and 3, length 258), a source that overlaps the literals still pending, distances up to the
    32 KiB window, matches that end exactly at the output capacity -- every DEFLATE level and the fixed code.  The
    payloads are not recordings: the inflate must succeed bit for bit and the section walk then refuse them (20)."""
    from cpx.cptv import inflate_files_on_device

    rng = np.random.default_rng(11)
    noise4 = rng.integers(0, 4, 300000, dtype=np.uint8).tobytes()            # dense short matches, all distances
    noise16 = rng.integers(0, 16, 200000, dtype=np.uint8).tobytes()
    chunk = rng.integers(0, 256, 300, dtype=np.uint8).tobytes()
    far = b"".join(chunk + rng.integers(0, 256, 32768 - 300, dtype=np.uint8).tobytes() for _ in range(4)) + chunk
    runs = b"a" * 5000 + b"abc" * 3000 + bytes(range(256)) * 40 + b"\0" * 70000
    echo = b"".join(rng.integers(0, 256, 5, dtype=np.uint8).tobytes() * 3 for _ in range(20000))   # literals, then their echo
    mixed = noise4[:50000] + runs + far + echo[:60000] + noise16[:50000]
    payloads = [noise4, noise16, far, runs, echo, mixed]
    blobs, want = [], []
    for data in payloads:
        for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_RLE)):
            blobs.append(regzip(data, level, strategy))
            want.append(data)
    got = inflate_files_on_device(engine, blobs, names=["m%d" % i for i in range(len(blobs))])
    raw = got.inflated_dev.cpu().numpy()
    for k, data in enumerate(want):
        assert int(got.results["status"][k]) == 20, (k, int(got.results["status"][k]))    # inflated; not a recording
        assert int(got.results["out_bytes"][k]) == len(data), k
        o = int(got.files["out_offset"][k])
        assert raw[o:o + len(data)].tobytes() == data, k


def test_gzip_crc_is_checked_on_the_device(engine):
    """A stream that inflates cleanly and matches ISIZE but carries other bytes than its CRC-32 says (a flipped bit inside
    a stored block, or a wrong CRC field) is refused with status 12 -- what zlib / the reference's gzip reader do --
    while its neighbours decode; sizes around the 64-chunk split (1 ... 300 bytes, odd lengths) are covered."""
    from cpx.cptv import inflate_files_on_device

    good = fixture("hedgehog")
    data = zlib.decompress(good, 47)
    stored = regzip(data, level=0)                       # stored blocks: payload bytes sit in the file as they are
    flipped = bytearray(stored)
    flipped[len(stored) // 2] ^= 0x10                    # a payload byte (block headers are 5 bytes every 64 KiB)
    bad_field = bytearray(good)
    bad_field[-8] ^= 0x01                                # the CRC field itself
    small = [regzip(bytes((7 * i + 3) & 0xFF for i in range(n))) for n in (1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 300)]
    blobs = [good, bytes(flipped), bytes(bad_field), stored] + small
    got = inflate_files_on_device(engine, blobs, names=["c%d" % i for i in range(len(blobs))])
    st = [int(x) for x in got.results["status"]]
    assert st[0] == 0 and st[3] == 0 and got.ok == [0, 3]
    assert st[1] == 12 and st[2] == 12, st[:4]
    assert st[4:] == [20] * len(small), st[4:]           # CRC fine (else 12); not recordings
