"""The GPU DEFLATE decoder parses untrusted bytes: thousands of corrupted gzip members per
launch -- bit flips, truncations, spliced streams, broken Huffman descriptions, matches that reach before the start
of the output, invalid length / distance symbols, stored blocks whose LEN / NLEN disagree -- each must END with a
status, agree with zlib on accept / reject, be byte-equal when accepted, leave its neighbours alone, and the launch
must finish in bounded time (a hang fails the assertion instead of wedging the box).  A second test repeats one
decode beside different co-runners (the convolutions, the NLM kernel, nothing): the bytes may not depend on what
else is on the chip (the round-3 store -> load race only showed under load)."""
import struct
import threading
import time
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ACCEPT = {0} | set(range(20, 28))      # gzip member valid; 20..27: what it holds is not a (well-formed) recording
LAUNCH_SECONDS = 60.0                  # bound on one fuzz launch (a clean launch of this size takes well under 1 s)


@pytest.fixture(scope="module")
def engine():
    from cpx.engine import TrackEngine

    eng = TrackEngine(model="lepton3")
    yield eng
    eng.close()


def gz(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
    return c.compress(data) + c.flush()


def gz_wrap(deflate_bytes, payload=b"x" * 64):
    """A gzip member around a raw DEFLATE stream that claims to hold `payload` (ISIZE is the capacity the device gives
    the stream: 64 bytes let the hand-made streams reach their defect instead of the end of the output)."""
    return (b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03" + deflate_bytes
            + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload) & 0xFFFFFFFF))


class Bits:
    """DEFLATE bit order: fields LSB first, Huffman codes MSB first."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, value, nbits):
        self.acc |= value << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, nbits):
        self.put(int(format(code, "0%db" % nbits)[::-1], 2), nbits)

    def fixed_sym(self, s):
        if s < 144:
            self.code(0x30 + s, 8)
        elif s < 256:
            self.code(0x190 + s - 144, 9)
        elif s < 280:
            self.code(s - 256, 7)
        else:
            self.code(0xC0 + s - 280, 8)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
        return bytes(self.out)


def handcrafted():
    """(name, gzip member) of streams that are wrong in exactly one documented way (RFC 1951)."""
    out = []

    def fixed(body, prefix=b""):
        b = Bits()
        b.put(1, 1); b.put(1, 2)            # BFINAL, fixed codes
        body(b)
        return gz_wrap(prefix + b.done())

    stored_ok = b"\x00\x05\x00\xfa\xffhello"          # a non-final stored block of five bytes
    for name, prefix in (("", b""), ("after_stored_", stored_ok)):
        out.append((name + "distance_before_start", fixed(lambda b: (b.fixed_sym(257), b.code(0, 5), b.fixed_sym(256)), prefix)
                    if not prefix else fixed(lambda b: (b.fixed_sym(257), b.code(5, 5), b.put(1, 1), b.fixed_sym(256)), prefix)))
        out.append((name + "length_symbol_286", fixed(lambda b: (b.fixed_sym(65), b.fixed_sym(286), b.code(0, 5), b.fixed_sym(256)), prefix)))
        out.append((name + "length_symbol_287", fixed(lambda b: (b.fixed_sym(65), b.fixed_sym(287), b.code(0, 5), b.fixed_sym(256)), prefix)))
        out.append((name + "distance_symbol_30", fixed(lambda b: (b.fixed_sym(65), b.fixed_sym(257), b.code(30, 5), b.fixed_sym(256)), prefix)))
        out.append((name + "distance_symbol_31", fixed(lambda b: (b.fixed_sym(65), b.fixed_sym(257), b.code(31, 5), b.fixed_sym(256)), prefix)))
        out.append((name + "no_end_of_block", fixed(lambda b: [b.fixed_sym(66) for _ in range(40)], prefix)))
        out.append((name + "reserved_block_type", gz_wrap(prefix + b"\x07")))
        out.append((name + "stored_len_nlen", gz_wrap(prefix + b"\x01\x05\x00\x00\x00hello")))
        out.append((name + "stored_longer_than_file", gz_wrap(prefix + b"\x01\xff\x7f\x00\x80abc")))

        def dyn(hlit, hdist, hclen, cl_lengths, tail=lambda b: None):
            b = Bits()
            b.put(1, 1); b.put(2, 2)
            b.put(hlit, 5); b.put(hdist, 5); b.put(hclen, 4)
            for v in cl_lengths:
                b.put(v, 3)
            tail(b)
            b.put(0, 32)
            return gz_wrap(prefix + b.done())

        out.append((name + "hlit_287_codes", dyn(30, 0, 15, [3] * 19)))
        out.append((name + "hlit_288_codes", dyn(31, 0, 15, [3] * 19)))
        out.append((name + "hdist_31_codes", dyn(0, 30, 15, [3] * 19)))
        out.append((name + "code_length_code_all_zero", dyn(0, 0, 15, [0] * 19)))
        out.append((name + "code_length_code_oversubscribed", dyn(0, 0, 15, [1] * 19)))
        out.append((name + "code_length_code_incomplete", dyn(0, 0, 15, [7] + [0] * 18)))
        # code length code: symbols 16 and 0 with one bit each; the first code is a repeat with nothing to repeat
        order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
        cl = [0] * 19
        cl[order.index(16)] = 1
        cl[order.index(0)] = 1
        out.append((name + "repeat_without_previous", dyn(0, 0, 15, cl, lambda b: (b.code(1, 1), b.put(3, 2)))))
        # every literal / length code length zero (257 zeros by 18-repeats): no end-of-block code can exist
        cl = [0] * 19
        cl[order.index(18)] = 1
        cl[order.index(0)] = 1
        out.append((name + "no_codes_at_all", dyn(0, 0, 15, cl, lambda b: [(b.code(1, 1), b.put(127, 7)) for _ in range(3)])))
    return out


def zlib_verdict(blob, limit):
    """(accepted, bytes): what zlib makes of a single gzip member that must end exactly at the end of the file."""
    d = zlib.decompressobj(31)
    try:
        data = d.decompress(blob, limit + 1)
    except zlib.error:
        return False, None
    if not d.eof or d.unused_data or d.unconsumed_tail or len(data) > limit:
        return False, None
    return True, data


def decode_batch(engine, blobs):
    from cpx.cptv import inflate_files_on_device

    t0 = time.time()
    got = inflate_files_on_device(engine, blobs, names=["z%d" % i for i in range(len(blobs))])
    seconds = time.time() - t0
    return got, seconds


def make_corpus(golden_dir):
    from cpx import synth
    from cpx.cptv import encode_cptv as encode_recording

    rng = np.random.default_rng(2024)
    t_on, ffc = synth.frame_times(10)
    plain = []
    for _ in range(3):
        rec = encode_recording(synth.make_clip(rng, 10), t_on, ffc, level=6)
        plain.append(zlib.decompress(rec, 47))
    bases = []
    for data in plain:
        for level, strategy in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (0, zlib.Z_DEFAULT_STRATEGY)):
            bases.append(gz(data, level, strategy))
    return bases, plain


def test_two_thousand_corrupted_members_per_launch(engine, golden_dir):
    bases, plain = make_corpus(golden_dir)
    rng = np.random.default_rng(7)
    variants, kinds = [], []
    # bit flips: anywhere, and concentrated where the Huffman descriptions and the first symbols live
    for k in range(1500):
        b = bytearray(bases[k % len(bases)])
        head = k % 3 == 0
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(10, 120)) if head else int(rng.integers(0, len(b)))
            b[pos] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(b)); kinds.append("flip")
    # flips whose DEFLATE stream still ends cleanly, with the trailer re-made for what it now inflates to: valid
    # members with unusual content (desynchronised symbols, long runs, matches into fresh territory) that zlib ACCEPTS
    n_fixed = 0
    for k in range(6000):
        if n_fixed >= 400:
            break
        b = bytearray(bases[k % len(bases)])
        for _ in range(int(rng.integers(1, 3))):
            b[int(rng.integers(10, len(b) - 8))] ^= 1 << int(rng.integers(0, 8))
        d = zlib.decompressobj(-15)
        try:
            data = d.decompress(bytes(b[10:-8]), 1 << 22)
        except zlib.error:
            continue
        if not d.eof or d.unused_data or d.unconsumed_tail:
            continue
        b[-8:] = struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)
        variants.append(bytes(b)); kinds.append("flip_fixed")
        n_fixed += 1
    assert n_fixed >= 100, n_fixed
    for k in range(120):                                   # the gzip header: MTIME / XFL / OS are free, FLG is not
        b = bytearray(bases[k % len(bases)])
        b[int(rng.integers(3, 10))] ^= 1 << int(rng.integers(0, 8))
        variants.append(bytes(b)); kinds.append("header")
    for k in range(300):                                   # truncations (incl. inside the header / the trailer)
        b = bases[k % len(bases)]
        cut = int(rng.integers(0, 30)) if k % 5 == 0 else int(rng.integers(1, len(b)))
        if k % 7 == 0:
            cut = len(b) - int(rng.integers(1, 9))
        variants.append(b[:cut]); kinds.append("cut")
    for k in range(300):                                   # splices: the head of one stream, the tail of another
        a, c = bases[int(rng.integers(0, len(bases)))], bases[int(rng.integers(0, len(bases)))]
        i, j = int(rng.integers(10, len(a))), int(rng.integers(10, len(c)))
        variants.append(a[:i] + c[j:]); kinds.append("splice")
    crafted = handcrafted()
    for name, blob in crafted:
        variants.append(blob); kinds.append(name)
        for _ in range(5):                                 # and their neighbourhood
            b = bytearray(blob)
            b[int(rng.integers(10, len(b)))] ^= 1 << int(rng.integers(0, 8))
            variants.append(bytes(b)); kinds.append(name + "~")
    # good members in between: nobody may disturb them
    good_every = 50
    blobs, is_good = [], []
    for i, v in enumerate(variants):
        if i % good_every == 0:
            blobs.append(bases[(i // good_every) % len(bases)]); is_good.append(True)
        blobs.append(v); is_good.append(False)
    assert sum(not g for g in is_good) >= 2000

    got, seconds = decode_batch(engine, blobs)
    assert seconds < LAUNCH_SECONDS, seconds
    st = [int(x) for x in got.results["status"]]
    raw = got.inflated_dev.cpu().numpy()
    limit = 1 << 22
    n_accept = n_reject = 0
    seen = {}
    vi = 0
    for k, blob in enumerate(blobs):
        o, cap = int(got.files["out_offset"][k]), int(got.files["out_capacity"][k])
        nbytes = int(got.results["out_bytes"][k])
        assert 0 <= nbytes <= cap, (k, nbytes, cap)        # never beyond the capacity it was given
        if is_good[k]:
            assert st[k] == 0, (k, st[k])
            want = zlib.decompress(blob, 47)
            assert raw[o:o + len(want)].tobytes() == want, k
            continue
        kind = kinds[vi]
        vi += 1
        ok, want = zlib_verdict(blob, limit)
        seen.setdefault(kind.rstrip("~"), set()).add(st[k])
        if st[k] in ACCEPT:
            n_accept += 1
            assert ok, (k, kind, st[k], "the device accepted a member zlib refuses")
            assert nbytes == len(want) and raw[o:o + nbytes].tobytes() == want, (k, kind)
        else:
            n_reject += 1
            assert 1 <= st[k] <= 12, (k, kind, st[k])
            # (a member whose header carries FHCRC is left to the host reader, valid or not: status 10)
            assert not ok or (st[k] == 10 and blob[3] & 2), (k, kind, st[k], "the device refused a member zlib accepts")
    assert n_reject > 1500 and n_accept > 150, (n_accept, n_reject)   # both verdicts are exercised
    # the hand-made streams end with the status their defect names
    expect = {"distance_before_start": {6}, "length_symbol_286": {5}, "length_symbol_287": {5}, "distance_symbol_30": {5},
              "distance_symbol_31": {5}, "reserved_block_type": {1}, "stored_len_nlen": {2}, "hlit_287_codes": {3, 4, 5},
              "hlit_288_codes": {3, 4, 5}, "hdist_31_codes": {3, 4, 5}, "code_length_code_all_zero": {3, 4, 5},
              "code_length_code_oversubscribed": {3, 4, 5}, "code_length_code_incomplete": {3, 4, 5},
              "repeat_without_previous": {3, 4, 5}}   # (3 = block header counts, 4 = code lengths, 5 = a symbol no code describes)
    by_name = {}
    for (name, _), k in zip(crafted, [i for i, kd in enumerate(kinds) if not kd.endswith("~") and kd not in ("flip", "flip_fixed", "header", "cut", "splice")]):
        by_name[name] = k
    vpos = [i for i, g in enumerate(is_good) if not g]
    for name, want_status in expect.items():
        for prefix in ("", "after_stored_"):
            k = vpos[by_name[prefix + name]]
            assert st[k] in want_status, (prefix + name, st[k])


def test_decode_does_not_depend_on_the_co_runner(engine):
    """1,024 noisy recordings (half a million matches each) decoded alone, beside the convolutions of the network
    and beside the NLM kernel on another stream: the same bytes every time, and zlib's."""
    import torch

    from cpx import synth
    from cpx.cptv import encode_cptv as encode_recording
    from cpx.engine import TrackEngine
    from cpx.ml_tools import wrresnet as wr

    rng = np.random.default_rng(99)
    T, ND, N = 60, 8, 1024
    t_on, ffc = synth.frame_times(T)
    distinct = [encode_recording(synth.make_clip(rng, T), t_on, ffc, level=6) for _ in range(ND)]
    want = [zlib.crc32(zlib.decompress(d, 47)) for d in distinct]
    blobs = [distinct[i % ND] for i in range(N)]

    def crcs():
        got, seconds = decode_batch(engine, blobs)
        assert seconds < LAUNCH_SECONDS
        assert all(int(s) == 0 for s in got.results["status"]), sorted(set(int(s) for s in got.results["status"]))
        raw = got.inflated_dev.cpu().numpy()
        out = []
        for k in range(N):
            o, nb = int(got.files["out_offset"][k]), int(got.results["out_bytes"][k])
            out.append(zlib.crc32(raw[o:o + nb].tobytes()))
        return out

    other = TrackEngine(model="lepton3", max_frames=8, denoise=True)
    stop = threading.Event()

    def conv_load():
        net = wr.WRResNetDevice(other, wr.random_weights(5, seed=1), 5)
        x = torch.rand((256, 160, 160, 2), device=other.device) * 255
        while not stop.is_set():
            y = net.forward(x)   # held until the stream has drained: a dropped output goes back to torch's
            other.synchronize()  # allocator while the kernels still write it, and the decode would be handed it
            del y

    def nlm_load():
        frames, _ = synth.make_batch(64, 8, seed=5)
        dev = torch.from_numpy(frames.view(np.int16)).to(other.device).repeat(4, 1, 1).contiguous()
        offs = (np.arange(257) * 8).astype(np.int32)
        meta = np.concatenate([other.make_meta(8) for _ in range(256)])
        while not stop.is_set():
            res = other.track_batch(dev, offs, meta, want_labels=False, want_filtered=False)
            other.synchronize()
            del res

    try:
        for load in (None, conv_load, nlm_load, None):
            stop.clear()
            th = None
            if load is not None:
                th = threading.Thread(target=load, daemon=True)
                th.start()
                time.sleep(0.5)
            try:
                for _ in range(3):
                    got = crcs()
                    bad = [k for k in range(N) if got[k] != want[k % ND]]
                    assert bad == [], (getattr(load, "__name__", "alone"), bad[:5])
            finally:
                stop.set()
                if th is not None:
                    th.join(timeout=120)
    finally:
        other.close()
