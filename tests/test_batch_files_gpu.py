"""Directory / multi-file drivers (extract_files, ClipClassifier.process_files): many recordings decoded, tracked and
associated as one device batch must give, per file, the metadata of the one-file-at-a-time path."""
import json
import os
import shutil

import numpy as np
import pytest

from helpers import encode_cptv

pytestmark = pytest.mark.gpu


def _strip(meta):
    from cpx.ml_tools.tools import CustomJSONEncoder

    m = json.loads(json.dumps(meta, cls=CustomJSONEncoder))
    m.pop("tracking_time", None)
    m.pop("source", None)
    m.pop("id", None)  # Clip.CLIP_ID: a per-process counter
    for model in m.get("models", []):
        model.pop("classify_time", None)
    for t in m.get("tracks", []):
        for p in t.get("predictions", []) or []:
            p.pop("classify_time", None)
    return m


def _files(golden_dir, tmp_path):
    """The two fixture clips, a lepton3.5 clip and a short synthetic clip without tracks (different lengths, two
    camera models -> two device groups)."""
    from cpx import synth

    paths = []
    for name in ("possum", "hedgehog"):
        dst = tmp_path / (name + ".cptv")
        shutil.copy(os.path.join(golden_dir, name + ".cptv"), dst)
        paths.append(dst)
    rng = np.random.default_rng(8)
    for k, (model, n, blobs) in enumerate((("lepton3.5", 70, 3), ("lepton3", 60, 2), ("lepton3", 25, 0))):
        clip = synth.make_clip(rng, n, model=model, max_blobs=blobs)
        p = tmp_path / ("synth%d.cptv" % k)
        t_on = [100000 + 111 * i for i in range(n)]
        encode_cptv(p, clip, [16] * n, time_on=t_on, last_ffc=[40000] * n, model=model.encode())
        paths.append(p)
    return paths


@pytest.mark.parametrize("denoise", [False, True])
def test_extract_files_equals_extract_file(golden_dir, tmp_path, denoise):
    from cpx.config import Config
    from cpx.track.trackextractor import extract_file, extract_files

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = denoise
    paths = _files(golden_dir, tmp_path)
    single = [_strip(extract_file(p, cfg, False, save_meta=False)[2]) for p in paths]
    batch = extract_files(paths, cfg, False, save_meta=False)
    assert len(batch) == len(paths)
    n_tracks = 0
    for (clip, ex, meta), want in zip(batch, single):
        got = _strip(meta)
        assert got == want, clip.source_file
        n_tracks += len(got["tracks"])
    assert n_tracks >= 4 and any(len(m["tracks"]) == 0 and m.get("thumbnail_region") is not None for m in single)


def test_directory_drivers_write_the_same_files(golden_dir, tmp_path):
    from cpx.config import Config
    from cpx.track.trackextractor import TrackExtractor, extract_file

    cfg = Config.get_defaults()
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    for p in _files(golden_dir, a):
        shutil.copy(p, b / p.name)
    for p in sorted(a.glob("*.cptv")):
        extract_file(p, cfg, False)
    ex = TrackExtractor(cfg)
    ex.batch_files = 3   # 5 files -> two device batches
    ex.extract(b)
    for p in sorted(a.glob("*.txt")):
        with open(p) as fa, open(b / p.name) as fb:
            ma, mb = json.load(fa), json.load(fb)
        for m in (ma, mb):
            m.pop("tracking_time", None)
            m.pop("source", None)
            m.pop("id", None)
        assert ma == mb, p.name


def test_directory_driver_isolates_bad_recordings(golden_dir, tmp_path, caplog):
    """A truncated recording, a corrupt one, one whose only defect is the gzip CRC, a file that is not gzip and an empty
    file in the directory: each is logged
    and skipped; every other recording gets the metadata it gets on its own."""
    import logging

    from cpx.config import Config
    from cpx.track.trackextractor import TrackExtractor, extract_file

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir()
    b.mkdir()
    good = _files(golden_dir, a)
    for p in good:
        shutil.copy(p, b / p.name)
    raw = (a / "possum.cptv").read_bytes()
    (b / "bad_truncated.cptv").write_bytes(raw[: len(raw) // 3])
    flipped = bytearray(raw)
    rng = np.random.default_rng(1)
    for _ in range(60):
        flipped[int(rng.integers(200, len(flipped) - 8))] ^= 0x5A
    (b / "bad_flipped.cptv").write_bytes(bytes(flipped))
    # decodes cleanly, sizes agree -- only the gzip CRC-32 tells that a payload bit flipped (stored blocks)
    import zlib

    comp = zlib.compressobj(0, zlib.DEFLATED, 31)
    stored = bytearray(comp.compress(zlib.decompress(raw, 47)) + comp.flush())
    stored[len(stored) // 2] ^= 0x04
    (b / "bad_crc.cptv").write_bytes(bytes(stored))
    (b / "bad_text.cptv").write_bytes(b"this is not a recording\n" * 10)
    (b / "bad_empty.cptv").write_bytes(b"")
    for p in sorted(a.glob("*.cptv")):
        extract_file(p, cfg, False)
    ex = TrackExtractor(cfg)
    with caplog.at_level(logging.WARNING):
        ex.extract(b)
    for name in ("bad_truncated", "bad_flipped", "bad_crc", "bad_text", "bad_empty"):
        assert not (b / (name + ".txt")).exists(), name
        assert any(name in r.getMessage() for r in caplog.records), name
    for p in sorted(a.glob("*.txt")):
        with open(p) as fa, open(b / p.name) as fb:
            ma, mb = json.load(fa), json.load(fb)
        for m in (ma, mb):
            m.pop("tracking_time", None)
            m.pop("source", None)
            m.pop("id", None)
        assert ma == mb, p.name
    assert ex.last_run["files"] == len(good)


def test_bulk_metadata_text_is_json_dump_text(golden_dir, tmp_path):
    """The text the batched path writes (positions by cpx_format_regions, the rest by json) is, character for
    character, what json.dump(indent=4) writes for the same metadata -- and the one-line form with to_stdout."""
    from cpx.config import Config
    from cpx.ml_tools.tools import CustomJSONEncoder
    from cpx.track.bulk import extract_files_bulk

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    paths = _files(golden_dir, tmp_path)
    out, _ = extract_files_bulk(paths, cfg, save_meta=False, want_text=True)
    n_pos = 0
    for p in paths:
        text = out[str(p)]
        meta = json.loads(text)
        assert json.dumps(meta, indent=4, cls=CustomJSONEncoder) == text, p
        n_pos += sum(len(t["positions"]) for t in meta["tracks"])
    assert n_pos > 100


def test_bulk_pipeline_is_deterministic_under_load(tmp_path):
    """3,072 recordings (copies of 16 synthetic clips, sensor noise: half a million DEFLATE matches each) through
    run_files_bulk with a classifier -- inflate of batch k+1 on its own stream beside the tracking and the network of
    batch k: no file may be skipped and the metadata text of every copy of a clip must be the same, character for
    character (file name, clip id and timings aside).  Guards the inflate kernel's store -> load ordering (a match
    reads bytes the wave stored a moment ago) and the buffer hand-over between the pipeline's threads."""
    import re

    from cpx import synth
    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.config import Config
    from cpx.config.config import ModelConfig
    from cpx.cptv import encode_cptv as encode_recording
    from cpx.ml_tools import wrresnet as wr
    from cpx.track.bulk import run_files_bulk

    labels = ["bird", "cat", "false-positive", "possum", "rodent"]
    wr.save_model(str(tmp_path / "wr"), wr.random_weights(len(labels), seed=3), labels, hyperparams={"frame_size": 32})
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr", "model_file": str(tmp_path / "wr.npz")})]
    cfg.classify.meta_to_stdout = False
    rng = np.random.default_rng(77)
    T, ND, N = 120, 16, 3072
    t_on, ffc = synth.frame_times(T)
    distinct = [encode_recording(synth.make_clip(rng, T), t_on, ffc, level=6) for _ in range(ND)]
    blobs = [distinct[i % ND] for i in range(N)]
    names = ["d%05d.cptv" % i for i in range(N)]
    out, _ = run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=1024, track_files=512,
                            clip_classifier=ClipClassifier(cfg), blobs=blobs)

    def norm(text):
        text = re.sub(r'"(tracking_time|classify_time|predicted_time)": [^,\n]*', '"t": 0', text)
        text = re.sub(r'^    "id": \d+,', '    "id": 0,', text, flags=re.M)
        return re.sub(r"d\d{5}\.cptv", "F", text)

    assert not [k for k, v in out.items() if v.startswith("error")]
    ref = [norm(out[names[j]]) for j in range(ND)]
    assert sum('"tracking_score"' in r for r in ref) >= ND // 2      # the clips do carry tracks
    differ = [names[i] for i in range(N) if norm(out[names[i]]) != ref[i % ND]]
    assert differ == [], differ[:5]


def test_memory_budgets_split_batches_without_changing_results(golden_dir, tmp_path):
    """Decode launches bounded by compressed bytes and tracking groups bounded by frames (directories of long
    recordings must not take the device's memory with them): with budgets that force one recording per launch and per
    group the metadata text of every file is what the unconstrained run writes."""
    import re

    from cpx.config import Config
    from cpx.track.bulk import run_files_bulk

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    paths = [str(p) for p in _files(golden_dir, tmp_path)]
    ref, _ = run_files_bulk(paths, cfg, save_meta=False, want_text=True)
    tight, tr = run_files_bulk(paths, cfg, save_meta=False, want_text=True, decode_bytes=1 << 20, track_frames=100)
    mid, _ = run_files_bulk(paths, cfg, save_meta=False, want_text=True, batch_files=3, track_files=2, track_frames=300)
    # (the clip id is a counter of Clip objects in the process: it differs from run to run)
    norm = lambda t: re.sub(r'^    "id": \d+,', '    "id": 0,', re.sub(r'"tracking_time": [^,\n]*', '"tracking_time": 0', t), flags=re.M)
    for p in paths:
        assert not ref[p].startswith("error")
        assert norm(tight[p]) == norm(ref[p]) == norm(mid[p]), p
    assert tr.timings["files"] == len(paths)


def test_in_memory_recordings_fail_alone_and_valid_ones_are_never_lost(golden_dir, tmp_path):
    """run_files_bulk(blobs=...): a VALID recording the device decoder refuses (two gzip members: status 11, "further
    member") is retried from its bytes by the host reader -- the reference's one reader never loses a valid file
    (cliptrackextractor.py:108-129) -- and gives the text of the one-member copy; a recording whose header has no
    timestamp, one whose gzip trailer claims 3 GB, and a corrupt one each end as an "error: ..." entry of their own;
    the run completes and every other recording is unchanged (VERDICT r03 missing 4, ADVICE r03 bulk.py:839/227)."""
    import re
    import struct
    import zlib

    from cpx.config import Config
    from cpx.track.bulk import run_files_bulk

    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    raw = open(os.path.join(golden_dir, "possum.cptv"), "rb").read()
    plain = zlib.decompress(raw, 47)

    def gz(data, level=6):
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(data) + c.flush()

    two_members = gz(plain[:50000]) + gz(plain[50000:])
    # header without its 'T' (timestamp) field: field count - 1, the 10-byte field (len 8, code 'T', u64) cut out
    assert plain[:6] == b"CPTV\x02H"
    pos, fields = 7, []
    for _ in range(plain[6]):
        ln = plain[pos]
        fields.append(plain[pos:pos + 2 + ln])
        pos += 2 + ln
    kept = [f for f in fields if f[1:2] != b"T"]
    assert len(kept) == len(fields) - 1
    no_ts = gz(plain[:6] + bytes([len(kept)]) + b"".join(kept) + plain[pos:])
    big_claim = bytearray(gz(plain))
    big_claim[-4:] = struct.pack("<I", 3 << 30)
    corrupt = bytearray(raw)
    rng = np.random.default_rng(3)
    for _ in range(80):
        corrupt[int(rng.integers(100, len(corrupt) - 8))] ^= 0xA5
    hedgehog = open(os.path.join(golden_dir, "hedgehog.cptv"), "rb").read()
    names = ["one.cptv", "two.cptv", "nots.cptv", "big.cptv", "bad.cptv", "hedgehog.cptv"]
    blobs = [gz(plain), two_members, no_ts, bytes(big_claim), bytes(corrupt), hedgehog]
    names = [str(tmp_path / n) for n in names]
    out, _ = run_files_bulk(names, cfg, save_meta=False, want_text=True, blobs=blobs)
    assert set(out) == set(names)

    def norm(text):
        text = re.sub(r'"(tracking_time)": [^,\n]*', '"t": 0', text)
        text = re.sub(r'"id": \d+,', '"id": 0,', text)
        return json.loads(text.replace("two.cptv", "one.cptv"))

    one, two = out[names[0]], out[names[1]]
    assert not one.startswith("error") and not two.startswith("error"), (one[:200], two[:200])
    a, b = norm(one), norm(two)
    for m in (a, b):
        m.pop("source", None)
    assert a == b and len(a["tracks"]) == 2
    for k in (2, 3, 4):
        assert out[names[k]].startswith("error"), (names[k], out[names[k]][:200])
    assert not out[names[5]].startswith("error") and len(json.loads(out[names[5]])["tracks"]) == 1


def test_metadata_worker_processes_write_the_same_text(golden_dir, tmp_path, model_dir=None):
    """run_files_bulk with a MetaPool (the metadata text formatted by worker processes that never touch the GPU) gives,
    recording for recording, the text the in-process stage gives -- with and without a classifier."""
    import re

    import cnn_oracle as co
    from cpx.classify.clipclassifier import ClipClassifier
    from cpx.config import Config
    from cpx.config.config import ModelConfig
    from cpx.ml_tools import wrresnet as wr
    from cpx.track.bulk import MetaPool, run_files_bulk

    paths = [str(p) for p in _files(golden_dir, tmp_path)] * 3
    names = ["%02d_%s" % (k, os.path.basename(p)) for k, p in enumerate(paths)]
    blobs = [open(p, "rb").read() for p in paths]
    labels = ["bird", "cat", "deer", "dog", "false-positive", "hedgehog", "human", "kiwi", "leporidae", "mustelid", "penguin",
              "possum", "rodent", "sheep", "vehicle", "wallaby", "land-bird"]
    rng = np.random.default_rng(3)
    w = co.calibrate_bn(wr.random_weights(17, seed=1), rng.uniform(0, 255, size=(2, 160, 160, 2)).astype(np.float32))
    wr.save_model(tmp_path / "wr", w, labels, hyperparams={"frame_size": 32})
    cfg = Config.get_defaults()
    cfg.tracking["thermal"].denoise = False
    cfg.classify.models = [ModelConfig.load({"id": 1, "name": "wr", "model_file": str(tmp_path / "wr.npz")})]
    pool = MetaPool.make(2)
    assert pool is not None and len(pool.pids) >= 1 and os.getpid() not in pool.pids

    def strip(text):
        text = re.sub(r'"tracking_time": [0-9.e+-]+', '"tracking_time": 0', text)
        text = re.sub(r'"classify_time": [0-9.e+-]+', '"classify_time": 0', text)
        text = re.sub(r'"predicted_time": [0-9.e+-]+', '"predicted_time": 0', text)
        return re.sub(r'"id": [0-9]+,(\s+)"start_time"', r'"id": 0,\1"start_time"', text)   # Clip.CLIP_ID: a per-process counter

    try:
        for cc in (None, ClipClassifier(cfg)):
            plain, _ = run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=8, clip_classifier=cc,
                                      blobs=blobs, track_files=5)
            pooled, tr = run_files_bulk(names, cfg, save_meta=False, want_text=True, batch_files=8, clip_classifier=cc,
                                        blobs=blobs, track_files=5, meta_pool=pool)
            assert set(plain) == set(pooled) == set(names)
            for n in names:
                assert not plain[n].startswith("error") and strip(plain[n]) == strip(pooled[n]), n
            assert tr.timings["files"] == len(names) and tr.timings["frames"] > 0
            if cc is not None:
                assert any('"all_class_confidences"' in v for v in pooled.values())
    finally:
        pool.close()
