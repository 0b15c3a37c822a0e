"""The drop-in command lines accept the reference's flags, one for one.  tests/golden/cli_golden.json is the option table
of the parsers the REFERENCE's own main() functions build (src/extract.py:25-89, src/classify/main.py:28-142), dumped by
tests/golden/make_golden_cli.py under oracle/refharness.py; cpx/extract.py and cpx/classify/main.py must parse every
option string to the same destination, arity, constant, default and value type."""
import argparse
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli_golden.json")


def _parsers():
    from cpx import extract
    from cpx.classify import main as classify_main

    return {"extract": extract.build_parser(), "classify": classify_main.build_parser()}


def _actions(parser):
    out = {}
    for a in parser._actions:
        for s in a.option_strings or [a.dest]:
            out[s] = a
    return out


@pytest.mark.parametrize("which", ["extract", "classify"])
def test_every_reference_option_parses_the_same_way(which):
    with open(GOLDEN) as fh:
        golden = json.load(fh)
    assert which in golden, golden.get(which + "_error")
    parser = _parsers()[which]
    mine = _actions(parser)
    for row in golden[which]:
        names = row["option_strings"] or [row["dest"]]
        for s in names:
            assert s in mine, "%s: the reference's %s is not accepted" % (which, s)
            a = mine[s]
            assert a.dest == row["dest"], s
            assert a.nargs == row["nargs"], s
            assert a.const == row["const"], s
            assert a.default == row["default"], s
            takes = not isinstance(a, (argparse._StoreTrueAction, argparse._StoreFalseAction, argparse._CountAction))
            assert takes == row["takes_value"], s
            assert type(a).__name__ == row["action"], s
            if "type_probe" in row:
                assert a.type is not None, s
                for word, want in row["type_probe"].items():
                    try:
                        got = a.type(word)
                    except Exception as e:  # noqa: BLE001
                        got = "error:" + type(e).__name__
                    assert got == want, (s, word)


def test_reference_command_lines_parse():
    """The spellings a user of the reference types (the flags VERDICT r04 listed as rejected)."""
    from cpx import extract
    from cpx.classify import main as classify_main

    a = extract.parse_args(["-o", "-vv", "--retrack", "true", "--cache", "-p", "tracking", "-T", "-c", "x.yaml", "all"])
    assert (a.meta_to_stdout, a.verbose, a.retrack, a.cache, a.preview_type, a.timestamps, a.config_file, a.source) == \
        (1, 2, True, True, "tracking", True, "x.yaml", "all")
    a = extract.parse_args(["--retrack", "no", "--cache", "0", "clip.cptv"])
    assert (a.retrack, a.cache, a.meta_to_stdout, a.verbose) == (False, False, None, None)
    a = extract.parse_args(["--meta-to-stdout", "--retrack", "--", "clip.cptv"])
    assert a.retrack is True and a.meta_to_stdout == 1
    with pytest.raises(SystemExit):
        extract.parse_args(["--retrack", "perhaps", "clip.cptv"])
    c = classify_main.parse_args(["-t", "-w", "w.weights.h5", "-m", "model.tflite", "-p", "none", "-oo", "-v",
                                  "--reuse-prediction-frames", "--calculate-thumbnails", "--cache", "yes", "dir"])
    assert (c.track, c.model_weights, c.model_file, c.preview_type, c.meta_to_stdout, c.verbose,
            c.reuse_prediction_frames, c.calculate_thumbnails, c.cache, c.source) == \
        (True, "w.weights.h5", "model.tflite", "none", 2, 1, 1, True, True, "dir")
    c = classify_main.parse_args(["--track", "clip.cptv"])
    assert c.track and c.cache is None and c.model_weights is None and not c.post_process


def test_options_without_a_kernel_behind_them_say_so_when_reached(tmp_path):
    """--cache true (the reference's disk cache of frames) and a preview type parse; the class that would have honoured
    them raises NotImplementedError -- no silent ignore, no parse error."""
    from cpx.config import Config
    from cpx.track.framebuffer import FrameBuffer

    with pytest.raises(NotImplementedError):
        FrameBuffer(str(tmp_path / "x.cptv"), False, True, False, True, None)
    cfg = Config.get_defaults()
    cfg.classify.preview = "tracking"
    from cpx.classify.clipclassifier import ClipClassifier

    with pytest.raises(NotImplementedError):
        ClipClassifier(cfg, None)
