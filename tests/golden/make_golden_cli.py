#!/usr/bin/env python3
"""Generates tests/golden/cli_golden.json by IMPORTING THE REFERENCE's two command lines (src/extract.py:25-89,
src/classify/main.py:28-142, under oracle/refharness.py) and dumping the option tables of the argparse parsers their
main() functions build: option strings, dest, nargs, const, default, whether the option takes a value, and -- for options
with a type function -- what that function returns for a list of probe strings.  Only the JSON travels; the drop-in CLIs
(cpx/extract.py, cpx/classify/main.py) must accept every option string the same way (tests/test_cli_cpu.py).

Build container only:   python tests/golden/make_golden_cli.py
"""
import argparse
import json
import logging
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))

import refharness  # noqa: E402

PROBES = ["yes", "true", "t", "y", "1", "no", "false", "f", "n", "0", "True", "FALSE", "maybe", ""]


class _Captured(Exception):
    pass


def capture(module_name):
    """Run the reference module's main() until its parser is asked to parse: the parser is what we came for."""
    refharness.install()
    # absl is not installed here; the reference only silences its log handler (extract.py:26-27)
    absl = types.ModuleType("absl")
    absl_logging = types.ModuleType("absl.logging")
    absl_logging._absl_handler = logging.NullHandler()
    absl_logging._warn_preinit_stderr = False
    absl.logging = absl_logging
    sys.modules.setdefault("absl", absl)
    sys.modules.setdefault("absl.logging", absl_logging)
    mod = refharness.ref(module_name)
    holder = {}
    orig = argparse.ArgumentParser.parse_args

    def grab(self, *a, **k):
        holder["parser"] = self
        raise _Captured()

    argparse.ArgumentParser.parse_args = grab
    try:
        mod.main(["x"])
    except _Captured:
        pass
    finally:
        argparse.ArgumentParser.parse_args = orig
    return holder["parser"]


def table(parser):
    rows = []
    for a in parser._actions:
        if isinstance(a, argparse._HelpAction):
            continue
        row = {"option_strings": list(a.option_strings), "dest": a.dest, "nargs": a.nargs,
               "action": type(a).__name__, "const": a.const, "default": a.default,
               "takes_value": not isinstance(a, (argparse._StoreTrueAction, argparse._StoreFalseAction,
                                                 argparse._CountAction)),
               "positional": not a.option_strings}
        if a.type is not None:
            probe = {}
            for p in PROBES:
                try:
                    probe[p] = a.type(p)
                except Exception as e:  # noqa: BLE001
                    probe[p] = "error:" + type(e).__name__
            row["type_probe"] = probe
        rows.append(row)
    return rows


def main():
    out = {"extract": table(capture("extract"))}
    try:
        out["classify"] = table(capture("classify.main"))
    except Exception as e:  # noqa: BLE001 -- classify/main.py imports the TensorFlow-side modules at import time
        out["classify_error"] = "%s: %s" % (type(e).__name__, e)
    with open(os.path.join(HERE, "cli_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
