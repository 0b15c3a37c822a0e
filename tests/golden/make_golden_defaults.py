#!/usr/bin/env python3
"""Default HyperParams of the REFERENCE (ml_tools/hyperparams.py, under oracle/refharness.py), for an empty dict and
for a few overrides -> defaults_golden.json.      python tests/golden/make_golden_defaults.py   (build container only)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import refharness as rh  # noqa: E402


def plain_value(v):
    from enum import Enum

    if isinstance(v, Enum):
        return v.name
    if isinstance(v, (list, tuple)):
        return [plain_value(x) for x in v]
    return v


def plain(hp):
    return {k: plain_value(v) for k, v in hp.items()}


if __name__ == "__main__":
    rh.install()
    hp = rh.ref("ml_tools.hyperparams")
    cases = [{}, {"frame_size": 64}, {"use_segments": False}, {"segment_types": ["ALL_RANDOM"], "square_width": 3},
             {"channels": ["thermal"], "smooth_predictions": True}]
    out = []
    for c in cases:
        h = hp.HyperParams(dict(c))
        out.append({"input": c, "params": plain(h), "output_dim": list(h.output_dim) if hasattr(h, "output_dim") else None})
    with open(os.path.join(HERE, "defaults_golden.json"), "w") as fh:
        json.dump(out, fh, indent=1, default=lambda o: getattr(o, "name", str(o)))
    print(json.dumps(out[0], default=lambda o: getattr(o, "name", str(o)))[:700])
