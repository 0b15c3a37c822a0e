#!/usr/bin/env python3
"""Config golden: the REFERENCE's Config (config/config.py under oracle/refharness.py) loaded from its own
classifier_TEMPLATE.yaml and from no file at all -> tracking (thermal + IR) / classify sections as plain dictionaries.
-> config_golden.json          python tests/golden/make_golden_config.py   (build container only)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
import refharness as rh  # noqa: E402


OVERRIDES = """
base_data_folder: "/tmp/clips"
worker_threads: 2
verbose: true
tracking:
  thermal:
    denoise: false
    edge_pixels: 2
    frame_padding: 6
    min_tracks: 1
    max_tracks: 5
    aoi_min_mass: 6.0
    filters:
      track_min_offset: 5.0
      min_duration_secs: 1.5
    tracker: RegionTracker
    params:
      base_distance_change: 500
      max_blanks: 10
  IR:
    min_dimension: 20
classify:
  meta_to_stdout: false
  models:
    - id: 3
      name: "wr"
      model_file: "/models/wr.npz"
      thumbnail_model: true
      ignored_tags: ["false-positive"]
"""


def plain(o):
    import attr
    from enum import Enum
    from pathlib import Path

    if attr.has(type(o)):
        return {k: plain(v) for k, v in attr.asdict(o, recurse=False).items()}
    if isinstance(o, dict):
        return {str(k): plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [plain(v) for v in o]
    if isinstance(o, Enum):
        return o.name
    if isinstance(o, Path):
        return str(o)
    if hasattr(o, "as_dict") and not isinstance(o, (int, float, str, bool, type(None))):
        return plain(o.as_dict())
    return o


if __name__ == "__main__":
    rh.install()
    cfgmod = rh.ref("config.config")
    out = {}
    import io

    for name, cfg in (("overrides", cfgmod.Config.load_from_stream(io.StringIO(OVERRIDES))),
                      ("defaults", cfgmod.Config.get_defaults())):
        out[name] = {"tracking": {k: plain(v) for k, v in cfg.tracking.items()}, "classify": plain(cfg.classify),
                     "labels": plain(cfg.labels), "use_opt_flow": cfg.use_opt_flow, "verbose": cfg.verbose,
                     "worker_threads": cfg.worker_threads, "reprocess": cfg.reprocess}
    out["overrides_yaml"] = OVERRIDES
    with open(os.path.join(HERE, "config_golden.json"), "w") as fh:  # noqa
        json.dump(out, fh, indent=1, default=str)
    print(json.dumps(out["defaults"]["tracking"]["thermal"], default=str)[:900])
