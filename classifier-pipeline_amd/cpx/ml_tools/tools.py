"""JSON helpers of the metadata path (reference src/ml_tools/tools.py:42-61,90-103)."""

import datetime
import json
from enum import Enum
from pathlib import Path

import numpy as np

from .rectangle import Rectangle


class CustomJSONEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.bool_):
            return bool(obj)
        if isinstance(obj, np.ndarray):
            return list(obj)
        if isinstance(obj, datetime.datetime):
            return obj.isoformat()
        if isinstance(obj, Rectangle):
            return obj.meta_dictionary()
        if isinstance(obj, Path):
            return str(obj)
        if isinstance(obj, Enum):
            return str(obj.name)
        return json.JSONEncoder.default(self, obj)


def load_clip_metadata(filename):
    with open(filename, "r") as fh:
        meta = json.load(fh)
    if meta.get("recordingDateTime"):
        from dateutil import parser

        meta["recordingDateTime"] = parser.parse(meta["recordingDateTime"])
    if meta.get("tracks") is None and meta.get("Tracks"):
        meta["tracks"] = meta["Tracks"]
    return meta


def eucl_distance_sq(first, second):
    dx = first[0] - second[0]
    dy = first[1] - second[1]
    return dx * dx + dy * dy
