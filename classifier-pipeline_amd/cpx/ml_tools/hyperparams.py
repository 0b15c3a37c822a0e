"""Model hyper-parameters with the reference's defaults (reference src/ml_tools/hyperparams.py:6-192)."""

from .datasetstructures import SegmentType

_DEFAULTS = {
    "model_name": "wr-resnet",
    "dense_sizes": None,
    "base_training": True,
    "retrain_layer": None,
    "dropout": 0.3,
    "learning_rate": 0.001,
    "learning_rate_decay": None,
    "use_movement": True,
    "use_segments": True,
    "frame_size": 32,
    "multi_label": True,
    "diff_norm": True,
    "thermal_diff_norm": False,
    "smooth_predictions": False,
    "channels": ["thermal", "filtered"],
    "keep_edge": True,
    "mvm": False,
}


class HyperParams(dict):
    """dict with attribute access and derived defaults."""

    def __init__(self, *args):
        super().__init__(*args)
        for key in ("model_name", "dense_sizes", "base_training", "retrain_layer", "dropout", "learning_rate",
                    "learning_rate_decay", "use_movement", "use_segments", "square_width", "frame_size",
                    "segment_width", "segment_types", "diff_norm", "thermal_diff_norm", "smooth_predictions",
                    "channels"):
            self[key] = getattr(self, key)
        self["multi_label"] = True

    def __getattr__(self, name):
        if name in _DEFAULTS:
            return self.get(name, _DEFAULTS[name])
        raise AttributeError(name)

    @property
    def square_width(self):
        return self.get("square_width", 5 if self.use_segments else 1)

    @property
    def segment_width(self):
        return self.get("segment_width", 25 if self.use_segments else 1)

    @property
    def segment_types(self):
        types = self.get("segment_types", [SegmentType.ALL_RANDOM_MASKED])
        if isinstance(types, str):
            return [SegmentType[types]]
        return [SegmentType[t] if isinstance(t, str) else t for t in types]

    @property
    def output_dim(self):
        side = self.frame_size * (self.square_width if self.use_movement else 1)
        return (side, side, len(self.channels))

    @property
    def excluded_labels(self):
        return self.get("excluded_labels")

    @property
    def remapped_labels(self):
        return self.get("remapped_labels")
