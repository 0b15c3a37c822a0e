"""YAML -> configuration objects with the reference's defaults and merge rules
(reference src/config/config.py:19-115, defaultconfig.py:36-43,
trackingconfig.py:68-177, trackingmotionconfig.py:15-133, classifyconfig.py:31-123).
Only the sections the hot path reads are modelled: tracking (per camera type),
classify, and the top-level switches."""

import copy
import logging
from pathlib import Path

import yaml

CONFIG_FILENAME = "classifier.yaml"
CONFIG_DIRS = [Path("/etc/cacophony"), Path(__file__).resolve().parent.parent]


def deep_merge_missing(defaults, target):
    """Fill keys missing from `target` with `defaults`, recursively (defaultconfig.py:36-43)."""
    for key, value in defaults.items():
        if isinstance(value, dict):
            deep_merge_missing(value, target.setdefault(key, {}))
        elif key not in target:
            target[key] = value
    return target


class _Section:
    """Attribute bag that can round-trip to a plain dict (for metadata JSON)."""

    _fields = ()

    def __init__(self, **kw):
        for f in self._fields:
            setattr(self, f, kw[f])

    def as_dict(self):
        out = {}
        for f in self._fields:
            v = getattr(self, f)
            if isinstance(v, _Section):
                v = v.as_dict()
            elif isinstance(v, dict):
                v = {k: (x.as_dict() if isinstance(x, _Section) else copy.deepcopy(x)) for k, x in v.items()}
            elif isinstance(v, list):
                v = [x.as_dict() if isinstance(x, _Section) else x for x in v]
            out[f] = v
        return out

    def validate(self):
        return True


class ThresholdConfig(_Section):
    _fields = ("camera_model", "temp_thresh", "background_thresh", "default", "min_temp_thresh",
               "max_temp_thresh", "track_min_delta", "track_max_delta")

    DEFAULTS = dict(camera_model="default-model", temp_thresh=2900, background_thresh=20, default=False,
                    min_temp_thresh=None, max_temp_thresh=None, track_min_delta=1.0, track_max_delta=150)

    @classmethod
    def load(cls, raw):
        return cls(**deep_merge_missing(cls.DEFAULTS, dict(raw)))


class TrackingMotionConfig(_Section):
    _fields = ("camera_thresholds", "dynamic_thresh")

    @classmethod
    def get_defaults(cls):
        mk = ThresholdConfig
        th = {
            "lepton3": mk(**dict(mk.DEFAULTS, camera_model="lepton3", temp_thresh=2900, background_thresh=20, default=True)),
            "lepton3.5": mk(**dict(mk.DEFAULTS, camera_model="lepton3.5", temp_thresh=28000, background_thresh=50)),
            "IR": mk(**dict(mk.DEFAULTS, camera_model="IR", temp_thresh=None, background_thresh=12)),
        }
        return cls(camera_thresholds=th, dynamic_thresh=True)

    @classmethod
    def load(cls, raw):
        if raw is None:
            return cls.get_defaults()
        if isinstance(raw, TrackingMotionConfig):
            return raw
        ct = raw.get("camera_thresholds")
        thresholds = None
        if ct is not None:
            thresholds = {}
            for t in ct.values():
                t = t if isinstance(t, ThresholdConfig) else ThresholdConfig.load(t)
                thresholds[t.camera_model] = t
        return cls(camera_thresholds=thresholds, dynamic_thresh=raw["dynamic_thresh"])

    def threshold_for_model(self, camera_model):
        """Per-camera thresholds with fall-back to the default one (trackingmotionconfig.py:76-86)."""
        if self.camera_thresholds is None:
            return None
        t = self.camera_thresholds.get(camera_model)
        if t:
            return t
        for cand in self.camera_thresholds.values():
            if cand.default:
                return cand
        return self.camera_thresholds["default-model"]


class TrackingConfig(_Section):
    _fields = ("tracker", "params", "type", "motion", "edge_pixels", "min_dimension", "frame_padding",
               "track_smoothing", "denoise", "high_quality_optical_flow", "max_tracks", "track_overlap_ratio",
               "min_duration_secs", "track_min_offset", "track_min_mass", "aoi_min_mass", "aoi_pixel_variance",
               "cropped_regions_strategy", "enable_track_output", "min_tag_confidence", "moving_vel_thresh",
               "min_moving_frames", "max_blank_percent", "max_mass_std_percent", "max_jitter", "filters",
               "areas_of_interest", "filter_regions_pre_match", "min_hist_diff")

    @classmethod
    def type_defaults_dict(cls, type):
        d = dict(
            tracker="RegionTracker",
            type="thermal",
            motion=TrackingMotionConfig.get_defaults().as_dict(),
            edge_pixels=1, frame_padding=4, min_dimension=0, track_smoothing=False, denoise=True,
            high_quality_optical_flow=False, max_tracks=None,
            filters=dict(track_overlap_ratio=0.5, min_duration_secs=0, track_min_offset=4.0, track_min_mass=2.0,
                         moving_vel_thresh=4),
            areas_of_interest=dict(min_mass=4.0, pixel_variance=2.0, cropped_regions_strategy="cautious"),
            min_tag_confidence=0.8, enable_track_output=True, min_moving_frames=2, max_blank_percent=30,
            max_mass_std_percent=0.55, max_jitter=20,
            params=dict(base_distance_change=450, min_mass_change=20, restrict_mass_after=1.5,
                        mass_change_percent=0.55, max_distance=2000, max_blanks=18, velocity_multiplier=2,
                        base_velocity=2),
            filter_regions_pre_match=True, min_hist_diff=None,
        )
        if type == "IR":
            d.update(type="IR", filter_regions_pre_match=False, min_dimension=10, frame_padding=10, edge_pixels=0,
                     params=dict(base_distance_change=12000, min_mass_change=None, restrict_mass_after=1.5,
                                 mass_change_percent=None, max_distance=30752, max_blanks=18,
                                 velocity_multiplier=8, base_velocity=10))
            d["areas_of_interest"].update(pixel_variance=0, min_mass=0)
            d["filters"].update(min_duration_secs=0, track_min_offset=7)
        return d

    @classmethod
    def from_dict(cls, raw, type):
        raw = deep_merge_missing(cls.type_defaults_dict(type), dict(raw or {}))
        f, aoi = raw["filters"], raw["areas_of_interest"]
        obj = cls(
            tracker=raw["tracker"], params=raw["params"], type=type, motion=TrackingMotionConfig.load(raw.get("motion")),
            edge_pixels=raw["edge_pixels"], min_dimension=raw["min_dimension"], frame_padding=raw["frame_padding"],
            track_smoothing=raw["track_smoothing"], denoise=raw["denoise"],
            high_quality_optical_flow=raw["high_quality_optical_flow"], max_tracks=raw["max_tracks"],
            track_overlap_ratio=f["track_overlap_ratio"], min_duration_secs=f["min_duration_secs"],
            track_min_offset=f["track_min_offset"], track_min_mass=f["track_min_mass"],
            aoi_min_mass=aoi["min_mass"], aoi_pixel_variance=aoi["pixel_variance"],
            cropped_regions_strategy=aoi["cropped_regions_strategy"], enable_track_output=raw["enable_track_output"],
            min_tag_confidence=raw["min_tag_confidence"], moving_vel_thresh=f["moving_vel_thresh"],
            min_moving_frames=raw["min_moving_frames"], max_blank_percent=raw["max_blank_percent"],
            max_mass_std_percent=raw["max_mass_std_percent"], max_jitter=raw["max_jitter"], filters=f,
            areas_of_interest=aoi, filter_regions_pre_match=raw["filter_regions_pre_match"],
            min_hist_diff=raw["min_hist_diff"],
        )
        return obj

    @classmethod
    def get_defaults(cls):
        out = {t: cls.from_dict({}, t) for t in ("thermal", "IR")}
        # the reference's in-code IR defaults (trackingconfig.py:179-208) change the filters / areas_of_interest
        # dictionaries but leave the attributes they were copied to at construction, and set track_min_offset = 20
        # on the attribute only; a configuration loaded from YAML takes the dictionary values (from_dict above)
        ir = out["IR"]
        ir.aoi_min_mass, ir.aoi_pixel_variance, ir.track_min_offset = 4.0, 2.0, 20
        return out

    @classmethod
    def load(cls, tracking):
        if tracking is None:
            return None
        out = {}
        for type, raw in tracking.items():
            if isinstance(raw, TrackingConfig):
                out[raw.type] = raw
            else:
                cfg = cls.from_dict(raw, type)
                out[cfg.type] = cfg
        return out

    def rescale(self, scale):
        self.frame_padding = int(scale * self.frame_padding)
        self.min_dimension = int(scale * self.min_dimension)
        for k in ("base_distance_change", "min_mass_change", "max_distance", "base_velocity"):
            if self.params.get(k):
                self.params[k] *= scale
        self.track_min_offset *= scale
        self.track_min_mass *= scale
        self.aoi_min_mass *= scale


class ModelConfig(_Section):
    DEFAULT_SCORE = 0
    _fields = ("id", "name", "type", "model_file", "model_weights", "wallaby", "tag_scores", "ignored_tags",
               "thumbnail_model", "reclassify", "submodel", "run_over_network", "port")

    @classmethod
    def load(cls, raw):
        scores = dict(raw.get("tag_scores", {}))
        scores.setdefault("default", cls.DEFAULT_SCORE)
        return cls(id=raw["id"], name=raw["name"], type=raw.get("type"), model_file=raw["model_file"],
                   model_weights=raw.get("model_weights"), wallaby=raw.get("wallaby", False), tag_scores=scores,
                   ignored_tags=raw.get("ignored_tags", []), thumbnail_model=raw.get("thumbnail_model", False),
                   reclassify=raw.get("reclassify"), submodel=raw.get("submodel", False),
                   run_over_network=raw.get("run_over_network", False), port=raw.get("port", 8123))

    def validate(self):
        if not Path(self.model_file).exists():
            logging.warning("%s does not exist", self.model_file)
        return True


PREVIEW_OPTIONS = ("none", "raw", "classified", "tracking", "boxes")


class ClassifyConfig(_Section):
    _fields = ("models", "meta_to_stdout", "preview", "cache_to_disk")

    @classmethod
    def get_defaults(cls):
        return cls(models=None, meta_to_stdout=False, preview="none", cache_to_disk=False)

    @classmethod
    def load(cls, raw):
        models = raw.get("models")
        if models is not None:
            models = [m if isinstance(m, ModelConfig) else ModelConfig.load(m) for m in models]
        preview = raw["preview"]
        pv = preview.lower() if preview is not None else None
        if pv not in PREVIEW_OPTIONS:
            raise Exception("Cannot parse preview as '{}'.  Valid options are {}.".format(preview, PREVIEW_OPTIONS))
        return cls(models=models, meta_to_stdout=raw["meta_to_stdout"], preview=pv, cache_to_disk=raw["cache_to_disk"])

    def validate(self):
        for m in self.models or []:
            m.validate()
        return True


class Config(_Section):
    DEFAULT_LABELS = ["bird", "cat", "false-positive", "hedgehog", "insect", "leporidae", "mustelid", "possum",
                      "rodent", "wallaby"]
    _fields = ("base_folder", "labels", "tracking", "classify", "reprocess", "previews_colour_map",
               "worker_threads", "debug", "use_opt_flow", "verbose")

    @classmethod
    def get_defaults(cls):
        return cls(base_folder=".", labels=list(cls.DEFAULT_LABELS), reprocess=True,
                   previews_colour_map="custom_colormap.dat", worker_threads=0,
                   tracking=TrackingConfig.get_defaults(), classify=ClassifyConfig.get_defaults(), debug=False,
                   use_opt_flow=False, verbose=False)

    @classmethod
    def load_from_file(cls, filename=None):
        if filename is None or not Path(filename).exists():
            filename = find_config()
        if filename is None:
            return cls.get_defaults()
        logging.info("Loading config from %s", filename)
        with open(filename) as stream:
            return cls.load_from_stream(stream)

    @classmethod
    def load_from_stream(cls, stream):
        raw = yaml.safe_load(stream) or {}
        d = cls.get_defaults()
        defaults = dict(base_folder=".", labels=d.labels, reprocess=True, previews_colour_map=d.previews_colour_map,
                        worker_threads=0, debug=False, use_opt_flow=False, verbose=False,
                        classify=ClassifyConfig.get_defaults().as_dict(),
                        tracking={t: TrackingConfig.type_defaults_dict(t) for t in ("thermal", "IR")})
        deep_merge_missing(defaults, raw)
        return cls(base_folder=Path(raw.get("base_data_folder", ".")), labels=raw["labels"],
                   tracking=TrackingConfig.load(raw["tracking"]), classify=ClassifyConfig.load(raw["classify"]),
                   reprocess=raw["reprocess"], previews_colour_map=raw["previews_colour_map"],
                   worker_threads=raw["worker_threads"], debug=raw["debug"], use_opt_flow=raw["use_opt_flow"],
                   verbose=raw["verbose"])

    def validate(self):
        for t in self.tracking.values():
            t.validate()
        self.classify.validate()
        return True


def find_config():
    for directory in CONFIG_DIRS:
        p = directory / CONFIG_FILENAME
        if p.is_file():
            return str(p)
    return None
