"""Interpreter: per-track classification with the reference's interface
(reference src/ml_tools/interpreter.py:13-474, 597-628).  Segment pre-processing (limits, crop,
resize, normalise, 5x5 tiling) and the CNN forward run on the GPU through the C-ABI; segment
selection and prediction aggregation are host work as in the reference."""

import json
import logging
import time
from pathlib import Path

import numpy as np

from .._lib import CROP_REQ_DTYPE, REGION_REF_DTYPE
from ..classify.trackprediction import TrackPrediction
from .datasetstructures import get_segments, segments_from_frames
from .hyperparams import HyperParams


class Interpreter:
    TYPE = "abstract"

    def __init__(self, model_file, run_over_network=False):
        self.model_file = Path(model_file)
        self.load_json(model_file)
        self.run_over_network = bool(run_over_network)
        self.port = 8123
        self.id = None
        self.seed = None

    def load_json(self, filename):
        """Model sidecar <model>.json: labels, hyperparams, thresholds, type, version (interpreter.py:23-41)."""
        filename = Path(filename).with_suffix(".json")
        logging.info("Loading metadata from %s", filename)
        with open(filename, "r") as fh:
            metadata = json.load(fh)
        self.version = metadata.get("version")
        self.labels = metadata["labels"]
        self.params = HyperParams()
        self.params["remapped_labels"] = metadata.get("remapped_labels")
        self.params["excluded_labels"] = metadata.get("excluded_labels")
        self.params.update(metadata.get("hyperparams", {}))
        self.data_type = metadata.get("type", "thermal")
        self.mapped_labels = metadata.get("mapped_labels")
        self.label_probabilities = metadata.get("label_probabilities")
        self.thresholds = metadata.get("thresholds")
        # what this build's network covers (ml_tools/kerasmodel.py:259-350): the WR-ResNet base, optional hidden
        # dense layers, sigmoid or softmax output -- anything else must fail here, not classify with a wrong head
        if self.params.get("mvm") or self.params.get("mvm_forest") or self.params.get("lstm") or \
                self.params.get("model_merge"):
            raise NotImplementedError("models with track-feature (mvm), LSTM or merged heads are not supported")
        self.preprocess_fn = self.get_preprocess_fn()

    # model families whose Keras preprocess_input runs in 'tf' mode (x / 127.5 - 1), and inceptionv3's own copy of it
    TF_SCALED_MODELS = ("inceptionv3", "nasnet", "resnetv2", "mobilenet", "inceptionresnetv2")
    # 'caffe' / 'torch' modes: per-channel means of THREE colour channels (and a BGR swap) -- nothing the two-channel
    # thermal samples of this path can take
    CHANNEL_MEAN_MODELS = ("resnet", "resnet152", "vgg16", "vgg19", "densenet121")

    def get_preprocess_fn(self):
        """interpreter.py:64-98: the input scaling of the model family.  None for wr-resnet / efficientnetv2b3; the
        'tf'-mode families (and inceptionv3, interpreter.py:563-566) scale the finished sample x / 127.5 - 1 -- done by
        the crop kernel (CPX_LIMITS_TF_SCALING), the returned function is its host statement for callers that hold a
        sample; unknown names get None with the reference's warning."""
        name = self.params.model_name
        if name in ("wr-resnet", "efficientnetv2b3"):
            return None
        if name in self.TF_SCALED_MODELS:
            return inc3_preprocess
        if name in self.CHANNEL_MEAN_MODELS:
            raise NotImplementedError("model_name %r: tf.keras.applications' 'caffe' / 'torch' preprocess_input subtracts "
                                      "the means of three colour channels; not built for the thermal samples" % name)
        logging.warning("pretrained model %s has no preprocessing function", name)
        return None

    def shape(self):
        raise NotImplementedError

    def predict(self, frames):
        raise NotImplementedError

    def predict_over_network(self, data):
        """POST the samples to a model server on this host (interpreter.py:53-62; server: cpx/servemodel.py)."""
        import urllib.request

        data = np.ascontiguousarray(data, dtype="<f4")
        req = urllib.request.Request("http://127.0.0.1:%d/predict" % self.port, data=data.tobytes(),
                                     headers={"content-type": "application/octet-stream"}, method="POST")
        with urllib.request.urlopen(req) as resp:
            body = resp.read()
        return np.frombuffer(body, dtype=np.float32).reshape(len(data), -1)

    # ---- per-track entry points (interpreter.py:132-176) ----
    def classify_track(self, clip, track, segment_frames=None, min_segments=None):
        start = time.time()
        frames, output, masses = self.predict_track(clip, track, segment_frames=segment_frames,
                                                    frames_per_classify=self.params.square_width**2,
                                                    min_segments=min_segments)
        if output is not None and self.params.square_width == 1 and not isinstance(frames, list):
            frames = list(frames)
        if output is None:
            logging.info("Skipping track %s", track.get_id())
            return None
        pred = self.track_prediction_from_raw(track.get_id(), frames, output, masses)
        pred.classify_time = time.time() - start
        return pred

    def track_prediction_from_raw(self, track_id, prediction_frames, output, masses):
        pred = TrackPrediction(track_id, self.labels, smooth_preds=self.params.smooth_predictions)
        pred.classified_track(output, prediction_frames, masses)
        # a single segment built from very few distinct frames: only 'false-positive' may be confident
        # (single-frame models hand frame NUMBERS here, not frame lists: one number counts as one distinct frame)
        if len(prediction_frames) == 1 and len(set(np.atleast_1d(prediction_frames[0]).tolist())) < self.params.square_width**2 / 4:
            if pred.predicted_tag() != "false-positive":
                pred.cap_confidences(0.5)
        return pred

    def preprocess(self, clip, track, samples, **args):
        """interpreter.py:110-130: segments for frames_per_classify > 1, single frames otherwise.  (The reference's own
        single-frame branch calls preprocess_frames(clip, track) without the samples at this snapshot and raises
        TypeError; the functions behind it are what tests/golden/classify_variants_golden.json was taken from.)"""
        if args.get("frames_per_classify", 25) > 1:
            return self.preprocess_segments(clip, track, samples, predict_from_last=args.get("predict_from_last"))
        return self.preprocess_frames(clip, track, samples)

    def predict_track(self, clip, track, **args):
        samples = self.frames_for_prediction(clip, track, **args)
        frames, preprocessed, masses = self.preprocess(clip, track, samples, **args)
        if preprocessed is None or len(preprocessed) == 0:
            return None, None, None
        return frames, self.predict(preprocessed), masses

    def predict_recent_frames(self, clip, track, **args):
        samples = self.frames_for_prediction(clip, track, **args)
        frames, preprocessed, mass = self.preprocess(clip, track, samples, **args)
        if preprocessed is None or len(preprocessed) == 0:
            return None
        return self.predict(preprocessed), frames, mass

    def frames_for_prediction(self, clip, track, **args):
        """interpreter.py:178-253: segments for frames_per_classify > 1, else the track's usable regions (the last
        num_predictions of them)."""
        if args.get("frames_per_classify", 25) <= 1:
            frames = [r for r in track.bounds_history if not r.blank and r.width > 0 and r.height > 0]
            max_frames = args.get("num_predictions")
            if max_frames is not None and len(frames) >= max_frames:
                frames = frames[-max_frames:]
            return frames
        segment_frames = args.get("segment_frames")
        dont_filter = args.get("dont_filter", False)
        predict_from_last = args.get("predict_from_last")
        regions = track.bounds_history
        start_frame = track.start_frame
        if segment_frames is not None:
            return segments_from_frames(clip.get_id(), track.get_id(), start_frame, regions, segment_frames)
        if predict_from_last is not None:
            if predict_from_last == 0:
                return []
            available = len(regions) if clip.frames_kept() is None else min(len(regions), clip.frames_kept())
            want = min(predict_from_last, available)
            if available > want:
                valid = 0
                take = 0
                for i, r in enumerate(reversed(regions[-available:])):
                    if r.blank:
                        continue
                    valid += 1
                    take = i + 1
                    if valid >= want:
                        break
                want = take
            regions = regions[-want:]
            start_frame = regions[0].frame_number
        segments, _ = get_segments(
            clip.get_id(), track.get_id(), start_frame, regions, segment_width=self.params.square_width**2,
            ffc_frames=[] if dont_filter else clip.ffc_frames, repeats=1, segment_types=self.params.segment_types,
            max_segments=args.get("num_predictions"), dont_filter=dont_filter, min_segments=args.get("min_segments"),
            seed=self.seed)
        return segments

    def get_limits(self, clip, track):
        """(thermal_norm_limits, filtered_norm_limits) of a track, computed on the GPU (interpreter.py:315-363)."""
        _, limits = self._device_preprocess(clip, track, [])
        return None, (limits["filt_min"][0], limits["filt_max"][0])

    # ---- device pre-processing (interpreter.py:365-474) ----
    def preprocess_segments(self, clip, track, segments, predict_from_last=None):
        if not segments:
            return [], None, []
        if self.params.mvm:
            raise NotImplementedError("mvm models (RandomForest track features beside the images) are not part of this build")
        sq = self.params.square_width
        n_tiles = sq * sq
        for seg in segments:
            if len(seg.frame_indices) != n_tiles:
                raise ValueError("segments must hold %d frames" % n_tiles)
        x, _ = self._device_preprocess(clip, track, segments)
        return [s.frame_indices for s in segments], x, [s.mass for s in segments]

    def preprocess_frames(self, clip, track, samples):
        """Single-frame models (interpreter.py:255-313, ml_tools/preprocess.py:119-144): every usable region is one
        sample [frame_size, frame_size, channels]: the crop / resize / normalise of preprocess_frame with its default
        clip_thermals_at_zero = True, no tiling.  -> (frame numbers, device tensor [n, fs, fs, 2], [mass of the LAST
        region]) -- the reference returns that one mass (sic)."""
        samples = list(samples)
        if not samples:
            return [], None, []

        class _One:  # a one-frame "segment"
            def __init__(self, region):
                self.frame_indices = [region.frame_number]
                self.mass = region.mass

        x, _ = self._device_preprocess(clip, track, [_One(r) for r in samples], single=True)
        return [r.frame_number for r in samples], x, [samples[-1].mass]

    def limits_flags(self, single=False):
        """_lib.LIMITS_* of this model's hyper-parameters (include/cpx.h)."""
        from .._lib import (LIMITS_ALWAYS_CLIP, LIMITS_NO_DIFF_NORM, LIMITS_SWAP_CHANNELS, LIMITS_TF_SCALING,
                            LIMITS_THERMAL_DIFF_NORM)

        channels = [str(getattr(c, "name", c)) for c in self.params.channels]
        if channels not in (["thermal", "filtered"], ["filtered", "thermal"]):
            raise NotImplementedError("channels %s: the network kernels take the two channels thermal and filtered "
                                      "(in either order)" % (channels,))
        flags = 0
        if self.params.thermal_diff_norm:
            flags |= LIMITS_THERMAL_DIFF_NORM
        if not self.params.diff_norm:
            flags |= LIMITS_NO_DIFF_NORM
        if single:
            flags |= LIMITS_ALWAYS_CLIP
        if channels[0] == "filtered":
            flags |= LIMITS_SWAP_CHANNELS
        if self.preprocess_fn is not None:
            flags |= LIMITS_TF_SCALING
        return flags

    def _device_preprocess(self, clip, track, segments, single=False):
        state = getattr(clip, "device_state", None)
        if state is None:
            raise RuntimeError("clip was not tracked by cpx.ClipTrackExtractor: no device-resident frames")
        flags = self.limits_flags(single)
        used = set(int(f) for s in segments for f in s.frame_indices)
        by_frame = {}
        refs = []
        for r in track.bounds_history:
            by_frame[r.frame_number] = r
            if r.blank or r.width <= 0 or r.height <= 0:
                continue
            if state.frame_index(r.frame_number) is None:
                continue
            refs.append((state.frame_index(r.frame_number), r.x, r.y, r.width, r.height,
                         1 if r.frame_number in used else 0))
        reqs = []
        for s, seg in enumerate(segments):
            for tile, fn in enumerate(seg.frame_indices):
                fn = int(fn)
                r = by_frame.get(fn)
                if r is None or state.frame_index(fn) is None:
                    raise Exception("Clasifying clip {} track {} can't get frame {}".format(
                        clip.get_id(), track.get_id(), fn))
                reqs.append((state.frame_index(fn), r.x, r.y, r.width, r.height, 0, s, tile))
        x, limits = state.engine.preprocess_segments(
            state.frames_dev, state.track_result, np.array(refs, dtype=REGION_REF_DTYPE),
            np.array([0, len(refs)], np.int32), np.array(reqs, dtype=CROP_REQ_DTYPE), len(segments),
            frame_size=self.params.frame_size, square_width=1 if single else self.params.square_width,
            limits_flags=flags)
        return x, limits


class WRResNetInterpreter(Interpreter):
    """WR-ResNet on the MFMA kernels; model = <name>.npz (Keras-layout weights) or a released <name>.tflite (read in
    pure Python on load: cpx/ml_tools/tflite_reader.py -- what the reference's LiteInterpreter takes,
    interpreter.py:520-560) + the sidecar <name>.json."""

    TYPE = "cpx-hip"

    def __init__(self, model_file, run_over_network=False, load_model=True, engine=None):
        super().__init__(model_file, run_over_network)
        self._engine = engine
        self._net = None
        self._weights = None
        if self.params.model_name != "wr-resnet":
            # the samples of the other families are prepared here (crop / resize / normalise / tile / input scaling);
            # their NETWORKS are not built: they classify through a model server (run_over_network: POST /predict,
            # interpreter.py:53-62), as the reference's Pi does with its TFLite / Keras models
            if not self.run_over_network:
                raise NotImplementedError("model_name %r: only the wr-resnet network runs on the device; other families "
                                          "need run_over_network (a model server)" % self.params.model_name)
            return
        if load_model:
            self.load_model()

    def load_model(self):
        from .wrresnet import load_weights

        from .wrresnet import head_of

        if self.model_file.suffix == ".tflite":
            from .tflite_reader import load_tflite

            logging.info("Reading TFLite model %s", self.model_file)
            self._weights = load_tflite(self.model_file)
        else:
            self._weights = load_weights(self.model_file.with_suffix(".npz"))
        n = self._weights["prediction/bias"].shape[0]
        if n != len(self.labels):
            raise ValueError("model has %d outputs but %d labels" % (n, len(self.labels)))
        hidden, act = head_of(self._weights)
        sizes = [int(self._weights[h + "/bias"].shape[0]) for h in hidden]
        want_sizes = [int(v) for v in (self.params.dense_sizes or [])]
        want_act = "sigmoid" if self.params.get("multi_label", True) else "softmax"
        explicit = "prediction/activation" in self._weights
        if sizes != want_sizes or (explicit and act != want_act):
            raise ValueError("weights head (dense %s, %s) does not match the sidecar (dense_sizes %s, multi_label %s)"
                             % (sizes, act, want_sizes, self.params.get("multi_label", True)))
        if not explicit:
            self._weights["prediction/activation"] = want_act  # an archive without the record: the sidecar decides

    def _network(self, engine):
        from .wrresnet import WRResNetDevice

        # one device network per engine (= per handle / HIP stream): callers that alternate between engines -- the bulk
        # path's device lanes -- must not rebuild it at every switch
        nets = self.__dict__.setdefault("_nets", {})
        net = nets.get(id(engine))
        if net is None or net.eng is not engine or not engine.h:
            net = nets[id(engine)] = WRResNetDevice(engine, self._weights, len(self.labels))
        self._net = net
        return net

    def shape(self):
        return 1, (None,) + tuple(self.params.output_dim)

    def predict(self, frames):
        """frames: device tensor (from preprocess_segments) or host float32 [N,H,W,2] -> numpy [N, n_labels]."""
        import torch

        from ..track.cliptrackextractor import default_engine

        if self.run_over_network:  # the model lives in another process (cpx/servemodel.py), kerasmodel.py:856-859
            host = frames.cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames, dtype=np.float32)
            return self.predict_over_network(host)
        if isinstance(frames, torch.Tensor) and frames.is_cuda:
            engine = self._engine or default_engine(frames.device.index or 0)
            x = frames
        else:
            engine = self._engine or default_engine(0)
            x = torch.from_numpy(np.array(frames, dtype=np.float32, copy=True)).to(engine.device)
        _, probs = self._network(engine).forward(x.contiguous())
        return probs.cpu().numpy()


def inc3_preprocess(x):
    """interpreter.py:563-566 (= tf.keras.applications' 'tf' mode), in place on a float32 array."""
    x /= 127.5
    x -= 1.0
    return x


def get_interpreter(model, run_over_network=False, load_model=True, seed=None):
    """Factory with the reference's signature (interpreter.py:597-628)."""
    suffix = Path(model.model_file).suffix
    if suffix in (".keras", ".h5", ".pb", ".sav"):
        # interpreter.py:597-628 hands these to TensorFlow (KerasModel) or joblib (ForestModel): neither runtime is part
        # of this build -- a Keras WR-ResNet converts once, where TensorFlow is installed
        raise NotImplementedError(
            "%s: convert the Keras model once with `python tools/keras_to_npz.py %s <name>` (runs where TensorFlow is "
            "installed) and pass <name>.npz; a released .tflite model is read directly" % (model.model_file, model.model_file))
    if model.type not in (None, WRResNetInterpreter.TYPE, "tflite", "keras") or suffix not in (".npz", ".json", "", ".tflite"):
        raise NotImplementedError(
            "model type %r (%s): cpx runs WR-ResNet models as <name>.npz + <name>.json or as a released <name>.tflite "
            "(TensorFlow / RandomForest runtimes are not part of this build)" % (model.type, model.model_file))
    classifier = WRResNetInterpreter(model.model_file, run_over_network, load_model)
    classifier.id = model.id
    classifier.port = model.port
    if seed is not None:
        classifier.seed = seed
    return classifier
