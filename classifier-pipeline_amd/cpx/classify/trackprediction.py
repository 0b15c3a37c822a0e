"""Per-track prediction record and its aggregation over segments
(reference src/classify/trackprediction.py:14-507)."""

import time

import numpy as np

DEFAULT_THRESHOLD = 0.8


class Predictions:
    def __init__(self, labels, model, thresholds):
        self.labels = labels
        self.prediction_per_track = {}
        self.model = model
        self.model_load_time = None
        self.thresholds = thresholds

    def prediction_for(self, track_id):
        return self.prediction_per_track.get(track_id)

    def clear_predictions(self):
        self.prediction_per_track = {}

    def guesses_for(self, track_id):
        p = self.prediction_per_track.get(track_id)
        return p.guesses() if p else []

    def prediction_description(self, track_id):
        return self.prediction_per_track.get(track_id).description()

    @property
    def classify_time(self):
        return np.sum([p.classify_time for p in self.prediction_per_track.values() if p.classify_time is not None])


class Prediction:
    def __init__(self, prediction, smoothed_prediction, frames, predicted_at_frame, mass):
        self.prediction = prediction
        self.smoothed_prediction = smoothed_prediction
        self.frames = frames
        self.predicted_at_frame = predicted_at_frame
        self.mass = mass
        self.predicted_time = time.time()

    def get_metadata(self):
        # (plain lists / ints: the same JSON as the NumPy arrays the reference puts here, without a trip through the
        # encoder's default() hook per element)
        if np.ndim(self.frames) == 0:  # single-frame models: one frame number per prediction
            frames = int(self.frames)
        else:
            frames = self.frames.tolist() if isinstance(self.frames, np.ndarray) else [int(f) for f in self.frames]
        meta = {"prediction": np.uint8(np.round(100 * self.prediction)).tolist(),
                "smoothed_prediction": None if self.smoothed_prediction is None
                else np.uint32(np.round(self.smoothed_prediction)).tolist(),
                "frames": frames, "predicted_at_frame": int(self.predicted_at_frame),
                # smoothed models carry the mass as a 1-element array (classified_track: masses[:, None]) and the
                # reference serialises it as a list (trackprediction.py:56, CustomJSONEncoder)
                "mass": self.mass.tolist() if isinstance(self.mass, np.ndarray) and self.mass.ndim > 0 else int(self.mass),
                "predicted_time": self.predicted_time}
        return meta

    def clarity(self):
        best = np.argsort(self.prediction)
        return self.prediction[best[-1]] - self.prediction[best[-2]]


class TrackPrediction:
    def __init__(self, track_id, labels, keep_all=True, start_frame=None, smooth_preds=False):
        self.fp_index = labels.index("false-positive") if "false-positive" in labels else None
        self.track_id = track_id
        self.predictions = []
        self.class_best_score = np.zeros(len(labels))
        self.start_frame = start_frame
        self.last_frame_classified = None
        self.num_frames_classified = 0
        self.keep_all = keep_all
        self.labels = labels
        self.classify_time = None
        self.tracking = False
        self.masses = []
        self.normalized = False
        self.smooth_preds = smooth_preds

    def cap_confidences(self, max_confidence):
        total = np.sum(self.class_best_score)
        if total > max_confidence:
            self.class_best_score *= max_confidence / total

    def classified_track(self, predictions, prediction_frames, masses):
        """Sum (or mass-weighted sum) of the segment predictions, normalised (trackprediction.py:127-171)."""
        top_score = None
        smoothed = None
        predictions = np.asarray(predictions)
        if self.smooth_preds:
            masses = np.array(masses)
            top_score = np.sum(masses)
            masses = masses[:, None]
            smoothed = predictions * masses
        self.num_frames_classified = len(predictions)
        for i, (pred, frames, mass) in enumerate(zip(predictions, prediction_frames, masses)):
            self.predictions.append(Prediction(pred, None if smoothed is None else smoothed[i], frames,
                                               np.amax(frames), mass))
        if self.num_frames_classified > 0:
            if smoothed is None:
                score = np.sum(predictions, axis=0)
                self.class_best_score = score / np.sum(score)
            else:
                self.class_best_score = np.sum(smoothed, axis=0) / top_score

    # ---- read-outs ----
    @property
    def best_label_index(self):
        return None if self.class_best_score is None else np.argmax(self.class_best_score)

    @property
    def max_score(self):
        return None if self.class_best_score is None else float(np.amax(self.class_best_score))

    def predicted_tag(self):
        i = self.best_label_index
        return None if i is None else self.labels[i]

    def score(self, n=None):
        if n is None:
            return self.max_score
        return None if self.class_best_score is None else float(sorted(self.class_best_score)[-n])

    def label_index(self, n=None):
        if n is None:
            return self.best_label_index
        return None if self.class_best_score is None else int(np.argsort(self.class_best_score)[-n])

    @property
    def clarity(self):
        if self.class_best_score is None or len(self.class_best_score) < 2:
            return None
        return self.max_score - self.score(2)

    @property
    def num_frames(self):
        return self.num_frames_classified

    def class_confidences(self):
        return {} if self.class_best_score is None else {
            self.labels[i]: round(float(v), 3) for i, v in enumerate(self.class_best_score)}

    def guesses(self):
        return ["{} ({:.1f})".format(self.labels[self.label_index(i)], self.score(i) * 10)
                for i in range(1, min(len(self.labels), 4)) if self.score(i) and self.score(i) > 0.5]

    def description(self):
        score = self.max_score
        if score is None:
            return None
        first = "{} {:.1f} (clarity {:.1f})".format(self.labels[self.best_label_index], score * 10, self.clarity * 10)
        if score <= 0.5:
            first = "[nothing] " + first
        second = ""
        if self.score(2) > 0.5:
            second = "[second guess - {} {:.1f}]".format(self.labels[self.label_index(2)], self.score(2) * 10)
        return (first + " " + second).strip()

    def get_metadata(self, thresholds):
        meta = {}
        if self.classify_time is not None:
            meta["classify_time"] = round(self.classify_time, 1)
        meta["tag"] = self.predicted_tag()
        confidence = self.max_score if self.max_score else 0
        threshold = thresholds[self.best_label_index] if thresholds is not None else DEFAULT_THRESHOLD
        meta["threshold_used"] = threshold
        meta["confident"] = confidence >= threshold
        meta["confidence"] = round(confidence, 2)
        meta["clarity"] = round(self.clarity, 3) if self.clarity else 0
        meta["all_class_confidences"] = {}
        meta["predictions"] = [p.get_metadata() for p in self.predictions]
        if self.class_best_score is not None:
            # round(numpy float64, 3) per label is NumPy's rounding: done on the whole vector at once
            meta["all_class_confidences"] = dict(zip(self.labels, np.round(np.asarray(self.class_best_score), 3).tolist()))
        return meta
