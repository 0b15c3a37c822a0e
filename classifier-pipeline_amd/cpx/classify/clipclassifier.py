"""ClipClassifier -- file-level driver of track + classify
(reference src/classify/clipclassifier.py:20-383).  The low-memory re-read path
(post_process_file) and previews are not part of this build."""

import json
import logging
import os
import time
from pathlib import Path

import numpy as np

from ..ml_tools.interpreter import get_interpreter
from ..ml_tools.tools import CustomJSONEncoder, load_clip_metadata
from ..track.clip import Clip
from ..track.cliptrackextractor import ClipTrackExtractor
from ..sharding import rank_world, shard_files
from ..track.trackextractor import extract_file, extract_files
from .thumbnail import best_trackless_thumb, get_thumbnail_info
from .trackprediction import Predictions


class ClipClassifier:
    FRAME_SKIP = 1

    def __init__(self, config, model=None, keep_original_predictions=False, tracking_events=False,
                 model_by_country=True):
        self.keep_original_predictions = keep_original_predictions
        self.batch_files = 64  # recordings per device batch of process(directory, track=True)
        self.config = config
        self.model = model
        self.model_by_country = model_by_country
        if self.keep_original_predictions:
            self.model.id = f"post-{self.model.id}"
            self.model.name = f"post-{self.model.name}"
        if config.classify.preview and config.classify.preview.lower() != "none":
            raise NotImplementedError("previews (MP4 export) are outside the cpx hot path")
        self.previewer = None
        self.models = {}
        self.tracking_events = tracking_events

    def get_classifier(self, model, location=None):
        """Classifier cached per model id in this process (clipclassifier.py:60-83)."""
        if model.id in self.models:
            return self.models[model.id]
        start = time.time()
        classifier = get_interpreter(model, model.run_over_network)
        logging.info("classifier loaded (%s)", time.time() - start)
        self.models[model.id] = classifier
        return classifier

    def process(self, source, cache=None, reuse_frames=None, track=False, calculate_thumbnails=False):
        if not os.path.exists(source):
            logging.error("Could not find file or directory %s", source)
            return
        if os.path.isfile(source):
            self.process_file(source, cache=cache, reuse_frames=reuse_frames, track=track,
                              calculate_thumbnails=calculate_thumbnails)
            return
        todo = []
        for folder, _, files in os.walk(source):
            for name in sorted(files):
                if os.path.splitext(name)[1] == ".cptv":
                    todo.append(os.path.join(folder, name))
        if not track:
            for filename in todo:
                self.process_file(filename, cache=cache, reuse_frames=reuse_frames, track=False,
                                  calculate_thumbnails=calculate_thumbnails)
            return
        rank, world, local_rank = rank_world()  # under torchrun: this rank's share of the files, on its own GPU
        todo = shard_files(todo, rank, world)
        for i in range(0, len(todo), self.batch_files):
            self.process_files(todo[i:i + self.batch_files], reuse_frames=reuse_frames,
                               calculate_thumbnails=calculate_thumbnails, device=local_rank if world > 1 else 0)

    def process_files(self, filenames, reuse_frames=None, calculate_thumbnails=False, device=0):
        """process_file(track=True) for a list of recordings whose decode / tracking / association run as one device
        batch (trackextractor.extract_files); classification and metadata per file as in process_file."""
        results = []
        tracked = extract_files(filenames, self.config, False, to_stdout=False, save_meta=False, device=device)
        models = [self.model] if self.model else (self.config.classify.models or [])
        for filename, (clip, track_extractor, meta_data) in zip(filenames, tracked):
            meta_file = os.path.splitext(str(filename))[0] + ".txt"
            predictions_per_model = {}
            for model in models:
                predictions_per_model[model.id] = self.classify_clip(clip, model, meta_data, reuse_frames=reuse_frames)
            results.append(self.save_metadata(meta_data, meta_file, clip, predictions_per_model, models,
                                              calculate_thumbnails=calculate_thumbnails))
        return results

    def process_file(self, filename, cache=None, reuse_frames=None, track=False, calculate_thumbnails=False):
        """Track (optionally) and classify one recording; writes / returns the metadata (clipclassifier.py:145-250)."""
        filename = str(filename)
        _, ext = os.path.splitext(filename)
        if ext != ".cptv":
            logging.error("Unknown extention %s", ext)
            return False
        if not os.path.exists(filename):
            logging.error("File %s not found.", filename)
            return False
        meta_file = os.path.splitext(filename)[0] + ".txt"
        meta_data = None
        if track:
            clip, track_extractor, meta_data = extract_file(filename, self.config, False, to_stdout=False,
                                                            save_meta=False)
        else:
            if not os.path.exists(meta_file):
                logging.error("File %s not found.", meta_file)
                return False
            meta_data = load_clip_metadata(meta_file)
            track_extractor = ClipTrackExtractor(self.config.tracking, self.config.use_opt_flow, False,
                                                 do_tracking=False, calculate_filtered=True,
                                                 verbose=self.config.verbose)
            clip = Clip(track_extractor.config, filename)
            clip.load_metadata(meta_data)
            track_extractor.parse_clip(clip)
        predictions_per_model = {}
        models = [self.model] if self.model else (self.config.classify.models or [])
        for model in models:
            predictions_per_model[model.id] = self.classify_clip(clip, model, meta_data, reuse_frames=reuse_frames)
        return self.save_metadata(meta_data, meta_file, clip, predictions_per_model, models,
                                  calculate_thumbnails=calculate_thumbnails)

    def classify_clip(self, clip, model, meta_data, reuse_frames=None):
        start = time.time()
        classifier = self.get_classifier(model, meta_data.get("location"))
        predictions = Predictions(classifier.labels, model, classifier.thresholds)
        predictions.model_load_time = time.time() - start
        for i, track in enumerate(clip.tracks):
            segment_frames = None
            if reuse_frames:
                meta_track = next((x for x in meta_data.get("tracks") or [] if x["id"] == track.get_id()), None)
                if meta_track is not None:
                    tag = next((x for x in meta_track.get("tags", []) if x.get("data", {}).get("name") == model.name), None)
                    if tag is not None and "prediction_frames" in tag["data"]:
                        segment_frames = np.uint16(tag["data"]["prediction_frames"])
            prediction = classifier.classify_track(clip, track, segment_frames=segment_frames, min_segments=1)
            if prediction is not None:
                predictions.prediction_per_track[track.get_id()] = prediction
                logging.info("%s - [%s/%s] prediction: %s", track.get_id(), i + 1, len(clip.tracks),
                             prediction.description())
        return predictions

    def save_metadata(self, meta_data, meta_filename, clip, predictions_per_model, models, calculate_thumbnails=False):
        tracks = meta_data.get("tracks")
        for track in clip.tracks:
            meta_track = next((x for x in tracks if x["id"] == track.get_id()), None)
            if meta_track is None:
                logging.error("Got prediction for track which doesn't exist in metadata")
                continue
            info = []
            for model_id, predictions in predictions_per_model.items():
                prediction = predictions.prediction_for(track.get_id())
                if prediction is None:
                    continue
                pm = prediction.get_metadata(predictions.thresholds)
                pm["model_id"] = model_id
                if self.keep_original_predictions:
                    pm["reprocessed"] = True
                info.append(pm)
            if self.keep_original_predictions:
                info.extend(meta_track.get("predictions") or [])
            meta_track["predictions"] = info
            if calculate_thumbnails:
                best_thumb, best_score = get_thumbnail_info(clip, track)
                if best_thumb is None:
                    meta_track["thumbnail"] = None
                else:
                    meta_track["thumbnail"] = {
                        "region": best_thumb.region,
                        "contours": best_thumb.contours,
                        "median_diff": best_thumb.median_diff,
                        "score": round(best_score),
                    }
        if calculate_thumbnails and len(clip.tracks) == 0:
            meta_data["thumbnail_region"] = best_trackless_thumb(clip)  # if no tracks choose a clip thumb
        by_id = {m["id"]: m for m in meta_data.get("models", [])}
        for model in models:
            d = by_id.get(model.id) or model.as_dict()
            mp = predictions_per_model[model.id]
            d["classify_time"] = float(round(mp.classify_time + mp.model_load_time, 1))
            by_id[model.id] = d
        meta_data["models"] = list(by_id.values())
        if self.config.classify.meta_to_stdout:
            print(json.dumps(meta_data, cls=CustomJSONEncoder))
        else:
            logging.info("saving meta data %s", meta_filename)
            with open(meta_filename, "w") as fh:
                json.dump(meta_data, fh, indent=4, cls=CustomJSONEncoder)
        return meta_data

    def post_process_file(self, filename, service):
        raise NotImplementedError("post_process_file (low-memory re-read path) is not part of this build: "
                                  "use process_file(track=True)")
