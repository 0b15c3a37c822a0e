"""ClipClassifier -- file-level driver of track + classify
(reference src/classify/clipclassifier.py:20-651), incl. the re-read path post_process_file that classify.py runs
today (classify/main.py:129).  Previews are not part of this build."""

import json
import logging
import math
import os
import time
from datetime import datetime
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


# config/buildconfig.py:48-55 COUNTRY_LOCATIONS = Rectangle.from_ltrb(left, top, right, bottom) in degrees
COUNTRY_LOCATIONS = {
    "AU": (113.338953078, -10.6681857235, 153.569469029, -43.6345972634),
    "NZ": (166.509144322, -34.4506617165, 178.517093541, -46.641235447),
}


def country_by_location(lat, lng):
    """clipclassifier.py:654-660: the first country whose box contains the point -- Rectangle.contains(lng, lat)
    (ml_tools/rectangle.py:148-150) on a rectangle built by from_ltrb, i.e. right = left + (right - left) and
    bottom = top + (bottom - top) in floating point, as the reference evaluates them."""
    for country, (left, top, right, bottom) in COUNTRY_LOCATIONS.items():
        r = left + (right - left)
        b = top + (bottom - top)
        if left <= lng and r >= lng and top >= lat and b <= lat:
            return country
    return None


class ClipClassifier:
    FRAME_SKIP = 1

    def __init__(self, config, model=None, keep_original_predictions=False, tracking_events=False,
                 model_by_country=True):
        self.keep_original_predictions = keep_original_predictions
        self.batch_files = None  # recordings per decode batch of process(directory, track=True); None: bulk.auto_batch_files
        self.meta_pool_min_files = 2048  # metadata worker processes (bulk.MetaPool) for directories this large; None = never
        self.last_run = None
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
        """Classifier cached per model id in this process (clipclassifier.py:60-83).  With a recording location the
        reference looks for a country-specific model next to the configured one -- <models>/<country>/<file name>,
        the country decided by the bounding boxes of config/buildconfig.py (COUNTRY_LOCATIONS) -- and, like the
        reference, the first model loaded for an id is the one the process keeps.  model_by_country = False
        (ClipClassifier's constructor argument) keeps the configured file."""
        if model.id in self.models:
            return self.models[model.id]
        start = time.time()
        if location is not None and self.model_by_country:
            coordinates = location.get("coordinates") if isinstance(location, dict) else None
            if coordinates is not None:
                country = country_by_location(coordinates[1], coordinates[0])
                if country is not None:
                    model_file = Path(model.model_file)
                    country_model = model_file.parent.parent / country
                    logging.info("Checking if country model exists %s", country_model)
                    if country_model.exists():
                        model.model_file = str(country_model / model_file.name)
                        logging.info("Setting to country model %s", model.model_file)
        logging.info("classifier loading %s", model.model_file)
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
        rank, world, local_rank = rank_world()  # under torchrun: this rank's share of the files, on its own GPU
        todo = shard_files(todo, rank, world)
        if world > 1:  # the reader / staging threads started below stay on the CPUs next to this rank's GPU
            from ..sharding import pin_to_gpu_numa

            logging.info("rank %d: %s", rank, pin_to_gpu_numa(local_rank))
        if not track:
            for filename in todo:
                self.process_file(filename, cache=cache, reuse_frames=reuse_frames, track=False,
                                  calculate_thumbnails=calculate_thumbnails, device=local_rank if world > 1 else 0)
            return
        models = [self.model] if self.model else (self.config.classify.models or [])
        # as in the reference the FIRST recording's location decides the (country) model the process keeps
        # (clipclassifier.py:60-83,254-256): it comes from the recording's existing metadata file
        location = self.first_location(todo)
        classifiers = [self.get_classifier(m, location) for m in models]
        # (a model served over the network -- the families whose networks are not built here, or any run_over_network
        # model -- has no device network for the bulk path's batched forward: its samples go to the server per file)
        per_file = bool(reuse_frames) or any(c.params.square_width == 1 for c in classifiers) \
            or len(set(c.limits_flags() for c in classifiers)) > 1 \
            or any(c.run_over_network or getattr(c, "_weights", None) is None for c in classifiers)
        if per_file:  # saved frames decide the segments / single-frame models / mixed normalisation variants / served models
            for i in range(0, len(todo), 64):
                self.process_files(todo[i:i + 64], reuse_frames=reuse_frames,
                                   calculate_thumbnails=calculate_thumbnails, device=local_rank if world > 1 else 0)
            return
        # the file-fed path at device speed (cpx/track/bulk.py): the recordings of a device batch are decoded, tracked,
        # cut into segments (the reference's get_segments with every random draw the identity: cpx_plan_segments),
        # cropped, classified and written without per-frame Python objects; a recording that fails is retried on its own
        from ..track.bulk import MetaPool, run_files_bulk

        # a large directory: the metadata text is formatted by worker processes (spawned children that never touch the
        # GPU), started while this process has not touched it either: the models above were read, not uploaded
        pool = MetaPool.make() if self.meta_pool_min_files is not None and len(todo) >= self.meta_pool_min_files else None
        try:
            _, tracker = run_files_bulk(todo, self.config, device=local_rank if world > 1 else 0,
                                        batch_files=self.batch_files, clip_classifier=self, meta_pool=pool)
        finally:
            if pool is not None:
                pool.close()
        self.last_run = tracker.timings

    @staticmethod
    def first_location(filenames):
        """`location` of the first recording's existing <file>.txt (None without one or when it cannot be read)."""
        for filename in filenames[:1]:
            meta_file = os.path.splitext(str(filename))[0] + ".txt"
            if os.path.exists(meta_file):
                try:
                    return load_clip_metadata(meta_file).get("location")
                except Exception as e:  # noqa: BLE001 -- the file itself is reported when its turn comes
                    logging.warning("could not read %s for the recording location: %s", meta_file, e)
        return None

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

    def process_file(self, filename, cache=None, reuse_frames=None, track=False, calculate_thumbnails=False, device=0,
                     blob=None):
        """Track (optionally) and classify one recording; writes / returns the metadata (clipclassifier.py:145-250).
        cache (the reference's HDF5 frame cache) has no meaning here -- frames stay on the device -- and is ignored.
        blob (with track=True): the recording's bytes, `filename` naming it (extract_file(blob=...))."""
        filename = str(filename)
        _, ext = os.path.splitext(filename)
        if ext != ".cptv":
            logging.error("Unknown extention %s", ext)
            return False
        if blob is None and not os.path.exists(filename):
            logging.error("File %s not found.", filename)
            return False
        meta_file = os.path.splitext(filename)[0] + ".txt"
        meta_data = None
        if track and blob is not None:
            clip, track_extractor, meta_data = extract_file(filename, self.config, False, to_stdout=False,
                                                            save_meta=False, blob=blob)
        elif track:
            if device:
                clip, track_extractor, meta_data = extract_files([filename], self.config, False, to_stdout=False,
                                                                 save_meta=False, device=device)[0]
            else:
                clip, track_extractor, meta_data = extract_file(filename, self.config, False, to_stdout=False,
                                                                save_meta=False)
        else:
            if not os.path.exists(meta_file):
                logging.error("File %s not found.", meta_file)
                return False
            meta_data = load_clip_metadata(meta_file)
            track_extractor = ClipTrackExtractor(self.config.tracking, self.config.use_opt_flow, False,
                                                 do_tracking=False, calculate_filtered=True,
                                                 verbose=self.config.verbose, device=device)
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
        """clipclassifier.py:385-651: tracks come from <clip>.txt when it exists (else from tracking the file), segments
        are chosen first, then the recording is walked AGAIN with the background model the tracking left behind
        (a fresh one in the metadata case) -- updated from the 45-frame running mean on every frame that is not
        FFC-affected -- and only the sampled regions are cropped: thermal - frame median, filtered = thermal -
        background as it stands at that frame; limits over the sampled crops; the median is subtracted before the
        resize, thermals are always clipped at zero; predictions in chunks of 5 segments.
        On the device the second walk is one cpx_track_batch_ex(KEEP_BACKGROUND | FREEZE_ON_FFC) over the resident
        frames, and the crops are cpx_track_limits_batch_ex(CPX_LIMITS_POST_PROCESS) + cpx_crop_tile."""
        from .._lib import (CROP_REQ_DTYPE, LIMITS_POST_PROCESS, REGION_REF_DTYPE, TRACK_FREEZE_ON_FFC,
                            TRACK_KEEP_BACKGROUND)

        filename = Path(filename)
        meta_file = filename.with_suffix(".txt")
        has_metadata = meta_file.exists()
        if not filename.exists():
            logging.error("File %s not found.", filename)
            return False
        if has_metadata:
            track_extractor = ClipTrackExtractor(self.config.tracking, self.config.use_opt_flow,
                                                 calculate_filtered=True, verbose=self.config.verbose)
            clip = Clip(track_extractor.config, filename)
            meta_data = load_clip_metadata(meta_file)
            rec_end = datetime.fromisoformat(meta_data["end_time"])
            clip.load_metadata(meta_data, getattr(getattr(self.config, "build", None), "tag_precedence", None))
            track_extractor.init_clip(clip)
            frames = track_extractor._frames
            eng = track_extractor._engine
            meta = eng.make_meta(len(frames), [f.time_on for f in frames], [f.last_ffc_time for f in frames],
                                 [bool(f.background_frame) for f in frames])
            flags = TRACK_FREEZE_ON_FFC  # init_clip seeded a fresh model from the file's first frame
        else:
            clip, track_extractor, meta_data = extract_file(filename, self.config, False, max_frames=45, save_meta=False)
            rec_end = datetime.fromisoformat(meta_data["end_time"])
            frames = track_extractor._frames
            eng = track_extractor._engine
            meta = track_extractor._meta
            # the model continues from where tracking left it (the reference re-uses track_extractor.background_alg)
            eng.set_background(0, *track_extractor.final_state())
            flags = TRACK_KEEP_BACKGROUND | TRACK_FREEZE_ON_FFC
        del rec_end  # only the dbus event of the Pi reads it (service.TrackReprocessed)

        logging.info("Just running on first model")
        start = time.time()
        model = self.config.classify.models[0]
        classifier = self.get_classifier(model)
        predictions = Predictions(classifier.labels, model, classifier.thresholds)
        predictions.model_load_time = time.time() - start
        if classifier.params.thermal_diff_norm:
            logging.error("Thermal min diff is not implemented so will not be used")
        # the model's normalisation variant: post_process_file computes limits only for diff_norm models
        # (clipclassifier.py:488) and never thermal ones (:465); without limits preprocess_frame normalises per tile
        from .._lib import LIMITS_THERMAL_DIFF_NORM

        post_flags = LIMITS_POST_PROCESS | (classifier.limits_flags() & ~LIMITS_THERMAL_DIFF_NORM)

        track_data = {}
        for track in clip.tracks:
            track_data[track.get_id()] = {"pred_frames": classifier.frames_for_prediction(clip, track), "track": track}

        # ---- the second walk over the recording, on the device ----
        n = len(frames)
        offs = np.array([0, n], np.int32)
        frames_dev = track_extractor._frames_dev
        res = eng.track_batch(frames_dev, offs, meta, want_filtered=True, flags=flags)
        res.check()
        proc = [i for i in range(n) if not meta["background_frame"][i]]  # frame number -> index in frames_dev

        sq, fs = classifier.params.square_width, classifier.params.frame_size
        for i, (track_id, data) in enumerate(track_data.items()):
            segments = data["pred_frames"]
            if len(segments) == 0:
                logging.info("No prediction made for track %s", track_id)
                continue
            by_frame = {}
            for seg in segments:
                for r in seg.regions:
                    by_frame[int(r.frame_number)] = r
            refs = [(proc[fn], r.x, r.y, r.width, r.height, 1) for fn, r in sorted(by_frame.items())]
            reqs = []
            for s_i, seg in enumerate(segments):
                for tile, fn in enumerate(seg.frame_indices):
                    r = by_frame[int(fn)]
                    reqs.append((proc[int(fn)], r.x, r.y, r.width, r.height, 0, s_i, tile))
            x, _ = eng.preprocess_segments(frames_dev, res, np.array(refs, dtype=REGION_REF_DTYPE),
                                           np.array([0, len(refs)], np.int32), np.array(reqs, dtype=CROP_REQ_DTYPE),
                                           len(segments), frame_size=fs, square_width=sq,
                                           limits_flags=post_flags)
            preds = []
            chunk_size = 5
            for chunk in range(int(math.ceil(len(segments) / chunk_size))):
                part = x[chunk * chunk_size: chunk * chunk_size + chunk_size]
                logging.info("Predicting chunk %s (%s #) of %s total preprocessed %s", chunk, len(part),
                             int(math.ceil(len(segments) / chunk_size)), len(segments))
                try:
                    pred = classifier.predict(part)
                except Exception:
                    logging.error("Could not classify chunk not trying again ", exc_info=True)
                    break
                preds.extend(pred)
            track_prediction = classifier.track_prediction_from_raw(
                track_id, [seg.frame_indices for seg in segments], preds, [seg.mass for seg in segments])
            predictions.prediction_per_track[track_id] = track_prediction
            logging.info("%s - [%s/%s] prediction: %s", track_id, i + 1, len(clip.tracks), track_prediction.description())
            if self.tracking_events and service is not None and len(track_prediction.predictions) > 0:
                # the dbus event of the Pi (clipclassifier.py:612-637); the region it reports is the last one the
                # reference's re-read loop touched (a leaked loop variable there)
                last_fn = max(int(r.frame_number) for d in track_data.values() for sg in d["pred_frames"] for r in sg.regions)
                region = [r for d in track_data.values() for sg in d["pred_frames"] for r in sg.regions
                          if int(r.frame_number) == last_fn][-1]
                scores = np.uint8(np.round(track_prediction.class_best_score.copy() * 100)).tolist()
                service.TrackReprocessed(
                    meta_data.get("id", 0), track_id, scores, track_prediction.predicted_tag(),
                    int(round(100 * track_prediction.max_score)), np.uint8(region.to_ltrb()).tolist(),
                    region.frame_number, int(region.mass), region.blank, True,
                    data["track"].bounds_history[-1].frame_number, model.id,
                    datetime.fromisoformat(meta_data["end_time"]).timestamp())
        return self.save_metadata(meta_data, meta_file, clip, {model.id: predictions}, [model], calculate_thumbnails=False)
