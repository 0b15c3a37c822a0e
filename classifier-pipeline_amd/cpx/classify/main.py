"""Command line of the classifier (reference src/classify/main.py:14-142):
    python -m cpx.classify.main [-c CONFIG] [-m MODEL_FILE] [--track] [--reuse-prediction-frames]
                                [--calculate-thumbnails] [--post-process] [-o] source
"""

import argparse
import logging
import sys

from ..config import Config
from ..config.config import ModelConfig
from .clipclassifier import ClipClassifier


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help="a CPTV file to process, or a folder name")
    ap.add_argument("-c", "--config-file", help="Path to config file to use")
    ap.add_argument("-m", "--model-file", help="Path to model file to use, will override config model")
    ap.add_argument("-T", "--timestamps", action="store_true", help="Emit log timestamps")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--track", action="store_true", help="Track the clip before classifying")
    ap.add_argument("--calculate-thumbnails", action="store_true",
                    help="Calculate thumbnail regions for each track and save in metadata")
    ap.add_argument("--reuse-prediction-frames", action="count",
                    help="Use the prediction frames saved in the metadata")
    ap.add_argument("-o", "--meta-to-stdout", action="store_true", help="Print metadata to stdout instead of a file")
    ap.add_argument("--post-process", action="store_true",
                    help="run ClipClassifier.post_process_file on the source file (what the reference's classify.py "
                         "currently calls, classify/main.py:129) instead of process()")
    ap.add_argument("--cache", default=None,
                    help="accepted for compatibility: the reference's disk cache of frames; frames stay on the device here")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    fmt = "%(process)d %(thread)s:%(levelname)7s %(message)s"
    if args.timestamps:
        fmt = "%(asctime)s " + fmt
    logging.basicConfig(stream=sys.stderr, level=logging.INFO, format=fmt, datefmt="%Y-%m-%d %H:%M:%S")
    config = Config.load_from_file(args.config_file)
    if args.verbose:
        config.verbose = True
    if args.meta_to_stdout:
        config.classify.meta_to_stdout = True
    model = None
    if args.model_file:
        model = ModelConfig.load({"id": 1, "model_file": args.model_file, "name": args.model_file})
    if args.post_process:
        ClipClassifier(config, model).post_process_file(args.source, None)
        return
    ClipClassifier(config, model).process(args.source, cache=args.cache, reuse_frames=args.reuse_prediction_frames,
                                          track=args.track, calculate_thumbnails=args.calculate_thumbnails)


if __name__ == "__main__":
    main()
