"""Command line of the classifier -- the reference's flags, one for one (src/classify/main.py:28-142):

    python -m cpx.classify.main [-p PREVIEW_TYPE] [-v] [-c CONFIG_FILE] [-o] [-T] [-m MODEL_FILE] [-w MODEL_WEIGHTS]
                                [-t] [--calculate-thumbnails] [--reuse-prediction-frames] [--cache [BOOL]]
                                [--post-process] source

-v, -o and --reuse-prediction-frames count, --cache takes an optional boolean word, -t is --track, -w names the weights
file of -m (tests/golden/cli_golden.json holds the reference parser's option table; tests/test_cli_cpu.py checks this
one against it).  At this snapshot the reference's main() calls post_process_file(source, None) -- "testing stuff",
main.py:127-137 -- with the process() call commented out; here process() is what runs, and --post-process (the one flag
the reference does not have) selects the other.  Options this build cannot honour parse and then say so where they are
reached: a preview type other than "none" raises NotImplementedError in ClipClassifier, --cache true in the extractor.
"""

import argparse
import logging
import time

from ..config import Config
from ..config.config import ModelConfig
from ..extract import init_logging, str2bool
from .clipclassifier import ClipClassifier


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument(
        "source",
        help='a CPTV file to process, or a folder name, or "all" for all files within subdirectories of source folder.')
    parser.add_argument("-p", "--preview-type",
                        help="Create MP4 previews of this type (can be slow), this overrides the config")
    parser.add_argument("-v", "--verbose", action="count", help="Display additional information.")
    parser.add_argument("-c", "--config-file", help="Path to config file to use")
    parser.add_argument("-o", "--meta-to-stdout", action="count",
                        help="Print metadata to stdout instead of saving to file.")
    parser.add_argument("-T", "--timestamps", action="store_true", help="Emit log timestamps")
    parser.add_argument("-m", "--model-file", help="Path to model file to use, will override config model")
    parser.add_argument("-w", "--model-weights", help="Path to models file to use, will override config model")
    parser.add_argument("-t", "--track", action="store_true", help="Run tracking on the file before extracting")
    parser.add_argument("--calculate-thumbnails", action="store_true", help="Calculate thumbnails")
    parser.add_argument("--reuse-prediction-frames", action="count",
                        help="Use supplied prediction frames from metadata.txt")
    parser.add_argument("--cache", type=str2bool, nargs="?", const=True, default=None,
                        help="Dont keep video frames in memory for classification later, but cache them to disk "
                             "(not built here: frames stay on the device; --cache true raises NotImplementedError)")
    parser.add_argument("--post-process", action="store_true",
                        help="run ClipClassifier.post_process_file on the source file (what the reference's main() "
                             "calls at this snapshot, classify/main.py:129) instead of process()")
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def main(cmd_args=None):
    args = parse_args(cmd_args)
    config = Config.load_from_file(args.config_file)
    config.validate()
    init_logging(args.timestamps)
    if args.preview_type:
        config.classify.preview = args.preview_type
    if args.verbose:
        config.verbose = True
    if args.meta_to_stdout:
        config.classify.meta_to_stdout = True
    model = None
    if args.model_file:
        model = ModelConfig.load({"id": 0, "model_file": args.model_file, "name": args.model_file,
                                  "model_weights": args.model_weights})
        model.validate()
    clip_classifier = ClipClassifier(config, model)
    start = time.time()
    if args.post_process:
        clip_classifier.post_process_file(args.source, None)
    else:
        clip_classifier.process(args.source, cache=args.cache, reuse_frames=args.reuse_prediction_frames,
                                track=args.track, calculate_thumbnails=args.calculate_thumbnails)
    logging.info("Took %s", time.time() - start)


if __name__ == "__main__":
    main()
