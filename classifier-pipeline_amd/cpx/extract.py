"""Command line of the track extractor -- the reference's flags, one for one (src/extract.py:25-89):

    python -m cpx.extract [-p PREVIEW_TYPE] [-v] [-o] [-c CONFIG_FILE] [-T] [--retrack [BOOL]] [--cache [BOOL]] source

-v and -o count, --retrack / --cache take an optional boolean word (yes/true/t/y/1, no/false/f/n/0), exactly as there
(tests/golden/cli_golden.json holds the reference parser's option table; tests/test_cli_cpu.py checks this one against
it).  What the flags reach that this build does not have says so when it is reached, not at parse time: --cache true
(the disk cache of frames) and a preview type other than "none" raise NotImplementedError from the classes that would
have honoured them.  One deliberate difference: the reference reads config.classify.meta_to_stdout BEFORE it applies -o
(extract.py:80-83), so its -o never reaches extract(); here -o prints the metadata to stdout, as its help says.
"""

import argparse
import logging
import sys

from .config import Config
from .track.trackextractor import TrackExtractor


def str2bool(v):
    """The reference's boolean words (extract.py:14-22)."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    elif v.lower() in ("no", "false", "f", "n", "0"):
        return False
    else:
        raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument(
        "source",
        help='a CPTV file to process, or a folder name, or "all" for all files within subdirectories of source folder.')
    parser.add_argument("-p", "--preview-type",
                        help="Create MP4 previews of this type (can be slow), this overrides the config")
    parser.add_argument("-v", "--verbose", action="count", help="Display additional information.")
    parser.add_argument("-o", "--meta-to-stdout", action="count",
                        help="Print metadata to stdout instead of saving to file.")
    parser.add_argument("-c", "--config-file", help="Path to config file to use")
    parser.add_argument("-T", "--timestamps", action="store_true", help="Emit log timestamps")
    parser.add_argument("--retrack", type=str2bool, nargs="?", const=True, default=None,
                        help="Use existing metadata to correct tracks")
    parser.add_argument("--cache", type=str2bool, nargs="?", const=True, default=None,
                        help="Dont keep video frames in memory for classification later, but cache them to disk "
                             "(not built here: frames stay on the device; --cache true raises NotImplementedError)")
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def init_logging(timestamps=False):
    fmt = "%(process)d %(thread)s:%(levelname)7s %(message)s"
    if timestamps:
        fmt = "%(asctime)s " + fmt
    logging.basicConfig(stream=sys.stderr, level=logging.INFO, format=fmt, datefmt="%Y-%m-%d %H:%M:%S")


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
    extractor = TrackExtractor(config, cache_to_disk=args.cache, retrack=bool(args.retrack))
    extractor.extract(args.source, to_stdout=bool(config.classify.meta_to_stdout))


if __name__ == "__main__":
    main()
