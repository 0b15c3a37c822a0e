"""Command line of the track extractor (reference src/extract.py:25-89):
    python -m cpx.extract [-c CONFIG] [-o OUTPUT] [--cache] [-T] [-v] [-r] source
"""

import argparse
import logging
import sys

from .config import Config
from .track.trackextractor import TrackExtractor


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help='a CPTV file to process, or a folder name')
    ap.add_argument("-c", "--config-file", help="Path to config file to use")
    ap.add_argument("--cache", type=lambda s: s.lower() in ("1", "true", "yes"), default=None,
                    help="(unsupported) cache frames to disk")
    ap.add_argument("-T", "--timestamps", action="store_true", help="Emit log timestamps")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("-r", "--retrack", action="store_true", help="Track again using existing metadata")
    ap.add_argument("--to-stdout", action="store_true", help="Print metadata JSON instead of writing <clip>.txt")
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
    extractor = TrackExtractor(config, cache_to_disk=args.cache or False, retrack=args.retrack)
    extractor.extract(args.source, to_stdout=args.to_stdout)


if __name__ == "__main__":
    main()
