"""TEST INFRASTRUCTURE ONLY -- imports the *unmodified* Python reference from
/root/reference/src in the build container, with harness-side stand-ins for its
un-installed third-party wheels, so that the reference itself can (1) validate
the restatements under oracle/ and (2) generate the fixtures in tests/golden.

Never importable on the GPU box (``/root/reference`` is absent there) and never
imported by the product package.  Stand-ins:
  cv2                      -> oracle/cv2_shim.py (NumPy restatement of OpenCV)
  cptv_rs_python_bindings  -> the build's own CPTV decoder (cpx.cptv.CptvReader)
  timezonefinder, h5py, toml, astral(+.sun), portalocker, inotify_simple
                           -> empty modules (imported, never used on the path)
"""

import importlib
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"
REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)


def reference_available():
    return os.path.isdir(REFERENCE_SRC)


def _empty_module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Put the stand-ins in sys.modules and the reference on sys.path."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_SRC)
    pkg_dir = os.path.join(_REPO, "classifier-pipeline_amd")
    if pkg_dir not in sys.path:
        sys.path.insert(0, pkg_dir)
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    import cv2_shim

    sys.modules["cv2"] = cv2_shim
    from cpx.cptv import CptvReader

    _empty_module("cptv_rs_python_bindings", CptvReader=CptvReader)

    class _TZF:
        def certain_timezone_at(self, lat=None, lng=None):
            return None

    _empty_module("timezonefinder", TimezoneFinder=_TZF)
    _empty_module("h5py")
    _empty_module("toml", load=lambda *a, **k: {}, loads=lambda *a, **k: {})
    astral = _empty_module("astral")
    astral.LocationInfo = object
    _empty_module("astral.sun", sun=lambda *a, **k: {})
    _empty_module("portalocker")
    _empty_module("inotify_simple", INotify=object, flags=object)
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    _installed = True


def ref(module):
    """Import a reference module by its in-reference dotted name."""
    install()
    return importlib.import_module(module)


def default_config():
    return ref("config.config").Config.get_defaults()


def run_tracking(cptv_path, denoise=None, hooks=None):
    """Run the reference ClipTrackExtractor.parse_clip on a CPTV file.

    hooks: optional dict of callables
       'frame'(clip, extractor, frame_index) called after every process_frame.
    Returns (clip, extractor).
    """
    install()
    cfg = default_config()
    if denoise is not None:
        cfg.tracking["thermal"].denoise = denoise
    cte = ref("track.cliptrackextractor")
    clipmod = ref("track.clip")
    extractor = cte.ClipTrackExtractor(cfg.tracking, cfg.use_opt_flow, False)
    clip = clipmod.Clip(cfg.tracking["thermal"], str(cptv_path))
    if hooks and "frame" in hooks:
        orig = extractor.process_frame

        def wrapped(c, f):
            r = orig(c, f)
            hooks["frame"](c, extractor, c.current_frame)
            return r

        extractor.process_frame = wrapped
    extractor.parse_clip(clip)
    return clip, extractor
