"""CPU-side checks of the boundary: the library loads and exports every symbol
include/cpx.h declares; struct layouts agree; host logic (config, CPTV decode,
geometry) behaves like the reference's.  No compute calls (no GPU here)."""
import ctypes as C
import io
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from cpx import _lib

    hdr = open(os.path.join(REPO, "include", "cpx.h")).read()
    declared = set(re.findall(r"^\s*(?:int|long|void|size_t|const char\*|void\*)\s+(cpx_[a-z_0-9]+)\s*\(", hdr, re.M))
    assert declared, "no prototypes found in cpx.h"
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.EXPORTS)
    assert lib.cpx_abi_version() == 3


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cpx import _lib

    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.Config(160, 120, 1, 45, 20.0, 0.1, 64, 512, 0, 0)
    assert lib.cpx_create(0, C.byref(cfg), C.byref(h)) == -3  # CPX_ERR_NO_DEVICE
    from cpx.engine import TrackEngine

    with pytest.raises(RuntimeError):
        TrackEngine()


def test_struct_layouts():
    from cpx import _lib, tracking

    assert C.sizeof(_lib.Config) == 48
    assert _lib.FRAME_META_DTYPE.itemsize == 24
    assert _lib.COMPONENT_DTYPE.itemsize == 32
    assert _lib.FRAME_INFO_DTYPE.itemsize == 80
    assert tracking.REGION_DTYPE.itemsize == 56
    assert tracking.TRACK_RECORD_DTYPE.itemsize == 32
    assert C.sizeof(tracking.TrackParams) == 120


def test_rectangle_like_reference_unit_tests():
    """Same cases as the reference's src/ml_tools/test_rectangle.py."""
    from cpx.ml_tools.rectangle import Rectangle

    r = Rectangle(2, 3, 5, 6)
    assert (r.left, r.top, r.width, r.height) == (2, 3, 5, 6)
    r = Rectangle(0, 0, 100, 100)
    r.crop(Rectangle(2, 3, 5, 6))
    assert (r.left, r.top, r.width, r.height) == (2, 3, 5, 6)
    image = np.arange(100).reshape((10, 10))
    assert np.array_equal(Rectangle(2, 3, 2, 3).subimage(image), [[32, 33], [42, 43], [52, 53]])
    e = Rectangle(5, 5, 10, 10)
    e.enlarge(4, max=Rectangle(1, 1, 158, 118))
    assert (e.x, e.y, e.width, e.height) == (1, 1, 18, 18)
    assert Rectangle(0, 0, 10, 10).overlap_area(Rectangle(5, 5, 10, 10)) == 25


def test_config_defaults_and_yaml_merge():
    from cpx.config import Config

    c = Config.get_defaults()
    t = c.tracking["thermal"]
    assert (t.edge_pixels, t.frame_padding, t.denoise, t.aoi_min_mass, t.aoi_pixel_variance) == (1, 4, True, 4.0, 2.0)
    assert t.params["base_distance_change"] == 450 and t.params["max_blanks"] == 18
    assert t.motion.threshold_for_model("lepton3.5").background_thresh == 50
    assert t.motion.threshold_for_model(None).background_thresh == 20
    c2 = Config.load_from_stream(io.StringIO("tracking:\n  thermal:\n    denoise: false\n    params:\n      max_blanks: 7\n"))
    assert c2.tracking["thermal"].denoise is False
    assert c2.tracking["thermal"].params["max_blanks"] == 7
    assert c2.tracking["thermal"].params["velocity_multiplier"] == 2
    assert set(c2.tracking["thermal"].as_dict()) == set(t.as_dict())


def test_cptv_reader_contract():
    from helpers import load_clip

    frames, t_on, ffc, bgf, hdr = load_clip("possum")
    assert frames.shape == (161, 120, 160) and frames.dtype == np.uint16
    assert (hdr.x_resolution, hdr.y_resolution, hdr.model, hdr.brand) == (160, 120, "lepton3", "flir")
    assert bgf[0] and not any(bgf[1:])
    # the background frame carries no times; every other frame has int milliseconds
    assert t_on[0] is None and all(isinstance(t, int) for t in t_on[1:]) and t_on[2] - t_on[1] == 114
    frames, t_on, ffc, bgf, hdr = load_clip("hedgehog")
    assert frames.shape == (119, 120, 160) and hdr.model is None and not any(bgf)
