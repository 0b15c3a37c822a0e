"""The reference's own unit tests of Rectangle (src/ml_tools/test_rectangle.py), same cases, against the host mirror
cpx.ml_tools.rectangle.Rectangle, plus the enlarge / area / overlap behaviour the trackers rely on."""
import numpy as np

from cpx.ml_tools.rectangle import Rectangle


def _is_2_3_5_6(rect):
    assert (rect.left, rect.top, rect.width, rect.height) == (2, 3, 5, 6)


def test_can_create_rectangle_from_width_and_height():
    _is_2_3_5_6(Rectangle(2, 3, 5, 6))


def test_crop():
    rectangle = Rectangle(0, 0, 100, 100)
    rectangle.crop(Rectangle(2, 3, 5, 6))
    _is_2_3_5_6(rectangle)


def test_subimage():
    image = np.arange(100).reshape((10, 10))
    subimage = Rectangle(2, 3, 2, 3).subimage(image)
    assert np.array_equal(subimage, [[32, 33], [42, 43], [52, 53]])


def test_edges_area_and_overlap():
    r = Rectangle(2, 3, 5, 6)
    assert (r.right, r.bottom, r.area) == (7, 9, 30)
    assert r.overlap_area(Rectangle(4, 5, 10, 10)) == 3 * 4
    assert r.overlap_area(Rectangle(20, 20, 2, 2)) == 0
