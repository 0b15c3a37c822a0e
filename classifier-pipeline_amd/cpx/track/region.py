"""Region = Rectangle + mass / variance / flags (reference src/track/region.py:27-209)."""

import numpy as np

from ..ml_tools.rectangle import Rectangle
from ..tracking import REGION_BLANK, REGION_BORDER, REGION_CENTROID_F32, REGION_CROPPED


class Region(Rectangle):
    __slots__ = ("centroid", "mass", "frame_number", "pixel_variance", "id", "was_cropped", "blank",
                 "is_along_border", "in_trap")

    def __init__(self, x, y, width, height, centroid=None, mass=0, frame_number=0, pixel_variance=0, id=0,
                 was_cropped=False, blank=False, is_along_border=False, in_trap=False):
        super().__init__(x, y, width, height)
        self.centroid = centroid
        self.mass = mass
        self.frame_number = frame_number
        self.pixel_variance = pixel_variance
        self.id = id
        self.was_cropped = was_cropped
        self.blank = blank
        self.is_along_border = is_along_border
        self.in_trap = in_trap

    @classmethod
    def from_record(cls, rec):
        """cpx_region (device output) -> Region; centroid keeps the reference's dtype
        (float32 pair for Kalman-predicted blanks, float64 otherwise)."""
        flags = int(rec["flags"])
        if flags & REGION_CENTROID_F32:
            centroid = [np.float32(rec["cx"]), np.float32(rec["cy"])]
        else:
            centroid = np.array([rec["cx"], rec["cy"]], dtype=np.float64)
        blank = bool(flags & REGION_BLANK)
        return cls(int(rec["x"]), int(rec["y"]), int(rec["width"]), int(rec["height"]), centroid=centroid,
                   mass=int(rec["mass"]), frame_number=int(rec["frame_number"]),
                   pixel_variance=0 if blank else np.float32(rec["pixel_variance"]), id=int(rec["id"]),
                   was_cropped=bool(flags & REGION_CROPPED), blank=blank,
                   is_along_border=bool(flags & REGION_BORDER))

    @classmethod
    def region_from_json(cls, j):
        frame = j.get("frame_number", j.get("frameNumber", j.get("order")))
        centroid = j.get("centroid") or [int(j["x"] + j["width"] / 2), int(j["y"] + j["height"] / 2)]
        return cls(j["x"], j["y"], j["width"], j["height"], frame_number=frame, mass=j.get("mass") or 0,
                   blank=j.get("blank", False), pixel_variance=j.get("pixel_variance", 0), centroid=centroid)

    @classmethod
    def region_from_array(cls, b):
        width = max(int(b[2]) - b[0], 0)
        height = max(int(b[3]) - b[1], 0)
        frame = b[4] if len(b) > 4 else None
        mass = b[5] if len(b) > 5 else 0
        blank = len(b) > 6 and b[6] == 1
        return cls(b[0], b[1], width, height, frame_number=frame, mass=mass, blank=blank,
                   centroid=[int(b[0] + width / 2), int(b[1] + height / 2)])

    def to_array(self):
        return np.uint16([self.left, self.top, self.right, self.bottom, self.frame_number, self.mass,
                          1 if self.blank else 0])

    def copy(self):
        return Region(self.x, self.y, self.width, self.height, self.centroid, self.mass, self.frame_number,
                      self.pixel_variance, self.id, self.was_cropped, self.blank, self.is_along_border)

    def has_moved(self, other):
        return (self.x != other.x and self.right != other.right) or (self.y != other.y and self.bottom != other.bottom)

    def set_is_along_border(self, bounds, edge=0):
        self.is_along_border = (self.was_cropped or self.x <= bounds.x + edge or self.y <= bounds.y + edge
                                or self.right >= bounds.width - edge or self.bottom >= bounds.height - edge)

    def on_height_edge(self, crop_region):
        return self.top == crop_region.top or self.bottom == crop_region.bottom

    def on_width_edge(self, crop_region):
        return self.left == crop_region.left or self.right == crop_region.right

    def meta_dictionary(self):
        pv = self.pixel_variance
        return {"x": self.x, "y": self.y, "width": self.width, "height": self.height, "mass": self.mass,
                "frame_number": self.frame_number, "pixel_variance": round(pv, 2) if pv is not None else 0,
                "blank": self.blank, "in_trap": self.in_trap}
