"""One frame of a clip as the tracker stores it (reference src/ml_tools/frame.py:20-32)."""


class Frame:
    __slots__ = ("thermal", "filtered", "frame_number", "mask", "flow", "flow_clipped", "scaled_thermal",
                 "ffc_affected", "region", "frame_temp_median", "preprocessed")

    def __init__(self, thermal, filtered, frame_number, mask=None, flow=None, flow_clipped=False,
                 scaled_thermal=None, ffc_affected=False, region=None, frame_temp_median=None, preprocessed=False):
        self.thermal = thermal
        self.filtered = filtered
        self.frame_number = frame_number
        self.mask = mask
        self.flow = flow
        self.flow_clipped = flow_clipped
        self.scaled_thermal = scaled_thermal
        self.ffc_affected = ffc_affected
        self.region = region
        self.frame_temp_median = frame_temp_median
        self.preprocessed = preprocessed

    def copy(self):
        return Frame(None if self.thermal is None else self.thermal.copy(),
                     None if self.filtered is None else self.filtered.copy(), self.frame_number,
                     None if self.mask is None else self.mask.copy(), ffc_affected=self.ffc_affected,
                     region=self.region, frame_temp_median=self.frame_temp_median)
