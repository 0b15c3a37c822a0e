"""In-memory frame store of a clip (reference src/track/framebuffer.py:26-166).
The HDF5 disk cache and optical flow of the reference are out of scope."""

from threading import Lock

from ..ml_tools.frame import Frame


class FrameBuffer:
    def __init__(self, cptv_name, high_quality_flow=False, cache_to_disk=False, calc_flow=False, keep_frames=True,
                 max_frames=None):
        if cache_to_disk:
            raise NotImplementedError("cpx keeps frames in memory / HBM: cache_to_disk is not supported")
        self.cache = None
        self.opt_flow = None
        self.max_frames = max_frames
        self.keep_frames = True if max_frames and max_frames > 0 else keep_frames
        self.current_frame_i = 0
        self.prev_frame = None
        self.current_frame = None
        self.frame_lock = Lock()
        self.reset()

    def reset(self):
        self.frames = []
        self.frames_by_frame_number = {}

    def add_frame(self, thermal, filtered, mask, frame_number, ffc_affected=False):
        self.prev_frame = self.current_frame
        frame = Frame(thermal, filtered, frame_number, mask=mask, ffc_affected=ffc_affected)
        self.current_frame = frame
        if self.keep_frames:
            if self.max_frames and len(self.frames) == self.max_frames:
                with self.frame_lock:
                    del self.frames_by_frame_number[self.frames[0].frame_number]
                    del self.frames[0]
            self.frames.append(frame)
            self.frames_by_frame_number[frame.frame_number] = frame
        return frame

    @property
    def has_flow(self):
        return False

    def get_frame(self, frame_number):
        if frame_number in self.frames_by_frame_number:
            return self.frames_by_frame_number[frame_number]
        if self.prev_frame and self.prev_frame.frame_number == frame_number:
            return self.prev_frame
        if self.current_frame and self.current_frame.frame_number == frame_number:
            return self.current_frame
        return None

    def get_last_x(self, x=25):
        return self.frames[-x:] if self.frames else None

    def close_cache(self):
        pass

    def remove_cache(self):
        pass

    def __len__(self):
        return len(self.frames)

    def __iter__(self):
        self.current_frame_i = 0
        return self

    def __next__(self):
        frame = self.get_frame(self.current_frame_i)
        if frame is None:
            raise StopIteration
        self.current_frame_i += 1
        return frame
