"""ctypes binding of libcpx_hip.so (include/cpx.h).  Loading never touches the
GPU; every compute call needs one and fails loudly otherwise."""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CPX_LIB") or os.path.join(_HERE, "libcpx_hip.so")  # CPX_LIB: tuning builds only

CPX_OK = 0
STATUS = {
    0: "CPX_OK",
    -1: "CPX_ERR_INVALID",
    -2: "CPX_ERR_UNSUPPORTED",
    -3: "CPX_ERR_NO_DEVICE",
    -4: "CPX_ERR_HIP",
    -5: "CPX_ERR_OVERFLOW",
    -6: "CPX_ERR_NOMEM",
}


class CpxError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        super().__init__("%s (%d) %s" % (STATUS.get(code, "CPX_ERR"), code, what))


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("edge_pixels", C.c_int32),
        ("window", C.c_int32),
        ("background_thresh", C.c_double),
        ("weight_add", C.c_double),
        ("max_components", C.c_int32),
        ("max_frames", C.c_int32),
        ("denoise", C.c_int32),
        ("reserved", C.c_int32),
    ]


FRAME_META_DTYPE = np.dtype(
    [("time_on_ms", "<i8"), ("last_ffc_ms", "<i8"), ("background_frame", "<i4"), ("has_times", "<i4")]
)
COMPONENT_DTYPE = np.dtype(
    [("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4"), ("area", "<i4"),
     ("sum_x", "<i4"), ("sum_y", "<i4"), ("pixel_variance", "<f4")]
)
FRAME_INFO_DTYPE = np.dtype(
    [("frame_number", "<i4"), ("n_components", "<i4"), ("status", "<i4"), ("ffc_affected", "<i4"),
     ("avg_change", "<i4"), ("norm_min", "<i4"), ("norm_max", "<i4"), ("threshold", "<f4"),
     ("filt_min", "<i4"), ("filt_max", "<i4"), ("thermal_min", "<i4"), ("thermal_max", "<i4"),
     ("thermal_sum", "<u4"), ("thermal_median", "<f4"), ("filtered_abs_sum", "<u8"),
     ("background_average", "<f8"), ("background_changed", "<i4"), ("reserved", "<i4")]
)
REGION_REF_DTYPE = np.dtype([("frame", "<i4"), ("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4"),
                             ("in_segment", "<i4")])
TRACK_LIMITS_DTYPE = np.dtype([("filt_min", "<f4"), ("filt_max", "<f4"), ("clip_at_zero", "<i4"), ("flags", "<i4"),
                               ("therm_min", "<f4"), ("therm_max", "<f4"), ("reserved", "<i4", (2,))])
CROP_REQ_DTYPE = np.dtype([("frame", "<i4"), ("x", "<i4"), ("y", "<i4"), ("width", "<i4"), ("height", "<i4"),
                           ("track", "<i4"), ("sample", "<i4"), ("tile", "<i4")])
THUMB_STAT_DTYPE = np.dtype([("contours", "<i4"), ("status", "<i4"), ("median_diff", "<f8")])
CONV_TIMING_DTYPE = np.dtype([("key", "<i4"), ("launches", "<i4"), ("total_ms", "<f8"), ("flops", "<f8")])
# cpx_cptv_file / cpx_cptv_file_result / cpx_cptv_frame_slot (gzip inflate + section index on the device)
CPTV_FILE_DTYPE = np.dtype([("in_offset", "<i8"), ("in_bytes", "<i8"), ("out_offset", "<i8"), ("out_capacity", "<i8"),
                            ("slot_offset", "<i8"), ("slot_capacity", "<i4"), ("reserved", "<i4")])
CPTV_RESULT_DTYPE = np.dtype([("status", "<i4"), ("n_frames", "<i4"), ("out_bytes", "<i8"), ("in_consumed", "<i8"),
                              ("header_bytes", "<i4"), ("width", "<i4"), ("height", "<i4"), ("reserved", "<i4")])
CPTV_SLOT_DTYPE = np.dtype([("offset", "<i8"), ("bit_width", "<i4"), ("time_on_ms", "<u4"), ("last_ffc_ms", "<u4"),
                            ("temp_c", "<f4"), ("last_ffc_temp_c", "<f4"), ("flags", "<u4")])
CPTV_HEADER_BYTES = 1024
CPTV_BACKGROUND_FRAME, CPTV_HAS_TIME_ON, CPTV_HAS_LAST_FFC = 1, 2, 4
CPTV_STATUS = {1: "deflate: reserved block type", 2: "deflate: stored block length", 3: "deflate: block header",
               4: "deflate: code lengths", 5: "deflate: invalid symbol", 6: "deflate: distance too far back",
               7: "inflated data larger than the gzip trailer says", 8: "deflate: input exhausted (truncated file)",
               9: "deflate: no end-of-block code", 10: "not a gzip member", 11: "gzip trailer / further member", 12: "gzip CRC-32 mismatch",
               20: "not a CPTV file", 21: "unsupported CPTV version", 22: "CPTV header section", 23: "expected frame section",
               24: "truncated CPTV frame", 25: "malformed CPTV frame section", 26: "more frames than slots",
               27: "CPTV file has no frames"}
assert CPTV_FILE_DTYPE.itemsize == 48 and CPTV_RESULT_DTYPE.itemsize == 40 and CPTV_SLOT_DTYPE.itemsize == 32
assert REGION_REF_DTYPE.itemsize == 24 and TRACK_LIMITS_DTYPE.itemsize == 32 and CROP_REQ_DTYPE.itemsize == 32
assert FRAME_META_DTYPE.itemsize == 24 and COMPONENT_DTYPE.itemsize == 32 and FRAME_INFO_DTYPE.itemsize == 80

class WRResNetBlock(C.Structure):  # struct cpx_wrresnet_block
    _fields_ = [(k, C.c_void_p) for k in ("in_scale", "in_shift", "wa", "a_scale", "a_shift", "wb", "bb")]


class WRResNetParams(C.Structure):  # struct cpx_wrresnet_params
    _fields_ = [("n_labels", C.c_int32), ("blocks_per_stage", C.c_int32), ("groups", C.c_int32),
                ("in_channels", C.c_int32), ("filters", C.c_int32 * 4),
                ("conv1_w", C.c_void_p), ("conv1_b", C.c_void_p),
                ("block", (WRResNetBlock * 8) * 3),
                ("shortcut_w", C.c_void_p * 3), ("shortcut_b", C.c_void_p * 3),
                ("final_scale", C.c_void_p), ("final_shift", C.c_void_p),
                ("dense_w", C.c_void_p), ("dense_b", C.c_void_p),
                ("n_hidden", C.c_int32), ("activation", C.c_int32), ("hidden_sizes", C.c_int32 * 4),
                ("hidden_w", C.c_void_p * 4), ("hidden_b", C.c_void_p * 4)]


class HeadDesc(C.Structure):  # struct cpx_head_desc
    _fields_ = [("N", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("L", C.c_int32),
                ("n_hidden", C.c_int32), ("activation", C.c_int32), ("hidden_sizes", C.c_int32 * 4),
                ("in_dev", C.c_void_p), ("bn_scale_dev", C.c_void_p), ("bn_shift_dev", C.c_void_p),
                ("hidden_w_dev", C.c_void_p * 4), ("hidden_b_dev", C.c_void_p * 4),
                ("dense_w_dev", C.c_void_p), ("dense_b_dev", C.c_void_p),
                ("logits_dev", C.c_void_p), ("probs_dev", C.c_void_p)]


assert C.sizeof(WRResNetParams) == 1560
HEAD_SIGMOID, HEAD_SOFTMAX = 0, 1


EXPORTS = [
    "cpx_abi_version", "cpx_create", "cpx_destroy", "cpx_last_error", "cpx_stream", "cpx_synchronize", "cpx_join_medians", "cpx_release_memory",
    "cpx_track_batch", "cpx_track_workspace_bytes", "cpx_last_kernel_timing", "cpx_associate_batch",
    "cpx_track_limits_batch", "cpx_crop_tile", "cpx_conv2d", "cpx_cnn_head",
    "cpx_finalize_tracks", "cpx_counts_prefix", "cpx_plan_segments", "cpx_aggregate_predictions",
    "cpx_conv_timing_enable", "cpx_conv_timing_report", "cpx_cptv_unpack", "cpx_thumb_stats", "cpx_trackless_thumb",
    "cpx_trackless_thumb_batch", "cpx_thumb_stats_ex",
    "cpx_track_frame", "cpx_associate_frame", "cpx_cnn_create", "cpx_cnn_destroy", "cpx_cnn_forward", "cpx_ir_detect", "cpx_set_cnn_math", "cpx_get_cnn_math",
    "cpx_mog2_create", "cpx_mog2_apply", "cpx_mog2_background", "cpx_mog2_destroy",
    "cpx_track_batch_ex", "cpx_track_frame_ex", "cpx_set_background", "cpx_get_background", "cpx_track_limits_batch_ex",
    "cpx_cnn_head_ex", "cpx_ir_delta_variance", "cpx_cptv_inflate", "cpx_cptv_gather_index", "cpx_format_regions", "cpx_json_indent", "cpx_ir_merge", "cpx_ir_resize_area",
    "cpx_ir_frame_statistics", "cpx_cnn_last_overflow", "cpx_cnn_set_activation_bounds", "cpx_cnn_overflow_forwards",
]

IR_FRAME_STATS_DTYPE = np.dtype([("min", "<i4"), ("max", "<i4"), ("sum", "<i8"), ("median_x2", "<i4"), ("reserved", "<i4"),
                                 ("filtered_sum", "<i8")])
assert IR_FRAME_STATS_DTYPE.itemsize == 32

# flags of cpx_track_batch_ex / cpx_track_frame_ex and cpx_track_limits_batch_ex (include/cpx.h)
TRACK_KEEP_BACKGROUND, TRACK_FREEZE_ON_FFC, TRACK_FREEZE_BACKGROUND, TRACK_DEFER_MEDIANS = 1, 2, 4, 8
LIMITS_POST_PROCESS, LIMITS_THERMAL_DIFF_NORM, LIMITS_NO_DIFF_NORM, LIMITS_ALWAYS_CLIP, LIMITS_SWAP_CHANNELS = 1, 2, 4, 8, 16
LIMITS_TF_SCALING = 32

_lib = None


def load():
    """dlopen the library and declare the prototypes.  Raises if it is missing:
    there is no CPU fallback for the compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcpx_hip.so not built: run `python __graft_entry__.py` (or make -C classifier-pipeline_amd/csrc)")
    # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime of the
    # process (two runtimes cannot share the device), so load torch before us.
    try:
        import torch  # noqa: F401
    except ImportError:  # pure ctypes use without torch: the system runtime is fine
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
    lib.cpx_abi_version.restype = C.c_int
    lib.cpx_create.argtypes = [C.c_int, C.POINTER(Config), C.POINTER(vp)]
    lib.cpx_create.restype = C.c_int
    lib.cpx_destroy.argtypes = [vp]
    lib.cpx_destroy.restype = None
    lib.cpx_last_error.argtypes = [vp]
    lib.cpx_last_error.restype = C.c_char_p
    lib.cpx_stream.argtypes = [vp]
    lib.cpx_stream.restype = vp
    lib.cpx_synchronize.argtypes = [vp]
    lib.cpx_synchronize.restype = C.c_int
    lib.cpx_join_medians.argtypes = [vp]
    lib.cpx_join_medians.restype = C.c_int
    lib.cpx_release_memory.argtypes = [vp]
    lib.cpx_release_memory.restype = C.c_int
    lib.cpx_track_batch.argtypes = [vp, vp, i32p, vp, C.c_int, vp, vp, vp, vp, vp]
    lib.cpx_track_batch.restype = C.c_int
    lib.cpx_associate_batch.argtypes = [vp, vp, i32p, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cpx_associate_batch.restype = C.c_int
    lib.cpx_track_limits_batch.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp]
    lib.cpx_track_limits_batch.restype = C.c_int
    lib.cpx_crop_tile.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, vp]
    lib.cpx_crop_tile.restype = C.c_int
    lib.cpx_conv2d.argtypes = [vp, vp]
    lib.cpx_conv2d.restype = C.c_int
    lib.cpx_cnn_head.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp]
    lib.cpx_cnn_head.restype = C.c_int
    lib.cpx_cnn_head_ex.argtypes = [vp, C.POINTER(HeadDesc)]
    lib.cpx_cnn_head_ex.restype = C.c_int
    lib.cpx_finalize_tracks.argtypes = [vp, vp, i32p, vp, C.c_int, vp, vp, vp, vp, vp]
    lib.cpx_finalize_tracks.restype = C.c_int
    lib.cpx_plan_segments.argtypes = [vp, vp, i32p, vp, C.c_int, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]
    lib.cpx_plan_segments.restype = C.c_int
    lib.cpx_counts_prefix.argtypes = [vp, vp, C.c_int, vp]
    lib.cpx_counts_prefix.restype = C.c_int
    lib.cpx_aggregate_predictions.argtypes = [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.cpx_aggregate_predictions.restype = C.c_int
    lib.cpx_conv_timing_enable.argtypes = [vp, C.c_int]
    lib.cpx_conv_timing_enable.restype = C.c_int
    lib.cpx_conv_timing_report.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
    lib.cpx_conv_timing_report.restype = C.c_int
    lib.cpx_set_cnn_math.argtypes = [vp, C.c_int]
    lib.cpx_set_cnn_math.restype = C.c_int
    lib.cpx_get_cnn_math.argtypes = [vp]
    lib.cpx_get_cnn_math.restype = C.c_int
    lib.cpx_cnn_last_overflow.argtypes = [vp, C.POINTER(C.c_int)]
    lib.cpx_cnn_last_overflow.restype = C.c_int
    lib.cpx_cnn_overflow_forwards.argtypes = [vp, C.POINTER(C.c_int), C.c_int]
    lib.cpx_cnn_overflow_forwards.restype = C.c_int
    lib.cpx_cnn_set_activation_bounds.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    lib.cpx_cnn_set_activation_bounds.restype = C.c_int
    lib.cpx_mog2_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.POINTER(vp)]
    lib.cpx_mog2_create.restype = C.c_int
    lib.cpx_mog2_apply.argtypes = [vp, vp, C.c_double, vp]
    lib.cpx_mog2_apply.restype = C.c_int
    lib.cpx_mog2_background.argtypes = [vp, vp]
    lib.cpx_mog2_background.restype = C.c_int
    lib.cpx_mog2_destroy.argtypes = [vp]
    lib.cpx_mog2_destroy.restype = None
    lib.cpx_ir_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.cpx_ir_detect.restype = C.c_int
    lib.cpx_ir_delta_variance.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.cpx_ir_delta_variance.restype = C.c_int
    lib.cpx_ir_merge.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.cpx_ir_merge.restype = C.c_int
    lib.cpx_ir_resize_area.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]
    lib.cpx_ir_resize_area.restype = C.c_int
    lib.cpx_ir_frame_statistics.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.cpx_ir_frame_statistics.restype = C.c_int
    lib.cpx_cnn_create.argtypes = [vp, C.POINTER(WRResNetParams), C.POINTER(vp)]
    lib.cpx_cnn_create.restype = C.c_int
    lib.cpx_cnn_destroy.argtypes = [vp]
    lib.cpx_cnn_destroy.restype = None
    lib.cpx_cnn_forward.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.cpx_cnn_forward.restype = C.c_int
    lib.cpx_track_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    lib.cpx_track_frame.restype = C.c_int
    lib.cpx_track_batch_ex.argtypes = [vp, vp, i32p, vp, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    lib.cpx_track_batch_ex.restype = C.c_int
    lib.cpx_track_frame_ex.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int]
    lib.cpx_track_frame_ex.restype = C.c_int
    lib.cpx_set_background.argtypes = [vp, C.c_int, vp, vp, C.c_double]
    lib.cpx_set_background.restype = C.c_int
    lib.cpx_get_background.argtypes = [vp, C.c_int, vp, vp, C.POINTER(C.c_double)]
    lib.cpx_get_background.restype = C.c_int
    lib.cpx_track_limits_batch_ex.argtypes = [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int]
    lib.cpx_track_limits_batch_ex.restype = C.c_int
    lib.cpx_associate_frame.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.cpx_associate_frame.restype = C.c_int
    lib.cpx_thumb_stats.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    lib.cpx_thumb_stats.restype = C.c_int
    lib.cpx_thumb_stats_ex.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int]
    lib.cpx_thumb_stats_ex.restype = C.c_int
    lib.cpx_trackless_thumb.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    lib.cpx_trackless_thumb.restype = C.c_int
    lib.cpx_trackless_thumb_batch.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.cpx_trackless_thumb_batch.restype = C.c_int
    lib.cpx_cptv_unpack.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp]
    lib.cpx_cptv_unpack.restype = C.c_int
    lib.cpx_cptv_inflate.argtypes = [vp, vp, vp, C.c_int, vp, vp, vp, vp]
    lib.cpx_cptv_inflate.restype = C.c_int
    lib.cpx_cptv_gather_index.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp]
    lib.cpx_cptv_gather_index.restype = C.c_int
    lib.cpx_format_regions.argtypes = [vp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_int, vp, C.c_long]
    lib.cpx_format_regions.restype = C.c_long
    lib.cpx_json_indent.argtypes = [C.c_char_p, C.c_long, C.c_int, C.c_int, vp, C.c_long]
    lib.cpx_json_indent.restype = C.c_long
    lib.cpx_track_workspace_bytes.argtypes = [vp, C.c_int, C.c_int]
    lib.cpx_track_workspace_bytes.restype = C.c_size_t
    lib.cpx_last_kernel_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.cpx_last_kernel_timing.restype = C.c_int
    if lib.cpx_abi_version() != 3:
        raise ImportError("libcpx_hip.so ABI version mismatch")
    _lib = lib
    return lib
