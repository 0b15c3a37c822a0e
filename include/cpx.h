/*
 * cpx.h -- C-ABI of libcpx_hip.so, the MI355X (gfx950) implementation of the
 * thermal extract-and-classify hot path of TheCacophonyProject/classifier-pipeline.
 *
 * The reference has no FFI layer: its hot path is Python over OpenCV / NumPy.
 * Each entry point below replaces the arithmetic of the reference call sites
 * cited next to it (paths relative to /root/reference/src); the Python host
 * classes in classifier-pipeline_amd/cpx mirror the reference classes and bind
 * these symbols with ctypes.
 *
 * Conventions: plain C types; int return (0 = ok, negative = cpx_status);
 * caller-allocated outputs with explicit capacities (overflow is an error,
 * never truncation); pointers named *_dev are DEVICE pointers (e.g.
 * torch.Tensor.data_ptr()), all others are host pointers; one HIP stream per
 * handle; a handle is not thread-safe; no global state.
 */
#ifndef CPX_H
#define CPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPX_ABI_VERSION 3

typedef enum cpx_status {
  CPX_OK = 0,
  CPX_ERR_INVALID = -1,      /* bad argument */
  CPX_ERR_UNSUPPORTED = -2,  /* resolution / option outside the kernels' envelope */
  CPX_ERR_NO_DEVICE = -3,    /* no HIP device / wrong architecture */
  CPX_ERR_HIP = -4,          /* HIP runtime error, see cpx_last_error */
  CPX_ERR_OVERFLOW = -5,     /* a frame produced more components than max_components */
  CPX_ERR_NOMEM = -6
} cpx_status;

typedef struct cpx_handle cpx_handle;

/* Tracking configuration: the values the pixel stage reads from the reference's
 * TrackingConfig / ThresholdConfig (config/trackingconfig.py:126-177,
 * config/trackingmotionconfig.py:24-59) and cliptrackextractor.py:124-127. */
typedef struct cpx_config {
  int32_t width;             /* 160 */
  int32_t height;            /* 120 */
  int32_t edge_pixels;       /* 1   (crop rectangle = interior) */
  int32_t window;            /* 45  frames averaged for the background feed */
  double background_thresh;  /* 20 (lepton3) / 50 (lepton3.5) */
  double weight_add;         /* 0.1 (lepton3) / 1.0 (lepton3.5) */
  int32_t max_components;    /* capacity of the per-frame component list */
  int32_t max_frames;        /* longest clip (frames) the handle must support */
  int32_t denoise;           /* 1: cv2.fastNlMeansDenoising (tracking.denoise, the reference's default) */
  int32_t reserved;
} cpx_config;

/* Per-frame metadata delivered by the CPTV reader (cptv.py; reference
 * cliptrackextractor.py:160-169, cptvmotiondetector.py:211-223). */
typedef struct cpx_frame_meta {
  int64_t time_on_ms;
  int64_t last_ffc_ms;
  int32_t background_frame; /* 1: used to initialise the background only */
  int32_t has_times;        /* 0: time_on / last_ffc are None */
} cpx_frame_meta;

/* One connected component = one row of cv2.connectedComponentsWithStats
 * (imageprocessing.py:248) plus the centroid numerators and the variance of the
 * inter-frame delta over its bounding box (cliptracker.py:316-318). */
typedef struct cpx_component {
  int32_t x, y, width, height, area;
  int32_t sum_x, sum_y;   /* centroid = (sum_x/area, sum_y/area) in float64 */
  float pixel_variance;   /* np.var(delta_filtered[bbox]); 0 on a clip's first frame */
} cpx_component;

/* Per-frame scalars (cliptracker.py:93-122, clip.py:474-487, motiondetector.py:226). */
typedef struct cpx_frame_info {
  int32_t frame_number;     /* index among processed frames of the clip; -1 = skipped */
  int32_t n_components;
  int32_t status;           /* 0 ok, CPX_ERR_OVERFLOW */
  int32_t ffc_affected;
  int32_t avg_change;
  int32_t norm_min, norm_max; /* min / max of the clipped, shifted frame */
  float threshold;          /* mapped threshold handed to cv2.threshold */
  int32_t filt_min, filt_max; /* min / max of thermal - background */
  int32_t thermal_min, thermal_max;
  uint32_t thermal_sum;     /* mean = thermal_sum / (W*H) */
  float thermal_median;     /* np.median(thermal) */
  uint64_t filtered_abs_sum;
  double background_average; /* WeightedBackground.average after this frame's update */
  int32_t background_changed;
  int32_t reserved;
} cpx_frame_info;

/* One entry of Track.bounds_history / clip.region_history (track/region.py:27-42). */
#define CPX_REGION_BLANK 1         /* Region.blank */
#define CPX_REGION_CROPPED 2       /* Region.was_cropped */
#define CPX_REGION_BORDER 4        /* Region.is_along_border */
#define CPX_REGION_CENTROID_F32 8  /* centroid came from the float32 Kalman prediction */
#define CPX_REGION_WIDTH_PYINT 16  /* width / height are Python ints in the reference (not np.int32): decides the */
#define CPX_REGION_HEIGHT_PYINT 32 /* dtype of the next Kalman blank region's arithmetic (track.py:247-253) */
typedef struct cpx_region {
  int32_t x, y, width, height;
  int32_t mass;
  int32_t frame_number;
  float pixel_variance;
  int32_t flags;
  double cx, cy; /* centroid */
  int32_t id;    /* component index the region came from (Region.id) */
  int32_t pad;
} cpx_region;

/* One Track (track/track.py:372): its regions are pool[(start_frame + i) * max_active_tracks + slot],
 * i in [0, n_frames), relative to the clip's first pool row. */
typedef struct cpx_track_record {
  int32_t id, slot, start_frame, n_frames;
  int32_t blank_frames, since_seen, rt_frames; /* RegionTracker counters (track.py:65-75) */
  int32_t track_index;
} cpx_track_record;

/* Region filter + RegionTracker parameters (config/trackingconfig.py:126-177). */
typedef struct cpx_track_params {
  int32_t crop_x, crop_y, crop_w, crop_h; /* clip.crop_rectangle (clip.py:396-400) */
  int32_t frame_padding, min_dimension;
  int32_t cropped_regions_strategy; /* 0 "cautious", 1 "none", 2 "all" */
  int32_t filter_regions_pre_match;
  double aoi_min_mass, aoi_pixel_variance;
  double base_distance_change, min_mass_change, restrict_mass_after, mass_change_percent;
  double velocity_multiplier, base_velocity;
  int32_t has_min_mass_change, has_mass_change_percent; /* 0: the Python value is None */
  int32_t max_blanks, fps;
  int32_t max_active_tracks; /* pool slots per frame */
  int32_t max_tracks;        /* track records per clip */
} cpx_track_params;

/* ---- lifetime ---------------------------------------------------------- */
int cpx_abi_version(void);
int cpx_create(int device_id, const cpx_config* cfg, cpx_handle** out);
void cpx_destroy(cpx_handle* h);
const char* cpx_last_error(const cpx_handle* h);
/* HIP stream of the handle as an opaque pointer (hipStream_t). */
void* cpx_stream(cpx_handle* h);
int cpx_synchronize(cpx_handle* h);

/* ---- incremental tracking: one clip, frame by frame (the Pi-style caller) ------------------------------
 * Replaces ClipTrackExtractor.process_frame / start_tracking (track/cliptrackextractor.py:181-247) followed
 * by the caller's background update (cliptrackextractor.py:169-176): the same kernels as cpx_track_batch /
 * cpx_associate_batch, run only for the frames [n_prev, n_frames) of ONE clip whose earlier frames were
 * consumed by previous calls on this handle (its background, window sum, weights and association state
 * stay in the handle's workspace).  n_prev == 0 opens a new stream.  frames_dev holds the clip so far --
 * the kernels read the frame leaving the 45-frame window from it -- and the output arrays are indexed by
 * frame like the batch calls', caller-allocated for the clip's capacity; meta holds all n_frames entries.
 * Keep the same optional outputs (labels / filtered) for the whole stream.  A cpx_track_batch /
 * cpx_associate_batch call on the handle ends the stream.  cpx_track_frame: n_prev that does not match the
 * frames consumed -> CPX_ERR_INVALID.  cpx_associate_frame may skip frames: frames between its previous
 * n_frames and n_prev are never associated (start_tracking(track_frames=False) preview frames); the first
 * call after a new cpx_track_frame stream starts with no tracks. */
int cpx_track_frame(cpx_handle* h, const uint16_t* frames_dev, const cpx_frame_meta* meta, int n_prev, int n_frames,
                    cpx_component* comps_dev, cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                    float* background_dev);

/* ---- who owns the background ---------------------------------------------------------------------------
 * The reference's process_frame never updates the background itself (track/cliptrackextractor.py:198-247): its
 * callers do -- _track_clip with the 45-frame mean after every frame (:169-176), post_process_file the same but
 * not on FFC-affected frames and CONTINUING from the state tracking left (classify/clipclassifier.py:418-431,
 * 460-512), the Pi loop not at all: there the motion detector owns the WeightedBackground and hands it in
 * (start_tracking(..., background_alg=), update_background=False; piclassifier/piclassifier.py:322-333,423-431).
 * flags of the _ex calls:
 *   CPX_TRACK_KEEP_BACKGROUND    do not seed background / weights / average from the clips' first frames: every clip
 *                                continues from the state the previous track call on this handle left for the clip of
 *                                the same index (same B), or from what cpx_set_background staged for it; the 45-frame
 *                                window starts empty as always.  The per-pixel count of consecutive kept frames is a
 *                                uint16 and the weight tables hold all 65,536 values: a pixel kept for more than
 *                                65,535 frames in a row (two hours at 9 fps) across chained calls wraps to 0
 *   CPX_TRACK_FREEZE_ON_FFC      FFC-affected frames leave background, weights and average untouched
 *   CPX_TRACK_FREEZE_BACKGROUND  no frame updates them (update_background = False: the caller owns the model)
 *   CPX_TRACK_DEFER_MEDIANS      (cpx_track_batch_ex; ABI 3) cpx_frame_info.thermal_median -- np.median(thermal) of
 *                                ClipStats.add_frame, track/clip.py:474-487, which only the clip statistics and the
 *                                classifier's crops read -- is computed on the handle's SECOND stream behind the frame
 *                                kernel, beside whatever the caller enqueues next on cpx_stream(h): the association,
 *                                finalisation and segment-plan stages do not read it and are latency-bound, the medians
 *                                cost nothing next to them (10 ms of a 377 ms step of 4096 clips).  The medians are
 *                                complete for every later entry point of the handle that reads them
 *                                (cpx_track_limits_batch(_ex), cpx_crop_tile, cpx_thumb_stats(_ex)), for the next
 *                                cpx_track_* call, after cpx_synchronize(h), and for work enqueued on cpx_stream(h) after
 *                                cpx_join_medians(h) -- NOT for a caller who only synchronises cpx_stream(h) itself.
 *                                frames_dev and info_dev must stay allocated until one of those.
 * cpx_track_batch / cpx_track_frame are the _ex calls with flags 0. */
#define CPX_TRACK_KEEP_BACKGROUND 1
#define CPX_TRACK_FREEZE_ON_FFC 2
#define CPX_TRACK_FREEZE_BACKGROUND 4
#define CPX_TRACK_DEFER_MEDIANS 8
/* Orders everything enqueued on cpx_stream(h) from here on behind the medians a CPX_TRACK_DEFER_MEDIANS call left in
 * flight (a stream wait, the host does not block); nothing to do otherwise. */
int cpx_join_medians(cpx_handle* h);
/* Waits for the handle's streams and frees the device memory the handle allocates on demand and can allocate again: the
 * network's activation arena (four times the largest activation of the largest batch seen: 41 GB at 1,559 samples of
 * 160 x 160), the weight-split scratch, the track / association workspaces (with them the state CPX_TRACK_KEEP_BACKGROUND,
 * cpx_get_background and the *_frame calls continue from) and the IR scratch.  For a handle that is kept (its stream still
 * owns result buffers) but will not run soon; cpx_destroy frees everything. */
int cpx_release_memory(cpx_handle* h);
int cpx_track_batch_ex(cpx_handle* h, const uint16_t* frames_dev, const int32_t* clip_offsets,
                       const cpx_frame_meta* meta, int B, cpx_component* comps_dev, cpx_frame_info* info_dev,
                       int32_t* labels_dev, float* filtered_dev, float* background_dev, int flags);
int cpx_track_frame_ex(cpx_handle* h, const uint16_t* frames_dev, const cpx_frame_meta* meta, int n_prev, int n_frames,
                       cpx_component* comps_dev, cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                       float* background_dev, int flags);
/* The state of a WeightedBackground (piclassifier/motiondetector.py:178-248) as HOST arrays: background float
 * [H, W] (integer-valued, edges replicated), weights double [H - 2 edge, W - 2 edge] or NULL (all zero), average.
 * cpx_set_background stages it for clip `clip` of the NEXT track call on the handle (batch, or frame of a stream --
 * also in the middle of a stream: an externally owned model that changed between two frames); it replaces whatever
 * seeding that call would do for the clip.  Values the device state cannot hold exactly (a non-integer or > 65535
 * background, a weight that is not an accumulation of the handle's weight_add) -> CPX_ERR_UNSUPPORTED.  A track call
 * that is refused over the staged states (one staged for a clip index it does not have, or CPX_TRACK_KEEP_BACKGROUND
 * without a state for every clip) fails with CPX_ERR_INVALID and drops all staged states.
 * cpx_get_background reads the state the last track call left for clip `clip` (synchronises the handle's stream). */
int cpx_set_background(cpx_handle* h, int clip, const float* background, const double* weights, double average);
int cpx_get_background(cpx_handle* h, int clip, float* background, double* weights, double* average);
int cpx_associate_frame(cpx_handle* h, const cpx_track_params* params, const cpx_frame_meta* meta, int n_prev,
                        int n_frames, const cpx_component* comps_dev, const cpx_frame_info* info_dev,
                        cpx_region* pool_dev, cpx_track_record* tracks_dev, int32_t* n_tracks_dev,
                        int32_t* status_dev, cpx_region* regions_dev, int32_t* region_counts_dev);

/* ---- CPTV v2 frame payload decode ------------------------------------------
 * Replaces the bit-unpack / running-sum / snake-order / inter-frame accumulation of the Rust CPTV
 * reader (python-cptv 0.0.8, used at track/cliptrackextractor.py:108-129,160-162) for B clips.
 * The gzip container is inflated and its section headers parsed on the host (cpx/cptv.py); this call
 * gets the inflated bytes.  Per frame: int32 LE first value, then W*H-1 signed `bit_width`-bit deltas,
 * MSB first; their running sum, laid out in snake order (odd rows right to left), is added to the
 * previous frame of the same clip.
 * payload_dev        uint8  inflated file bytes of all clips, concatenated
 * frame_offsets_dev  int64  [total_frames] byte offset of each frame's payload inside payload_dev
 * bit_widths_dev     int32  [total_frames]
 * clip_offsets_dev   int32  [B+1] frame ranges of the clips
 * frames_out_dev     uint16 [total_frames, H, W]
 */
int cpx_cptv_unpack(cpx_handle* h, const uint8_t* payload_dev, const int64_t* frame_offsets_dev,
                    const int32_t* bit_widths_dev, const int32_t* clip_offsets_dev, int B,
                    uint16_t* frames_out_dev);

/* ---- CPTV v2 container on the GPU: gzip inflate + section index -------------------------
 * Replaces CptvReader(path) / get_header() / next_frame() of the Rust reader (python-cptv 0.0.8 -> flate2) at the
 * call sites track/cliptrackextractor.py:108-129,160-162 and classify/clipclassifier.py:460-469, for B WHOLE FILES
 * per call: the bytes of the .cptv files go to the device as they are on disk.  One wavefront per file parses the
 * gzip member header (RFC 1952), inflates the DEFLATE stream (RFC 1951: stored, fixed and dynamic blocks; decode
 * tables in LDS) into out_dev and walks the CPTV sections of what it inflated: header fields, then per frame the
 * field list and the offset of the payload -- what cpx_cptv_unpack needs.  A file that fails (corrupt stream, output
 * larger than its capacity, malformed section, more frames than slots) gets a non-zero status in its result record and
 * does not affect the others; nothing is written outside the file's own output / slot ranges.
 *
 * in_dev        the files' bytes, file i at files[i].in_offset (multiple of 4), in_bytes long; the buffer must be
 *               readable for 8 bytes past the last file
 * files_dev     [B] where each file's input, output and frame slots live
 * out_dev       inflated bytes: file i at out_offset (multiple of 16), at most out_capacity bytes (the gzip trailer's
 *               ISIZE, the last four bytes of a single-member file, is the exact figure)
 * slots_dev     frame slots: file i's frames at slot_offset .. slot_offset + n_frames (<= slot_capacity)
 * header_dev    [B][CPX_CPTV_HEADER_BYTES] the first bytes of every inflated file (magic, version, header section) for
 *               the host to read the camera model, timestamps ... from
 * results_dev   [B]
 * Status values: 0 ok; 1-9 DEFLATE errors (block type, stored length, code lengths, table, symbol, distance, output
 * capacity, input exhausted, no end-of-block code); 10 gzip header; 11 data after the member's trailer / ISIZE mismatch
 * (a multi-member file: inflate it on the host); 12 the CRC-32 of the inflated bytes differs from the trailer's (computed
 * by the file's own wave: byte-table recurrence per lane over 64 chunks, folded with the zero-bytes operator of
 * crc32_combine); 20-27 CPTV section errors (magic, version, header, section tag,
 * truncated, frame fields, slots, no frames).  Asynchronous on the handle's stream. */
#define CPX_CPTV_HEADER_BYTES 1024
#define CPX_CPTV_BACKGROUND_FRAME 1u
#define CPX_CPTV_HAS_TIME_ON 2u
#define CPX_CPTV_HAS_LAST_FFC 4u
typedef struct cpx_cptv_file {
  int64_t in_offset, in_bytes;
  int64_t out_offset, out_capacity;
  int64_t slot_offset;
  int32_t slot_capacity, reserved;
} cpx_cptv_file;
typedef struct cpx_cptv_file_result {
  int32_t status, n_frames;
  int64_t out_bytes;        /* inflated size */
  int64_t in_consumed;      /* bytes of the file up to and including the gzip trailer */
  int32_t header_bytes;     /* offset of the first frame section in the inflated data */
  int32_t width, height;    /* header fields X, Y */
  int32_t reserved;
} cpx_cptv_file_result;
typedef struct cpx_cptv_frame_slot {
  int64_t offset;           /* of the frame's payload inside out_dev */
  int32_t bit_width;
  uint32_t time_on_ms, last_ffc_ms;
  float temp_c, last_ffc_temp_c;
  uint32_t flags;           /* CPX_CPTV_* */
} cpx_cptv_frame_slot;
int cpx_cptv_inflate(cpx_handle* h, const uint8_t* in_dev, const cpx_cptv_file* files_dev, int B, uint8_t* out_dev,
                     cpx_cptv_frame_slot* slots_dev, uint8_t* header_dev, cpx_cptv_file_result* results_dev);
/* The slots of the files that decoded (clip b = file file_index[b], n_frames[b] frames) gathered into the dense
 * per-frame arrays of cpx_cptv_unpack: frame_offsets_dev int64 [total], bit_widths_dev int32 [total], and
 * slots_out_dev [total] (the same records, dense: times / flags for the host).  clip_offsets_dev int32 [B+1]. */
int cpx_cptv_gather_index(cpx_handle* h, const cpx_cptv_frame_slot* slots_dev, const int64_t* slot_offsets_dev,
                          const int32_t* clip_offsets_dev, int B, int64_t* frame_offsets_dev, int32_t* bit_widths_dev,
                          cpx_cptv_frame_slot* slots_out_dev);

/* ---- metadata text (host only; no device work) ---------------------------------------------------------------------
 * The JSON of `n` regions as json.dump(..., indent=indent, cls=CustomJSONEncoder) writes Region.meta_dictionary()
 * entries (reference src/track/track.py:1001-1031 "positions", src/ml_tools/rectangle.py:164-177): as a list whose items
 * sit at nesting `depth` (as_list = 1), or one dict whose keys sit at nesting `depth` (as_list = 0, n = 1); indent = 0
 * writes json.dumps' one-line form.  regs: records stride_bytes apart (a track's rows of the association pool).
 * Returns the bytes written, or minus the bytes needed when `cap` is too small. */
long cpx_format_regions(const cpx_region* regs, int n, long stride_bytes, int indent, int depth, int as_list, char* out,
                        long cap);
/* The text json.dumps(obj, indent=indent) writes, from the text json.dumps(obj) wrote (separators ", " and ": "): CPython
 * encodes in C only without an indent, so the file-fed path encodes compactly and lays the text out here.  depth0: the
 * nesting level the text starts at (0 for a whole document).  Returns the bytes written or minus the bytes needed. */
long cpx_json_indent(const char* in, long n, int indent, int depth0, char* out, long cap);

/* ---- track stage: background + filtered + threshold + CC + stats ---------
 * Replaces, for a batch of B independent clips, the per-frame arithmetic of
 *   ClipTrackExtractor.init_clip / _track_clip / process_frame
 *                                  (track/cliptrackextractor.py:98-247)
 *   ClipTracker._get_filtered_frame / get_delta_frame / np.var per region
 *                                  (track/cliptracker.py:93-122,249-261,316-318)
 *   detect_objects                 (ml_tools/imageprocessing.py:240-248)
 *   WeightedBackground.process_frame (piclassifier/motiondetector.py:197-248)
 *   ClipStats.add_frame            (track/clip.py:474-487)
 *
 * frames_dev        uint16 [total_frames, H, W]  every frame of every clip, file order
 * clip_offsets      host   [B+1]   frame offsets of the clips inside frames_dev
 * meta              host   [total_frames]
 * comps_dev         [total_frames * max_components]   component lists (label order)
 * info_dev          [total_frames]
 * labels_dev        int32 [total_frames, H, W] or NULL  (Frame.mask)
 * filtered_dev      float [total_frames, H, W] or NULL  (Frame.filtered)
 * background_dev    float [B, H, W] or NULL: final background of every clip
 * Asynchronous on the handle's stream; call cpx_synchronize before reading.
 */
int cpx_track_batch(cpx_handle* h, const uint16_t* frames_dev, const int32_t* clip_offsets,
                    const cpx_frame_meta* meta, int B, cpx_component* comps_dev,
                    cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                    float* background_dev);

/* ---- association stage: region filter + track matching + Kalman ------------
 * Replaces, for the same batch, one GPU lane per clip,
 *   ClipTracker._get_regions_of_interest      (track/cliptracker.py:263-365)
 *   ClipTracker._apply_region_matchings etc.  (track/cliptracker.py:124-247)
 *   RegionTracker / Track bookkeeping, Kalman (track/track.py:107-326,646-735; track/kalman.py)
 * Inputs are the outputs of cpx_track_batch (same clip_offsets / meta).
 * pool_dev     cpx_region [total_frames * max_active_tracks]  track histories
 * tracks_dev   cpx_track_record [B * max_tracks], n_tracks_dev int32 [B]
 * status_dev   int32 [B]: 0 or CPX_ERR_OVERFLOW (more simultaneous / total tracks than capacity, or a frame of the clip
 *              with more components than max_components); n_tracks_dev of such a clip is 0: run it again on larger
 *              tables (the reference has no limits: cliptracker.py:202-247)
 * regions_dev  cpx_region [total_frames * max_components] or NULL (clip.region_history),
 * region_counts_dev int32 [total_frames] or NULL
 */
int cpx_associate_batch(cpx_handle* h, const cpx_track_params* params, const int32_t* clip_offsets,
                        const cpx_frame_meta* meta, int B, const cpx_component* comps_dev,
                        const cpx_frame_info* info_dev, cpx_region* pool_dev,
                        cpx_track_record* tracks_dev, int32_t* n_tracks_dev, int32_t* status_dev,
                        cpx_region* regions_dev, int32_t* region_counts_dev);

/* ---- end of clip: trim, movement statistics, score, rejects (one GPU lane per clip) ----------
 * Replaces Track.trim / calculate_stats (track/track.py:737-905) and ClipTracker.filter_tracks /
 * filter_track (track/cliptracker.py:367-486) for every track cpx_associate_batch produced. */
typedef struct cpx_filter_params {
  double min_duration_secs, track_min_offset, track_min_mass, track_min_delta, track_max_delta;
  int32_t min_moving_frames, max_blank_percent, max_jitter, fps;
  int32_t max_tracks; /* -1: None */
  int32_t max_active_tracks, max_tracks_per_clip; /* the capacities used by cpx_associate_batch */
  int32_t reserved;
} cpx_filter_params;

enum { CPX_TRACK_KEPT = 0, CPX_REJECT_TOO_SHORT = 1, CPX_REJECT_DIDNT_MOVE = 2, CPX_REJECT_TOO_MANY_BLANKS = 3,
       CPX_REJECT_TOO_JITTERY = 4, CPX_REJECT_TOO_STATIC = 5, CPX_REJECT_TOO_DYNAMIC = 6,
       CPX_REJECT_MASS_TOO_SMALL = 7, CPX_REJECT_TOO_MANY_TRACKS = 8 };

typedef struct cpx_track_summary {
  int32_t id, slot, start_frame, n_frames; /* after trim */
  int32_t blank_frames, since_seen;        /* RegionTracker counters after trim */
  int32_t reject;                          /* CPX_TRACK_KEPT or a CPX_REJECT_* reason */
  int32_t rank;                            /* position in the clip's score order (stable, descending) */
  int32_t frames_moved, region_jitter, jitter_bigger, jitter_smaller, blank_percent;
  int32_t n_segments;                      /* classification segments planned for it (kept tracks) */
  double movement, max_offset, score, average_mass, median_mass, delta_std, mass_std, average_velocity;
} cpx_track_summary;

/* summaries_dev: [B * max_tracks_per_clip] in creation order (parallel to tracks_dev);
 * counts_dev: int32 [B][4] = kept tracks, region refs, samples (segments), reserved -- per clip. */
int cpx_finalize_tracks(cpx_handle* h, const cpx_filter_params* params, const int32_t* clip_offsets,
                        const cpx_frame_meta* meta, int B, const cpx_region* pool_dev,
                        const cpx_track_record* tracks_dev, const int32_t* n_tracks_dev,
                        cpx_track_summary* summaries_dev, int32_t* counts_dev);

/* ---- classification pre-processing: limits + crop / resize / normalise / tile ---------
 * Replaces Interpreter.get_limits / preprocess_segments (ml_tools/interpreter.py:315-474),
 * preprocess_frame (ml_tools/preprocess.py:56-113), resize_and_pad / resize_cv
 * (ml_tools/imageprocessing.py:11-82) and preprocess_movement / square_clip
 * (ml_tools/preprocess.py:151-202, ml_tools/imageprocessing.py:85-104) for channels
 * (thermal, filtered) with diff_norm = True, thermal_diff_norm = False, keep_edge = True.
 */
typedef struct cpx_region_ref { /* one non-blank region of a track */
  int32_t frame;                /* index into frames_dev / filtered_dev / info_dev */
  int32_t x, y, width, height;
  int32_t in_segment;           /* 1: the frame is used by a segment (clip_thermals_at_zero test) */
} cpx_region_ref;

typedef struct cpx_track_limits { /* per track */
  float filt_min, filt_max;       /* filtered_norm_limits: min / max over the track's region crops, max >= 0 */
  int32_t clip_at_zero;           /* clip_thermals_at_zero (interpreter.py:372-399) */
  int32_t flags;                  /* CPX_LIMITS_* the limits were computed with: cpx_crop_tile follows them */
  float therm_min, therm_max;     /* thermal_norm_limits (CPX_LIMITS_THERMAL_DIFF_NORM): min / max of thermal - median
                                     over the whole frames of the track's regions (interpreter.py:339-346) */
  int32_t reserved[2];
} cpx_track_limits;

/* cpx_track_limits_batch_ex flags.  CPX_LIMITS_POST_PROCESS = ClipClassifier.post_process_file
 * (classify/clipclassifier.py:472-497,525-538): limits over the frames the segments use only (refs with in_segment,
 * no floor of 0 on the maximum), thermals always clipped at zero, and the frame median subtracted from the thermal
 * crop BEFORE it is resized (preprocess_frame(cropped=True, sub_median=False) on a crop that already had it
 * subtracted) instead of after. */
#define CPX_LIMITS_POST_PROCESS 1
/* The normalisation variants of preprocess_frame (ml_tools/preprocess.py:56-113) a model's hyper-parameters select
 * (ml_tools/hyperparams.py): thermal_diff_norm -> the thermal tile is NOT clipped at zero and is normalised with the
 * track's thermal limits; diff_norm = False -> no limits are applied, both channels are normalised per tile
 * (Frame.normalize, ml_tools/frame.py:187-191; the thermal limits are then ignored too, as in the reference);
 * ALWAYS_CLIP -> clip_thermals_at_zero = True without the median test: what preprocess_frames passes for
 * single-frame models (interpreter.py:255-313); SWAP_CHANNELS -> channels = (filtered, thermal). */
#define CPX_LIMITS_THERMAL_DIFF_NORM 2
#define CPX_LIMITS_NO_DIFF_NORM 4
#define CPX_LIMITS_ALWAYS_CLIP 8
#define CPX_LIMITS_SWAP_CHANNELS 16
/* the finished sample is scaled x / 127.5 - 1 (float32): the preprocess_fn of inceptionv3 and of the Keras families whose
 * preprocess_input runs in 'tf' mode -- nasnet, resnetv2, mobilenet, inceptionresnetv2 (ml_tools/interpreter.py:64-98,
 * 563-566; applied by preprocess_movement / preprocess_single_frame, ml_tools/preprocess.py:142-143,200-201) */
#define CPX_LIMITS_TF_SCALING 32

typedef struct cpx_crop_req { /* one tile = one frame of one segment */
  int32_t frame;
  int32_t x, y, width, height; /* the track's region in that frame */
  int32_t track;               /* index into limits_dev */
  int32_t sample;              /* output sample (segment) index */
  int32_t tile;                /* 0 .. square_width^2 - 1, row-major */
} cpx_crop_req;

/* Exclusive prefix sums of cpx_finalize_tracks' per-clip work counts (counts_dev int32 [B][4]: kept tracks, region
 * refs, samples, spare) -> prefix_dev int32 [B + 1][4], row B = the totals: the addresses cpx_plan_segments writes a
 * clip's tracks / refs / samples to (the reference appends them to Python lists in clip order, interpreter.py:178-253,
 * clipclassifier.py:252-303).  One small launch on the handle's stream. */
int cpx_counts_prefix(cpx_handle* h, const int32_t* counts_dev, int B, int32_t* prefix_dev);

/* Plans the classification work of a batch on the device: get_segments(SegmentType.ALL_RANDOM_MASKED)
 * (ml_tools/datasetstructures.py:972-1301) as it plans when every random draw is the identity -- np.random.shuffle
 * leaves the order, choice(replace=False) takes the first k, choice(replace=True) cycles (the reference draws at
 * random; this is the member of that family a test can pin): usable frames (not blank, not FFC-affected,
 * mass > 0) in frame order, masked windows of square_width^2 frames at the segment spacing for tracks of >= 40 usable
 * frames, the short remainder padded both ways as the reference pads, segments whose mass falls below the track's
 * threshold dropped.  tests/golden/segments_identity_golden.json holds the reference's own output under these draws.
 * Inputs: outputs of cpx_associate_batch / cpx_finalize_tracks and, per clip, exclusive prefix sums
 * (device int32 [B][4]: cpx_counts_prefix) of counts_dev.  Fills refs / track offsets (one entry more than kept tracks:
 * the last one closes the last track's refs) / crop requests / per-sample track index for cpx_track_limits_batch and
 * cpx_crop_tile; track_clip_dev[t] = (clip, track id). */
int cpx_plan_segments(cpx_handle* h, const cpx_filter_params* params, const int32_t* clip_offsets,
                      const cpx_frame_meta* meta, int B, const cpx_region* pool_dev,
                      const cpx_track_summary* summaries_dev, const int32_t* n_tracks_dev,
                      const int32_t* prefix_dev, int square_width, struct cpx_region_ref* refs_dev,
                      int32_t* track_offsets_dev, struct cpx_crop_req* reqs_dev, int32_t* sample_track_dev,
                      int32_t* track_clip_dev);

/* refs_dev: regions of all tracks, track t owns [track_offsets[t], track_offsets[t+1]) (device arrays). */
int cpx_track_limits_batch(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                           const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev,
                           const int32_t* track_offsets_dev, int n_tracks, cpx_track_limits* limits_dev);
int cpx_track_limits_batch_ex(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                              const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev,
                              const int32_t* track_offsets_dev, int n_tracks, cpx_track_limits* limits_dev, int flags);

/* out_dev: float [n_samples, square_width*frame_size, square_width*frame_size, 2] (NHWC; thermal, filtered).
 * Every tile of every sample must be covered by exactly one request. */
int cpx_crop_tile(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                  const cpx_frame_info* info_dev, const cpx_crop_req* reqs_dev, int n_reqs,
                  const cpx_track_limits* limits_dev, int frame_size, int square_width, float* out_dev);

/* Per-track aggregation of the segment predictions (classify/trackprediction.py:127-171 with
 * smooth_predictions = False: sum over the track's segments, normalised to sum 1) plus the low-evidence
 * cap of Interpreter.track_prediction_from_raw (ml_tools/interpreter.py:151-168).  Samples of a track are
 * contiguous (cpx_plan_segments).  scores_dev float [n_tracks, L]; best_dev int32 [n_tracks] (argmax). */
int cpx_aggregate_predictions(cpx_handle* h, const float* probs_dev, const int32_t* sample_track_dev,
                              int n_samples, const struct cpx_crop_req* reqs_dev, int n_tracks, int n_labels,
                              int false_positive_index, int square_width, float* scores_dev, int32_t* best_dev);

/* ---- thumbnail stage (classify/thumbnail.py:13-188) -------------------------------
 * cpx_thumb_stats replaces the per-region body of get_track_thumb_stats (thumbnail.py:76-134): for each
 * region the external contours of np.uint8(region.subimage(frame.mask)) with CHAIN_APPROX_TC89_L1
 * (cv2.findContours) -> point count of the longest one, and np.median(thermal under the mask) -
 * np.median(frame.thermal).  The caller passes only non-blank regions with mass > 0 (thumbnail.py:78-79);
 * scoring / ranking (thumbnail.py:138-197) is a few flops per region and stays with the caller.
 * labels_dev is cpx_track_batch's labels output; refs_dev[i].in_segment is ignored.
 * out_dev[i].contours == 0: no contour in the region (the reference skips the frame). */
typedef struct cpx_thumb_stat {
  int32_t contours;    /* len(contours[0]) after sorting by length, 0 if none */
  int32_t status;      /* 0 or CPX_ERR_OVERFLOW (a border longer than the kernel's chain capacity) */
  double median_diff;  /* masked_median - t_median (thumbnail.py:117-120) */
} cpx_thumb_stat;

int cpx_thumb_stats(cpx_handle* h, const uint16_t* frames_dev, const int32_t* labels_dev,
                    const cpx_frame_info* info_dev, const struct cpx_region_ref* refs_dev, int n_refs,
                    cpx_thumb_stat* out_dev);
/* The same with the kernel's per-region scratch sized by the caller: one wavefront per region keeps the region's mask
 * ((max_width + 2) x (max_height + 2) bytes) and the border chain (8 bytes per step, chain_capacity steps) in LDS, and
 * the number of regions a CU works on at once is what fits its 160 KB -- cpx_thumb_stats sizes for a whole frame and
 * 8192 steps (one region per CU); 46 x 46 and 2304 steps (cpx/engine.py) runs seven.  A region larger than
 * max_width x max_height, or a border longer than chain_capacity, reports CPX_ERR_OVERFLOW in its status (nothing
 * else of it is written): run it again through the wider form. */
int cpx_thumb_stats_ex(cpx_handle* h, const uint16_t* frames_dev, const int32_t* labels_dev,
                       const cpx_frame_info* info_dev, const struct cpx_region_ref* refs_dev, int n_refs,
                       cpx_thumb_stat* out_dev, int max_width, int max_height, int chain_capacity);

/* best_trackless_thumb's window search (thumbnail.py:26-64) on frame `frame` with the clip background
 * frames_dev[background] (the clip's first frame, clip.py:152-158): every 64x64 window position of
 * range(H-64) x range(W-64), mean of the frame and of the uint16-wrapping difference frame - background,
 * walked in raster order with the reference's update rule.  out_dev int32[2] = x, y of the chosen window. */
int cpx_trackless_thumb(cpx_handle* h, const uint16_t* frames_dev, int frame, int background, int32_t* out_dev);
/* The same search for n recordings in one launch (one workgroup each): pairs_dev int32 [n][2] = frame, background
 * indices into frames_dev; out_dev int32 [n][2]. */
int cpx_trackless_thumb_batch(cpx_handle* h, const uint16_t* frames_dev, const int32_t* pairs_dev, int n, int32_t* out_dev);

/* ---- IR detection stage (partial) ------------------------------------------------------
 * Replaces detect_objects_ir (ml_tools/imageprocessing.py:185-199) for n frames of `width` x `height` uint8 pixels (the
 * foreground image a background subtractor produced; 640 x 480 in the reference): MORPH_OPEN with the reference's
 * tuple kernel, threshold (pixel > threshold), 8-connected components with OpenCV's statistics and numbering.
 * width must be a multiple of 64 (<= 640), height <= 480.  comps_dev [n][max_components] in label order (label i+1 =
 * entry i; centroid = sum / area), counts_dev [n], status_dev [n] = 0 or CPX_ERR_OVERFLOW (the frame has more than
 * max_components components; counts_dev then holds the true count and nothing else of the frame is written);
 * labels_dev optional int32 [n, height, width].  Frames with up to 8192 pixel runs and 1024 components are labelled
 * entirely in LDS; busier ones use scratch the handle allocates (up to 256 slots of ~3 MB at 640 x 480).  The IR background model (cpx_mog2_*) and the merge of fragments
 * (cpx_ir_merge; host form in cpx/track/irdetect.py) are calls of their own. */
int cpx_ir_detect(cpx_handle* h, const uint8_t* images_dev, int n_frames, int width, int height, int threshold,
                  int max_components, cpx_component* comps_dev, int32_t* counts_dev, int32_t* status_dev,
                  int32_t* labels_dev);

/* cv2.resize(src, (width / factor, height / factor), interpolation=cv2.INTER_AREA) of n uint8 images for an integer
 * factor that divides both sides: the down-scaled foreground the IR tracker detects objects in when it is created with
 * a `scale` (track/irtrackextractor.py:445-451, scale = 1 / factor; piclassifier.py:225 runs 0.25).  OpenCV's
 * integer-ratio area filter: the block mean, (sum + 2) >> 2 at factor 2, round-half-even of sum / factor^2 in float32
 * otherwise (restated from OpenCV, which is not in this container: parity with cv2 unpinned, as for cpx_mog2_*).
 * dst_dev uint8 [n][height / factor][width / factor].  Other ratios: CPX_ERR_UNSUPPORTED. */
int cpx_ir_resize_area(cpx_handle* h, const uint8_t* src_dev, int n, int width, int height, int factor, uint8_t* dst_dev);

/* The two steps between cpx_ir_detect and the association, on the device, for n videos advancing in lockstep:
 * merge_components (track/irtrackextractor.py:324-389: fragments of one object merged into one box -- rows with area
 * > 40 or both sides > 16 survive, largest first; a row absorbs rows closer than 40 pixels to, or overlapping, its
 * original box; the scan restarts after a merge) and the variance of the frame difference over every merged box
 * (cpx_ir_delta_variance).  comps_dev [n][cap_in] / counts_dev [n]: what cpx_ir_detect wrote for the n frames;
 * cur_dev / prev_dev uint8 [n][height][width]: the frames and the frames they are compared with (prev_dev NULL: no
 * comparison yet, variance 0).  Stream v's result goes to row v * out_stride + frame_number of out_comps_dev (rows of
 * cap_out components: x, y, width, height, area = merged box and mass, sum_x / sum_y = the truncated box centre times
 * the mass -- the IR tracker's centroid --, pixel_variance) and of out_info_dev (frame_number, n_components; optional):
 * the clip-major layout cpx_associate_batch reads, so that one association call follows the last step.  status_dev [n]:
 * CPX_ERR_OVERFLOW when more than cap_out rows survive the small-fragment filter. */
int cpx_ir_merge(cpx_handle* h, const cpx_component* comps_dev, const int32_t* counts_dev, int n, int cap_in, int cap_out,
                 const uint8_t* cur_dev, const uint8_t* prev_dev, int width, int height, int frame_number, int out_stride,
                 cpx_component* out_comps_dev, cpx_frame_info* out_info_dev, int32_t* status_dev);

/* Per-region variance of the frame-to-frame change the IR tracker gates regions with: replaces
 * np.var(np.abs(frame.thermal - frame_ago.thermal)[region]) (track/irtrackextractor.py:638-655 get_delta_frame,
 * track/cliptracker.py:303-312).  Both frames are uint8, so the difference wraps modulo 256 as NumPy's does.
 * rects_dev int32 [n][4] = x, y, width, height (clipped to the image like a NumPy slice); var_dev double [n]. */
int cpx_ir_delta_variance(cpx_handle* h, const uint8_t* cur_dev, const uint8_t* prev_dev, int width, int height,
                          const int32_t* rects_dev, int n, double* var_dev);

/* Per-frame statistics of an IR clip, for n uint8 frames of `pixels` pixels resident in HBM: what Clip.add_frame
 * (track/clip.py:330-347: np.min / np.max / np.median / np.nanmean of the frame, np.sum(np.abs(filtered))) computes on
 * the host.  median_x2 = twice np.median (the mean of the two middle order statistics of an even count);
 * mean = sum / pixels on the caller's side; filtered_sum = the sum of masks_dev's bytes (0 when masks_dev is NULL).
 * hist_dev: uint32 scratch [n][256] (the 256-bin histograms the statistics come from; zeroed by the call). */
typedef struct cpx_ir_frame_stats {
  int32_t min, max;
  int64_t sum;
  int32_t median_x2, reserved;
  int64_t filtered_sum;
} cpx_ir_frame_stats; /* 32 bytes */
int cpx_ir_frame_statistics(cpx_handle* h, const uint8_t* frames_dev, const uint8_t* masks_dev, int n, int pixels,
                            uint32_t* hist_dev, cpx_ir_frame_stats* out_dev);

/* ---- IR background model ---------------------------------------------------------------
 * Replaces CVBackground (track/cliptracker.py:561-613): cv2.createBackgroundSubtractorMOG2(history, varThreshold,
 * detectShadows=False) for `n_streams` independent 8-bit single-channel videos of `width` x `height` advancing in
 * lockstep (one frame each per call).  cpx_mog2_apply = .apply(frame, None, learning_rate): frames_dev uint8
 * [n_streams, height, width] -> fgmask_dev uint8 (0 / 255), the mixture state in the object is updated; a negative
 * learning_rate (and the first frame) uses 1 / min(2 * frames seen, history), as OpenCV does.
 * cpx_mog2_background = .getBackgroundImage() -> out_dev uint8 [n_streams, height, width].
 * The algorithm is OpenCV's (un-vendored: opencv-contrib-python-headless~=4.12, bgfg_gaussmix2.cpp), restated from
 * the published update: parity with cv2 itself is unpinned (no cv2 / golden here); the HIP kernel is bit-equal to
 * oracle/mog2_oracle.c.  The object belongs to its handle (freed by cpx_destroy if still alive). */
typedef struct cpx_mog2 cpx_mog2;
int cpx_mog2_create(cpx_handle* h, int n_streams, int width, int height, int history, float var_threshold,
                    cpx_mog2** out);
int cpx_mog2_apply(cpx_mog2* m, const uint8_t* frames_dev, double learning_rate, uint8_t* fgmask_dev);
int cpx_mog2_background(cpx_mog2* m, uint8_t* out_dev);
void cpx_mog2_destroy(cpx_mog2* m);

/* ---- CNN forward building blocks (WR-ResNet, ml_tools/resnet/wr_resnet.py:5-98) -----------
 * Replaces tf.keras Conv2D(groups) / BatchNormalization / Activation / Add / GlobalAveragePooling2D /
 * Dense as used by KerasModel.predict (ml_tools/kerasmodel.py:856-859).  Activations NHWC float32.
 * out = relu?( conv(relu?(in * in_scale + in_shift)) * out_scale + out_shift + residual )
 * weights_dev is packed [groups][ksize*ksize][Cin/groups][Cout/groups] (Keras HWIO regrouped).
 * pad_same: 1 = TensorFlow "SAME" (extra padding at the bottom / right), 0 = "valid". */
typedef struct cpx_conv_desc {
  int32_t N, H, W, Cin, Cout, groups, ksize, stride, pad_same, relu;
  const float* in_dev;
  float* out_dev;
  const float* weights_dev;
  const float* in_scale_dev;  /* [Cin] or NULL: BatchNorm + ReLU applied to the input */
  const float* in_shift_dev;
  const float* out_scale_dev; /* [Cout] or NULL */
  const float* out_shift_dev; /* [Cout] or NULL (bias, or bias folded with the following BatchNorm) */
  const float* residual_dev;  /* [N,Ho,Wo,Cout] or NULL */
} cpx_conv_desc;
int cpx_conv2d(cpx_handle* h, const cpx_conv_desc* desc);
/* How the float32 multiplications of the 3x3 stride-1 convolutions (>= 16 input channels per group) are carried out.
 * Both modes take and return float32, accumulate in float32 and meet the same tolerance against a float64
 * convolution (tests/test_cnn_gpu.py); every other layer always uses the float32 instruction.
 *   CPX_CNN_MATH_F32     v_mfma_f32_32x32x2_f32 on the operands as they are
 *   CPX_CNN_MATH_BF16X3  each operand split exactly into three bf16 terms, six v_mfma_f32_32x32x16_bf16 per K step
 *                        (the cross terms below 2^-26 of a product are dropped); default, 2.67x the MFMA rate
 *   CPX_CNN_MATH_BF16X2  the stride-1 layers with 32 / 64 / 128 channels per group (stages 2, 3 and 4 of
 *                        WR-ResNet-22-4) and the stride-2 first convolution of stage 3
 *                        take each operand as TWO bf16 terms rounded
 *                        to nearest (16 significand bits, relative error <= 2^-16 per operand: 32 times finer than
 *                        TF32) and three products per K step; every other layer as BF16X3.  Logits within 1e-6 of the float32 forward on the test network
 *                        (tests/test_cnn_gpu.py, same 2e-4 bound as the other modes)
 *   CPX_CNN_MATH_FP16X2  the same layers as BF16X2 take each operand as TWO fp16 terms rounded to nearest (11 + 11
 *                        significand bits and the sign of the remainder: relative error <= 2^-22 per operand, the
 *                        dropped lo x lo term <= 2^-22 of a product -- the level of the float32 accumulation error every
 *                        mode carries) and three products per K step on v_mfma_f32_*_f16, the bf16 forms' rate.  fp16
 *                        has a range, so the operands are scaled by powers of two (exact; undone in the epilogue): the
 *                        weights per output channel when their image is built, the activated input per layer by a
 *                        power of two a network derives from its BatchNorm parameters
 *                        (cpx_cnn_set_activation_bounds; 1 for a bare cpx_conv2d).  A scaled activation beyond fp16's
 *                        largest finite value never saturates silently: the kernel raises a device-side word and
 *                        the layer -- inside cpx_cnn_forward: the rest of its residual BLOCK, one word per block -- is
 *                        run again by the BF16X3 kernel, which is launched behind every fp16 launch and returns at
 *                        once while the word is clear (cpx_cnn_last_overflow reads whether any block did).
 *                        Inside cpx_cnn_forward the residual blocks whose convolutions are stride 1 with 32 output
 *                        channels per group (stage 2 of WR-ResNet-22-4) run as ONE launch each, the tensor between
 *                        their two convolutions kept on chip (environment CPX_CNN_BLOCK_FUSION=0|1|2, default 2).
 *                        Every other layer as BF16X3.
 *                        Two conditions on the 2^-22 figure and on the rerun, both met inside cpx_cnn_forward and the
 *                        caller's to meet for a bare cpx_conv2d (activation scale 1): (1) the low plane is an fp16
 *                        too -- an activated input below 2^-3 (times the layer's scale) keeps fewer than 11 low bits,
 *                        absolute error up to 2^-25, and a value below about 3e-8 contributes nothing; a caller whose
 *                        activations are that small selects BF16X3; (2) no aliasing: out_dev must not be residual_dev
 *                        or in_dev.  cpx_conv2d detects that case and runs the layer as BF16X3 directly (the rerun would
 *                        otherwise add a residual the fp16 pass has already overwritten).
 * The default can be preset with the environment variable CPX_CNN_MATH=f32|bf16x3|bf16x2|fp16x2 (read by cpx_create). */
#define CPX_CNN_MATH_F32 0
#define CPX_CNN_MATH_BF16X3 1
#define CPX_CNN_MATH_BF16X2 2
#define CPX_CNN_MATH_FP16X2 3
int cpx_set_cnn_math(cpx_handle* h, int mode);
int cpx_get_cnn_math(const cpx_handle* h);
/* CPX_CNN_MATH_FP16X2: *overflowed = 1 when the last cpx_cnn_forward (or bare cpx_conv2d) on this handle met an
 * activation outside fp16's range and fell back to the BF16X3 kernels (results are correct either way; this is for
 * tests and for whoever wonders where the time went).  Synchronises the handle's stream. */
int cpx_cnn_last_overflow(cpx_handle* h, int* overflowed);
/* ... and how many cpx_cnn_forward calls on this handle did so since the handle was created (or since the last call with
 * reset != 0): a network whose activation bounds do not fit its data pays the rerun often, and this is where that shows. */
int cpx_cnn_overflow_forwards(cpx_handle* h, int* count, int reset);
/* relu(in * bn_scale + bn_shift) -> mean over H*W -> dense [C][L] + bias -> logits (and sigmoid probs if not NULL) */
int cpx_cnn_head(cpx_handle* h, const float* in_dev, int N, int HW, int C, const float* bn_scale_dev,
                 const float* bn_shift_dev, const float* dense_w_dev, const float* dense_b_dev, int L,
                 float* logits_dev, float* probs_dev);
/* The head as KerasModel.build_model can make it (ml_tools/kerasmodel.py:308-350): after the pooling up to
 * CPX_HEAD_MAX_HIDDEN Dense(size, relu) layers (hyperparams.dense_sizes), then Dense(n_labels) with sigmoid
 * (multi_label, the default) or softmax.  hidden weights [in][out] row-major, sizes <= 2048. */
#define CPX_HEAD_MAX_HIDDEN 4
#define CPX_HEAD_SIGMOID 0
#define CPX_HEAD_SOFTMAX 1
typedef struct cpx_head_desc {
  int32_t N, HW, C, L;
  int32_t n_hidden, activation;
  int32_t hidden_sizes[CPX_HEAD_MAX_HIDDEN];
  const float* in_dev;        /* [N, HW, C] */
  const float* bn_scale_dev;  /* final BatchNorm folded to scale / shift, [C] */
  const float* bn_shift_dev;
  const float* hidden_w_dev[CPX_HEAD_MAX_HIDDEN];
  const float* hidden_b_dev[CPX_HEAD_MAX_HIDDEN];
  const float* dense_w_dev;   /* [last hidden size or C][L] */
  const float* dense_b_dev;
  float* logits_dev;          /* [N, L] pre-activation */
  float* probs_dev;           /* [N, L] or NULL */
} cpx_head_desc;
int cpx_cnn_head_ex(cpx_handle* h, const cpx_head_desc* desc);

/* ---- whole-network forward ------------------------------------------------------------------------------
 * KerasModel.predict (ml_tools/kerasmodel.py:856-859) for the WR-ResNet classifier in one call: conv1, three
 * stages of `blocks_per_stage` pre-activation blocks (+ 1x1 shortcut in the first block of a stage, strides
 * 1 / 2 / 3, wr_resnet.py:5-98), final BatchNorm + ReLU, global average pooling, Dense(n_labels) + sigmoid
 * (kerasmodel.py:308-350).  The parameter struct holds device pointers (float32) laid out as cpx_conv2d /
 * cpx_cnn_head take them; BatchNorm layers arrive folded to scale / shift.  The activation buffers belong to
 * the handle (grown to the largest N seen, shared by every network created on it: forwards on a handle are
 * serialised on its stream); a call enqueues 3 + 6 * 3 + 1 kernels on the handle's stream. */
typedef struct cpx_wrresnet_block {
  const float* in_scale;  /* BatchNorm 2a (applied with ReLU to the block input) */
  const float* in_shift;
  const float* wa;        /* conv 2a 3x3, packed */
  const float* a_scale;   /* conv 2a bias folded with BatchNorm 2b */
  const float* a_shift;
  const float* wb;        /* conv 2b 3x3, packed */
  const float* bb;        /* conv 2b bias */
} cpx_wrresnet_block;

#define CPX_WRRESNET_MAX_BLOCKS 8
typedef struct cpx_wrresnet_params {
  int32_t n_labels, blocks_per_stage, groups, in_channels;
  int32_t filters[4];     /* conv1, stage 2, 3, 4 */
  const float* conv1_w;
  const float* conv1_b;
  cpx_wrresnet_block block[3][CPX_WRRESNET_MAX_BLOCKS];
  const float* shortcut_w[3];
  const float* shortcut_b[3];
  const float* final_scale;
  const float* final_shift;
  const float* dense_w;   /* [filters[3] or the last hidden size][n_labels] */
  const float* dense_b;
  /* head variants (ml_tools/kerasmodel.py:337-345): hidden Dense(relu) layers, sigmoid or softmax output */
  int32_t n_hidden, activation; /* CPX_HEAD_SIGMOID / CPX_HEAD_SOFTMAX */
  int32_t hidden_sizes[CPX_HEAD_MAX_HIDDEN];
  const float* hidden_w[CPX_HEAD_MAX_HIDDEN];
  const float* hidden_b[CPX_HEAD_MAX_HIDDEN];
} cpx_wrresnet_params;

/* A network belongs to the handle it was created on: cpx_destroy(h) frees the networks still alive, after which
 * their pointers are invalid (do not call cpx_cnn_destroy on them). */
typedef struct cpx_cnn cpx_cnn;
int cpx_cnn_create(cpx_handle* h, const cpx_wrresnet_params* params, cpx_cnn** out);
void cpx_cnn_destroy(cpx_cnn* cnn);
/* CPX_CNN_MATH_FP16X2: an upper bound of the ACTIVATED input of each 3x3 convolution of the blocks, in launch order
 * ([stage][block][branch2a, branch2b]: n = 3 * blocks_per_stage * 2 values) -- e.g. max over channels of
 * |beta| + 64 |gamma| of the BatchNorm in front of it (the folded scale / shift the parameter struct carries no longer
 * say).  The network multiplies that input by the largest power of two (<= 2^14) that keeps the bound at or below 2^12
 * (sixteen-fold headroom below fp16's largest value: a bound from statistics is a guess) before the fp16 split, so that
 * small activations keep their low plane's bits; a bound that turns out too small costs
 * time (the overflow rerun), never correctness.  Without this call the scale is 1. */
int cpx_cnn_set_activation_bounds(cpx_cnn* cnn, const float* bounds, int n);
/* in_dev float32 [N, H, W, in_channels] (NHWC, values 0..255) -> logits_dev [N, n_labels] and, when not NULL,
 * probs_dev (sigmoid). */
int cpx_cnn_forward(cpx_cnn* cnn, const float* in_dev, int N, int H, int W, float* logits_dev, float* probs_dev);

/* Per-kernel timing of cpx_conv2d launches with HIP events on the handle's stream (bench.py's roofline).
 * cpx_conv_timing_enable(h, 1) starts collecting (and clears); cpx_conv_timing_report synchronises and
 * returns, per kernel variant, total ms / launches / algorithmic FLOPs.  key = Cin_g*10000 + Cout_g*10 + stride
 * (+ 5 for 1x1 kernels). */
typedef struct cpx_conv_timing {
  int32_t key, launches;
  double total_ms, flops;
} cpx_conv_timing;
int cpx_conv_timing_enable(cpx_handle* h, int enable);
int cpx_conv_timing_report(cpx_handle* h, cpx_conv_timing* out, int cap, int* n_out);

/* Bytes of device workspace cpx_track_batch needs for B clips / total frames
 * (allocated lazily inside the handle and reused). */
size_t cpx_track_workspace_bytes(const cpx_handle* h, int B, int total_frames);

/* Duration in ms of the frame kernel launches of the last cpx_track_batch
 * (HIP events on the handle's stream; valid after cpx_synchronize) and their count:
 * 1 when one workgroup per clip walks all of the clip's frames (the default without
 * denoise), one (or three, with denoise) per frame step otherwise. */
int cpx_last_kernel_timing(cpx_handle* h, float* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* CPX_H */
