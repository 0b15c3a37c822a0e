// cpx_kernels.h -- internal interface between the C-ABI translation unit
// (cpx_api.cpp) and the HIP kernels.  Not part of the public ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "cpx.h"

// Tunables of the track kernel (one workgroup per clip-frame).
// 768 threads (round 6; 1,024 before): two workgroups per CU either way (82 KB of LDS each), i.e. 24 waves = 6 per SIMD
// instead of 8, which raises the register budget from 64 to 80 per lane -- the kernel takes 70 and no longer spills
// (ScratchSize 36 -> 0 B per lane; 401 -> 385 us per frame step of 4,096 clips, scratch/track_ab.sh; 896 threads put
// only one workgroup on a CU (562 us), 640 and 512 were slower: 467 and 409 us)
#ifndef CPX_TRACK_THREADS
#define CPX_TRACK_THREADS 768
#endif
#ifndef CPX_TRACK_CHUNKS
#define CPX_TRACK_CHUNKS 7  // 4-pixel chunks per thread: W*H <= 4*7*768 = 21504
#endif
#ifndef CPX_TRACK_LDS_COMPONENTS
#define CPX_TRACK_LDS_COMPONENTS 256
#endif
#ifndef CPX_TRACK_MIN_WAVES_PER_SIMD
#define CPX_TRACK_MIN_WAVES_PER_SIMD 6  // __launch_bounds__ 2nd argument (waves per SIMD)
#endif

namespace cpx {

typedef cpx_component Component;
typedef cpx_frame_info FrameInfo;

struct ClipState {
  double bg_average;  // WeightedBackground.average (float until the first change, then integral)
  int prev_fmin, prev_fmax;  // min / max of the previous frame's filtered image
  int has_prev;
  int n_done;  // frames processed since the state was seeded: the final background sits in ping-pong slot n_done & 1
};

// block-reduced scalars of the streaming pass
struct Red1 {
  unsigned int sumpix;
  unsigned int minpix, maxpix;
  int fmin, fmax;
  unsigned int sumbg;
  unsigned int changed;
  unsigned long long sumabs;
};

// scalars that cross the front / back split of a frame step (denoise)
struct FrameCarry {  // front half -> (NLM ->) back half of a split frame step; two slots per clip (t & 1)
  Red1 R;
  int avg_change, mn, mx, ithr;
  float thresh, median;
  int pad;
  double bg_avg_in;  // background average the front half started from
};

struct TrackArgs {
  // geometry / config
  int W, H, edge, window, cap_out;
  int flags;                // CPX_TRACK_* of the call (include/cpx.h)
  double background_thresh;
  double weight_add;
  // inputs
  const uint16_t* frames;   // [total_frames, H, W]
  const int* clip_first;    // [B]   first frame (file order) of each clip: background init
  const int* proc_off;      // [B+1] offsets into proc_idx
  const int* proc_idx;      // [total_proc] frame index (into frames) of each processed frame
  const int* proc_ffc;      // [total_proc] is_affected_by_ffc
  const int* order;         // [B] clips by falling number of processed frames (workgroup index -> clip), or nullptr
  const double* wtab;       // [max_frames+2] k-fold float64 accumulation of weight_add
  int wtab_len;             // entries of wtab / wthr
  const uint32_t* wthr;     // [wtab_len] 2 hi_k - near_k: hi_k = floor(w_k) + 1; near_k: w_k within 1e-6 of an integer, not equal to it
  // per-clip state
  int nlm_flip;             // 1: a denoiser wrote the hand-over image into the other slot (back half reads that)
  double* bgavg;            // [B] background average after the last front half (split steps only)
  uint16_t* bg;             // [B][2][P] ping-pong background (interior authoritative); the background is a
                            // floor of a mean of uint16 frames, so 16 bits hold it exactly
  uint32_t* wsum;           // [B][P] sum of the last <= window frames
  uint16_t* kcnt;           // [B][P] consecutive "background kept" count -> weight = wtab[k]
  int packed_state;         // 1: this call keeps that count in wsum's top ten bits and leaves kcnt alone (cpx_frame_kernel<true>)
  float* filt_state;        // [B][2][P] ping-pong filtered (only when filtered_out == nullptr)
  ClipState* cstate;        // [B]
  unsigned char* u8_state;  // [B][P] normalised uint8 image between front / NLM / back (denoise only)
  FrameCarry* carry;        // [B]
  const int* nlm_lut;       // [64] fixed-point NLM weights (denoise only)
  uint32_t* big_stat;       // [B][9][cap_out] component tables in HBM for frames beyond the LDS tables, or nullptr
  // outputs
  Component* comps_out;     // [total_frames * cap_out]
  FrameInfo* info_out;      // [total_frames]
  int32_t* labels_out;      // [total_frames, P] or nullptr
  float* filtered_out;      // [total_frames, P] or nullptr
};

struct ActiveTrack;
struct ScoreRec;
struct AssocResume {  // per clip, carried between cpx_associate_frame calls
  int n_active, n_tracks, next_id, status;
};
struct AssocArgs {
  int B, cap;
  int t_begin;          // first processed-frame index to handle (0: start of clip)
  int fresh;            // 1: start with no tracks, 0: continue from `resume`
  AssocResume* resume;  // [B]
  cpx_track_params params;
  const int* clip_first;
  const int* proc_off;
  const int* proc_idx;
  const int* proc_ffc;
  const Component* comps;  // [total_frames * cap]
  const FrameInfo* info;   // [total_frames]
  cpx_region* pool;        // [total_frames * max_active]
  cpx_track_record* tracks;  // [B * max_tracks]
  int* n_tracks;           // [B]
  int* status;             // [B]
  cpx_region* regions_out; // [total_frames * cap] or nullptr
  int* region_counts;      // [total_frames] or nullptr
  // scratch
  ActiveTrack* active;     // [B * max_active]
  cpx_region* regs;        // [B * cap]
  ScoreRec* scores;        // [B * cap * max_active]
  unsigned char* used;     // [B * cap]
};
void launch_assoc(const AssocArgs& a, hipStream_t s);
struct FinalArgs {
  int B, square_width, max_frames;
  cpx_filter_params params;
  const int* clip_first;
  const int* proc_off;
  const int* proc_idx;
  const int* proc_ffc;
  const cpx_region* pool;
  const cpx_track_record* tracks;
  const int* n_tracks;
  cpx_track_summary* summaries;
  int* counts;          // [B][4]
  double* scratch_d;    // [B][2*max_frames]
  float* scratch_f;     // [B][max_frames]
  // plan pass
  const int* prefix;    // [B][4]
  cpx_region_ref* refs;
  int* track_offsets;
  cpx_crop_req* reqs;
  int* sample_track;
  int* track_clip;
};
void launch_finalize(const FinalArgs& a, hipStream_t s);
void launch_plan(const FinalArgs& a, hipStream_t s);
void launch_counts_prefix(const int* counts, int B, int* prefix, hipStream_t s);
size_t assoc_active_bytes();
size_t assoc_score_bytes();

struct ClassifyArgs {
  int W, H, crop_x, crop_y, crop_w, crop_h;
  int frame_size, square_width;
  int limits_flags;  // CPX_LIMITS_* (cpx_limits_kernel)
  const uint16_t* frames;
  const float* filtered;
  const FrameInfo* info;
  const cpx_region_ref* refs;
  const int* track_offsets;
  cpx_track_limits* limits;
  const cpx_crop_req* reqs;
  float* out;
};
struct AggregateArgs {
  int n_samples, n_tracks, n_labels, fp_index, square_width;
  const float* probs;
  const int* sample_track;
  const cpx_crop_req* reqs;
  float* scores;
  int* best;
};
void launch_aggregate(const AggregateArgs& a, hipStream_t s);
void launch_limits(const ClassifyArgs& a, int n_tracks, hipStream_t s);
void launch_crop(const ClassifyArgs& a, int n_reqs, hipStream_t s);

struct ConvArgs {
  int N, H, W, Cin, Cout, groups, ksize, stride, relu;
  int Ho, Wo, pad_top, pad_left;
  const float* in;
  float* out;
  const float* weights;
  const float* in_scale;
  const float* in_shift;
  const float* out_scale;
  const float* out_shift;
  const float* residual;
  // optional fused 1x1 shortcut (split-operand kernels only): instead of reading a residual tensor the kernel adds
  // conv1x1(sc_in, stride sc_stride, valid) + sc_bias, accumulated into the same float32 accumulators
  const float* sc_in;    // [N, sc_H, sc_W, sc_cin] or nullptr
  const float* sc_w;     // packed [groups][1][sc_cin / groups][Cout / groups]
  const float* sc_bias;  // [Cout]
  int sc_H, sc_W, sc_cin, sc_stride;
  // conv_block32_kernel<true, true> only: the network's first convolution computed while the block's patch is staged (`in` is
  // then never read): its raw input [N, H, W, groups], packed weights [groups][9][1][8] and bias [8 groups]
  const float* c1_in;
  const float* c1_w;
  const float* c1_b;
  // bf16 planes per operand on the split-operand path: 0 / 3 = the exact three-way split, 2 = CPX_CNN_MATH_BF16X2
  // (the layers conv_bf3_two_planes() names; every other layer keeps three)
  int planes;
  // CPX_CNN_MATH_FP16X2 (planes == 2 && half): the two planes are fp16 (11 + 11 significand bits, the products on
  // v_mfma_f32_*_f16).  fp16 has a range: the activated input is multiplied by act_scale (a power of two: exact) before
  // the split, the weights of output channel c by w_scale[c] (a power of two chosen per channel when the image is
  // built); the accumulators then hold act_scale * w_scale[c] times the sum and the epilogue multiplies by
  // act_unscale * w_unscale[c] (exact).  A scaled activation above fp16's largest finite value sets *ovf (atomicOr);
  // a workgroup that finds *ovf set on entry returns at once -- the layer is then run again by the three-plane bf16
  // kernel, which is launched right behind with guard = ovf and returns at once when *guard == 0.
  int half;
  float act_scale, act_unscale;
  const float* w_scale;    // [Cout] 2^kw, behind the plane image
  const float* w_unscale;  // [Cout] 2^-kw
  int* ovf;
  const int* guard;
  // producer-side split (fp16x2, a block's first convolution feeding its second): out_planes = the output is stored as
  // the NEXT layer's scaled fp16 planes (per pixel and channel quad 16 bytes: four hi halves, four lo halves) instead of
  // float32, multiplied by out_act_scale first; in_planes = the input is in that form (no prologue, no split: the
  // staging is a copy)
  int out_planes, in_planes;
  float out_act_scale;
};
struct HeadArgs {
  int N, HW, C, L;
  int n_hidden, activation;  // hidden Dense(relu) layers; CPX_HEAD_SIGMOID / CPX_HEAD_SOFTMAX
  int hidden_sizes[CPX_HEAD_MAX_HIDDEN];
  const float* in;
  const float* bn_scale;
  const float* bn_shift;
  const float* hidden_w[CPX_HEAD_MAX_HIDDEN];
  const float* hidden_b[CPX_HEAD_MAX_HIDDEN];
  const float* dense_w;
  const float* dense_b;
  float* logits;
  float* probs;
};
int launch_conv(const ConvArgs& a, hipStream_t s);
// float32 operands as three bf16 planes on the bf16 matrix pipe (cpx_cnn_bf3.hip)
bool conv_bf3_supported(const ConvArgs& a);
size_t conv_bf3_weight_bytes(const ConvArgs& a);
bool conv_bf3_two_planes(const ConvArgs& a);
bool conv_bf3_can_store_planes(const ConvArgs& a);
bool conv_bf3_can_load_planes(const ConvArgs& a);
void launch_split_weights(const ConvArgs& a, void* wimg, hipStream_t s);
int launch_conv_bf3(const ConvArgs& a, const void* wimg, hipStream_t s);
// a stage-2 residual block (two stride-1 3x3 convolutions, 32 channels per group) in one fp16x2 launch: `a` / `b` the two
// convolutions as launch_conv_bf3 would take them (half, act_scale, ovf set), wimg_a / wimg_b their weight images
bool conv_block32_supported(const ConvArgs& a, const ConvArgs& b);
int launch_conv_block32(const ConvArgs& a, const ConvArgs& b, const void* wimg_a, const void* wimg_b, hipStream_t s);
// the same block with the two convolutions on different waves of the workgroup (cpx_cnn_blk.hip); a / b prepared by
// launch_conv_block32 (w_scale / w_unscale set), wa / wb the fp16 plane images
int launch_conv_block32s(const ConvArgs& a, const ConvArgs& b, const void* wa, const void* wb, hipStream_t s);
// fp16x2 3x3 layers whose weights fit a workgroup's registers (cpx_cnn_rw.hip): kind 1 = stride 1, 64 -> 64 channels per
// group; 2 = stride 2, 32 -> 64; 3 = stride 3, 64 -> 128; 0 = not taken.  `wimg` = the layer's fp16 plane image in 32-channel chunks
int conv_rw_kind(const ConvArgs& a);
bool conv_rw_layer(const ConvArgs& a);  // kind 1
int launch_conv_rw(const ConvArgs& a, const void* wimg, hipStream_t s);
void launch_head(const HeadArgs& a, hipStream_t s);
void launch_count_overflow(int* ovf, int n_words, hipStream_t s);

struct CptvArgs {
  int W, H;
  const unsigned char* payload;
  const long long* frame_offsets;
  const int* bit_widths;
  const int* clip_offsets;
  uint16_t* frames_out;
};
int launch_cptv_unpack(const CptvArgs& a, int B, hipStream_t s);

struct CptvInflateArgs {
  int B;
  const unsigned char* in;
  const cpx_cptv_file* files;
  unsigned char* out;
  cpx_cptv_frame_slot* slots;
  unsigned char* header;
  cpx_cptv_file_result* results;
};
int launch_cptv_inflate(const CptvInflateArgs& a, hipStream_t s);

struct CptvGatherArgs {
  const cpx_cptv_frame_slot* slots;
  const long long* slot_offsets;
  const int* clip_offsets;
  long long* frame_offsets;
  int* bit_widths;
  cpx_cptv_frame_slot* slots_out;
};
int launch_cptv_gather(const CptvGatherArgs& a, int B, hipStream_t s);

struct ThumbArgs {
  int W, H, chain_cap;
  int max_w, max_h;   // the launch's LDS holds regions up to this size; a larger one reports CPX_ERR_OVERFLOW
  const uint16_t* frames;
  const int32_t* labels;
  const cpx_frame_info* info;
  const cpx_region_ref* refs;
  cpx_thumb_stat* out;
};
int launch_thumb(const ThumbArgs& a, int n_refs, hipStream_t s);

struct TracklessArgs {
  int W, H, frame, background;
  const uint16_t* frames;
  int32_t* out;           // [n][2]
  const int32_t* pairs;   // [n][2] = frame, background (null: the one pair above)
  int n;
};
int launch_trackless(const TracklessArgs& a, hipStream_t s);

struct IrArgs {
  int W, H, threshold, max_components;
  const unsigned char* images;  // [n, H, W]
  cpx_component* comps;         // [n][max_components]
  int32_t* counts;              // [n]
  int32_t* status;              // [n]
  int32_t* labels;              // [n, H, W] or nullptr
  unsigned char* slots;         // n_slots scratch slots of slot_bytes for frames whose tables do not fit LDS
  size_t slot_bytes;
  uint32_t* slot_bitmap;        // [8] busy bits
  int n_slots;
};
int launch_ir_detect(const IrArgs& a, int n_frames, hipStream_t s);
struct IrVarArgs {
  int W, H, n;
  const unsigned char* cur;   // [H, W]
  const unsigned char* prev;  // [H, W]
  const int* rects;           // [n][4] x, y, width, height
  double* out;                // [n]
};
void launch_ir_delta_variance(const IrVarArgs& a, hipStream_t s);
struct IrMergeArgs {
  int W, H, n, cap_in, cap_out;
  int frame_number, out_stride;   // stream v's outputs go to row v * out_stride + frame_number
  const cpx_component* comps;     // [n][cap_in] of cpx_ir_detect
  const int* counts;              // [n]
  const unsigned char* cur;       // [n][H][W]
  const unsigned char* prev;      // [n][H][W] or null
  cpx_component* out_comps;       // rows of cap_out
  cpx_frame_info* out_info;       // rows (optional)
  int* status;                    // [n]
};
void launch_ir_merge(const IrMergeArgs& a, hipStream_t s);
void launch_ir_resize_area(const unsigned char* src, unsigned char* dst, int n, int W, int H, int f, hipStream_t s);
struct IrStatsArgs {
  int n, pixels, vec16;
  const unsigned char* frames;   // [n][pixels]
  const unsigned char* masks;    // [n][pixels] or null
  unsigned int* hist;            // [n][256], zero on entry
  cpx_ir_frame_stats* out;       // [n], filtered_sum zero on entry
};
void launch_ir_frame_stats(const IrStatsArgs& a, hipStream_t s);
int ir_supported(int W, int H);
size_t ir_slot_bytes(int W, int H);

struct Mog2Args {
  size_t n;  // streams * width * height
  float alphaT, alpha1, prune;
  float var_threshold, background_ratio, var_threshold_gen, var_init, var_min, var_max;
  const unsigned char* frames;  // [n]
  float* weight;                // [5][n] mode-major planes
  float* var;
  float* mean;
  unsigned char* modes;         // [n]
  unsigned char* mask;          // [n]
};
void launch_mog2_apply(const Mog2Args& a, hipStream_t s);
void launch_mog2_background(const Mog2Args& a, unsigned char* out, hipStream_t s);

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (kernel, device): `done` is a per-kernel array indexed by the current device ordinal.
inline bool cpx_dyn_lds_ready(const void* fn, bool* done, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (!done[dev]) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
    done[dev] = true;
  }
  return true;
}

size_t track_lds_bytes(int W, int H);
int track_max_pixels();
int track_lds_components();
int frame_kernel_attr_setup();
void launch_init(const TrackArgs& a, int B, int keep, hipStream_t s);
void launch_unpack_state(uint32_t* wsum, uint16_t* kcnt, size_t n, hipStream_t s);
void launch_frame(const TrackArgs& a, int B, int t0, int t1, int mode, hipStream_t s);  // processed frames [t0, t1) of every clip
void launch_nlm(const TrackArgs& a, int B, int t, hipStream_t s);
void launch_median(const TrackArgs& a, int B, int t0, int t1, hipStream_t s);  // thermal medians of processed frames [t0, t1) -> FrameInfo
size_t nlm_lds_bytes(int W, int H);
int nlm_supported(int W, int H);
void launch_export_background(const TrackArgs& a, int B, float* out, hipStream_t s);

}  // namespace cpx
