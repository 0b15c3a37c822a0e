// cpx_api.cpp -- the C-ABI of libcpx_hip.so (include/cpx.h): handle lifetime,
// device workspace, host-side schedule of the per-frame launches.  No torch
// types, no exceptions across the boundary.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <map>
#include <string>
#include <vector>

#include "cpx.h"
#include "cpx_kernels.h"

struct cpx_cnn;
struct cpx_mog2;
static void cnn_free(cpx_cnn* c);
static void mog2_free(cpx_mog2* m);

struct cpx_handle {
  int device = 0;
  cpx_config cfg{};
  hipStream_t stream = nullptr;
  std::string err;
  // device workspace (grown lazily, reused)
  void* ws = nullptr;
  size_t ws_bytes = 0;
  double* wtab_dev = nullptr;
  uint32_t* wthr_dev = nullptr;
  int wtab_len = 0;
  std::vector<double> wtab_host;  // w_k, k = 0 .. wtab_len - 1 (the table the device holds)
  int* nlm_lut_dev = nullptr;
  // small device arrays for the schedule
  int* sched_dev = nullptr;
  size_t sched_ints = 0;
  struct ConvEv { int key; double flops; hipEvent_t e0, e1; };
  std::vector<ConvEv> conv_events;
  bool conv_timing = false;
  void* ws_assoc = nullptr;
  size_t ws_assoc_bytes = 0;
  // timing of the last batch
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int last_launches = 0;
  bool timing_valid = false;
  // incremental (one clip, frame by frame) tracking: frames consumed so far, -1 = no stream open
  // pipelined split of the frame step: back halves run on stream2 one step behind the front halves
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_back[2] = {nullptr, nullptr};
  int split_min_clips = 0;  // 0 = never split
  // the last track call kept the per-pixel kept-frame counts in the window sums' top ten bits (cpx_frame_kernel<true>): whatever
  // continues from that state, or exports it, unpacks it first (unpack_state)
  bool state_packed = false;
  bool packed_state_ok = true;   // CPX_TRACK_PACKED_STATE=0: never pack
  bool fuse_conv1 = true;   // conv1_1 inside the fused first block of stage 2 (CPX_CNN_FUSE_CONV1=0: a launch of its own)
  // CPX_TRACK_DEFER_MEDIANS: the median kernel of the last track call runs on stream2; ev_median marks its end
  bool medians_pending = false;
  hipEvent_t ev_median = nullptr;
  bool track_per_step = false;  // CPX_TRACK_PER_STEP=1: one launch per frame step (the form before the per-clip walk)
  std::vector<struct cpx_cnn*> cnns;  // networks created on this handle (destroyed with it)
  std::vector<struct cpx_mog2*> mog2s;  // background models created on this handle
  int stream_frames = -1;
  int stream_assoc_frames = -1;
  bool stream_filt_state = false;
  int last_B = 0;  // clips of the last track call: whose state cpx_get_background / CPX_TRACK_KEEP_BACKGROUND refer to
  struct StagedBackground { std::vector<uint16_t> bg, kcnt; double average; };
  std::map<int, StagedBackground> staged_bg;  // cpx_set_background: applied by the next track call
  int cnn_math = CPX_CNN_MATH_FP16X2;    // cpx_set_cnn_math / CPX_CNN_MATH (the default: include/cpx.h)
  bool fuse_shortcut = true;             // CPX_CNN_FUSE_SHORTCUT=0 keeps the 1x1 shortcuts as launches of their own
  void* bf3_scratch = nullptr;           // split weights of a cpx_conv2d call that brought none
  size_t bf3_scratch_bytes = 0;
  // activation buffers of cpx_cnn_forward (act0 | act1 | mid | sc), grown to the largest call seen and shared by every
  // network of the handle: forwards on one handle are serialised on its stream, and a second network (another model, another
  // leg of a run) must not bring 54 GB of its own (2,048 samples at frame size 32)
  float* cnn_arena = nullptr;
  size_t cnn_arena_floats = 0;
  int* cnn_ovf = nullptr;                // CPX_CNN_MATH_FP16X2: the overflow word of the forward (or bare convolution) in flight
  bool planes_handover = true;           // CPX_CNN_PLANES_HANDOVER=0: fp16x2 keeps `mid` float32 (every layer splits its own input)
  int block_fusion = 2;                  // CPX_CNN_BLOCK_FUSION: fp16x2 runs as ONE launch (conv_block32_kernel) 2 = every stage-2 block, 1 = all but the stage's first, 0 = none
  unsigned char* ir_scratch = nullptr;  // cpx_ir_detect: slots for frames whose run / component tables outgrow LDS
  size_t ir_scratch_bytes = 0;
  uint32_t* ir_bitmap = nullptr;
};

static_assert(sizeof(cpx_component) == 32, "cpx_component layout is part of the ABI");
static_assert(sizeof(cpx_frame_info) == 80, "cpx_frame_info layout is part of the ABI");
static_assert(sizeof(cpx_frame_meta) == 24, "cpx_frame_meta layout is part of the ABI");
static_assert(sizeof(cpx_config) == 48, "cpx_config layout is part of the ABI");
static_assert(sizeof(cpx_region_ref) == 24 && sizeof(cpx_track_limits) == 32 && sizeof(cpx_crop_req) == 32,
              "classification request layouts are part of the ABI");
static_assert(sizeof(cpx_filter_params) == 72 && sizeof(cpx_track_summary) == 120,
              "end-of-clip layouts are part of the ABI");
static_assert(sizeof(cpx_region) == 56, "cpx_region layout is part of the ABI");
static_assert(sizeof(cpx_track_record) == 32, "cpx_track_record layout is part of the ABI");
static_assert(sizeof(cpx_track_params) == 120, "cpx_track_params layout is part of the ABI");

namespace {

int fail(cpx_handle* h, int code, const char* what, hipError_t e = hipSuccess) {
  if (h) {
    h->err = what;
    if (e != hipSuccess) {
      h->err += ": ";
      h->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define CPX_HIP(h, call)                                            \
  do {                                                              \
    hipError_t _e = (call);                                         \
    if (_e != hipSuccess) return fail((h), CPX_ERR_HIP, #call, _e); \
  } while (0)

// every entry point: select the handle's device and drop stale errors other HIP users of the process left behind,
// so that the hipGetLastError() after our launches reports our launches only
#define CPX_ENTER(h)                           \
  do {                                         \
    CPX_HIP((h), hipSetDevice((h)->device));   \
    (void)hipGetLastError();                   \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct WsLayout {
  size_t bg, wsum, kcnt, filt, cstate, u8, carry, bgavg, big, total;
};

WsLayout ws_layout(const cpx_config& c, int B, bool need_filt_state) {
  const size_t P = (size_t)c.width * c.height;
  WsLayout l{};
  size_t off = 0;
  l.bg = off;
  off = align_up(off + (size_t)B * 2 * P * sizeof(uint16_t), 256);
  l.wsum = off;
  off = align_up(off + (size_t)B * P * sizeof(uint32_t), 256);
  l.kcnt = off;
  off = align_up(off + (size_t)B * P * sizeof(uint16_t), 256);
  l.filt = off;
  if (need_filt_state) off = align_up(off + (size_t)B * 2 * P * sizeof(float), 256);
  l.cstate = off;
  off = align_up(off + (size_t)B * sizeof(cpx::ClipState), 256);
  // hand-over buffers of split frame steps (front -> NLM -> back, or front || back pipelined): two slots per clip
  l.u8 = off;
  off = align_up(off + (size_t)B * 2 * P, 256);
  l.carry = off;
  off = align_up(off + (size_t)B * 2 * sizeof(cpx::FrameCarry), 256);
  l.bgavg = off;
  off = align_up(off + (size_t)B * sizeof(double), 256);
  // a handle whose frames may hold more components than the frame kernel's LDS tables: the same tables per clip in HBM
  // (eight statistics rows + the rank row of max_components entries; cpx_track.hip phase 7)
  l.big = off;
  if (c.max_components > cpx::track_lds_components())
    off = align_up(off + (size_t)B * 9 * c.max_components * sizeof(uint32_t), 256);
  l.total = off;
  return l;
}

struct Schedule {
  std::vector<int> clip_first, proc_off, proc_idx, proc_ffc, order;
  int total = 0, max_proc = 0;
};

// which frames are processed (background frames only initialise, cliptrackextractor.py:167-168)
// and their FFC flags (cptvmotiondetector.py:211-223 with int milliseconds)
int build_schedule(cpx_handle* h, const int32_t* clip_offsets, const cpx_frame_meta* meta, int B, Schedule* sc) {
  sc->total = clip_offsets[B];
  sc->clip_first.resize(B);
  sc->proc_off.assign(B + 1, 0);
  sc->proc_idx.reserve(sc->total);
  sc->proc_ffc.reserve(sc->total);
  for (int b = 0; b < B; ++b) {
    const int f0 = clip_offsets[b], f1 = clip_offsets[b + 1];
    if (f1 <= f0) return fail(h, CPX_ERR_INVALID, "empty clip in batch");
    sc->clip_first[b] = f0;
    for (int f = f0; f < f1; ++f) {
      if (meta[f].background_frame) continue;
      sc->proc_idx.push_back(f);
      int ffc = 0;
      if (meta[f].has_times) ffc = (meta[f].time_on_ms - meta[f].last_ffc_ms) < 9 ? 1 : 0;
      sc->proc_ffc.push_back(ffc);
    }
    sc->proc_off[b + 1] = (int)sc->proc_idx.size();
    const int np = sc->proc_off[b + 1] - sc->proc_off[b];
    if (np > h->cfg.max_frames) return fail(h, CPX_ERR_INVALID, "clip longer than max_frames");
    sc->max_proc = std::max(sc->max_proc, np);
  }
  // longest clips first: one workgroup walks a whole clip, and the dispatcher hands out workgroups in index order
  sc->order.resize(B);
  for (int b = 0; b < B; ++b) sc->order[b] = b;
  std::stable_sort(sc->order.begin(), sc->order.end(), [&](int x, int y) {
    return sc->proc_off[x + 1] - sc->proc_off[x] > sc->proc_off[y + 1] - sc->proc_off[y];
  });
  return CPX_OK;
}

// layout in sched_dev: clip_first[B] | proc_off[B+1] | proc_idx[n] | proc_ffc[n] | order[B]
int upload_schedule(cpx_handle* h, const Schedule& sc, int B) {
  const int n = std::max((int)sc.proc_idx.size(), 1);
  const size_t ints = (size_t)B + (B + 1) + 2 * (size_t)n + (size_t)B;
  if (ints > h->sched_ints) {
    if (h->sched_dev) hipFree(h->sched_dev);
    h->sched_dev = nullptr;
    h->sched_ints = 0;
    hipError_t e = hipMalloc((void**)&h->sched_dev, ints * sizeof(int));
    if (e != hipSuccess) return fail(h, CPX_ERR_NOMEM, "schedule hipMalloc", e);
    h->sched_ints = ints;
  }
  std::vector<int> flat(ints, 0);
  std::copy(sc.clip_first.begin(), sc.clip_first.end(), flat.begin());
  std::copy(sc.proc_off.begin(), sc.proc_off.end(), flat.begin() + B);
  std::copy(sc.proc_idx.begin(), sc.proc_idx.end(), flat.begin() + B + (B + 1));
  std::copy(sc.proc_ffc.begin(), sc.proc_ffc.end(), flat.begin() + B + (B + 1) + n);
  std::copy(sc.order.begin(), sc.order.end(), flat.begin() + B + (B + 1) + 2 * (size_t)n);
  CPX_HIP(h, hipMemcpyAsync(h->sched_dev, flat.data(), ints * sizeof(int), hipMemcpyHostToDevice, h->stream));
  CPX_HIP(h, hipStreamSynchronize(h->stream));  // `flat` dies with this scope
  return CPX_OK;
}

}  // namespace

extern "C" {

int cpx_abi_version(void) { return CPX_ABI_VERSION; }

int cpx_create(int device_id, const cpx_config* cfg, cpx_handle** out) {
  if (!cfg || !out) return CPX_ERR_INVALID;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CPX_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= ndev) return CPX_ERR_INVALID;
  const int W = cfg->width, H = cfg->height;
  if (W <= 0 || H <= 0 || (W % 8) != 0 || W >= 192 || W * H > cpx::track_max_pixels() || H < 5 || W < 8)
    return CPX_ERR_UNSUPPORTED;
  if (cfg->edge_pixels < 0 || 2 * cfg->edge_pixels >= std::min(W, H) || cfg->window < 1 ||
      cfg->max_components < 1 || cfg->max_frames < 1 || cfg->max_frames > 65534)
    return CPX_ERR_INVALID;
  cpx_handle* h = new (std::nothrow) cpx_handle();
  if (!h) return CPX_ERR_NOMEM;
  h->device = device_id;
  h->cfg = *cfg;
  if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_front[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_front[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_back[0], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_back[1], hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_median, hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
    cpx_destroy(h);
    return CPX_ERR_HIP;
  }
  // weight table: w_k = k-fold float64 accumulation of weight_add, exactly as
  // NumPy evaluates background_weight + weight_add (motiondetector.py:218-222)
  // one entry per value the uint16 per-pixel counter can take: CPX_TRACK_KEEP_BACKGROUND chains and long streams
  // carry a pixel's count past this handle's max_frames (the tables cost 786 KB per handle)
  h->wtab_len = 65536;
  h->wtab_host.resize(h->wtab_len);
  std::vector<double>& wt = h->wtab_host;
  double w = 0.0;
  for (int k = 0; k < h->wtab_len; ++k) {
    wt[k] = w;
    w = w + cfg->weight_add;
  }
  if (cfg->denoise) {
    if (!cpx::nlm_supported(W, H)) {
      cpx_destroy(h);
      return CPX_ERR_UNSUPPORTED;
    }
    // fastNlMeansDenoising weight table (h = 3, template 7x7, search 21x21), SURVEY.md Appendix A.6:
    // w[a] = round(fixed_point_mult * exp(-a * (64/49) / h^2)), zero below 0.001 * fixed_point_mult
    const int fixed_point_mult = (int)(2147483647LL / (21 * 21 * 255));
    std::vector<int> lut(64, 0);
    for (int a2 = 0; a2 < 64; ++a2) {
      const double wv = std::exp(-((double)a2 * (64.0 / 49.0)) / 9.0);
      const double wr = std::nearbyint((double)fixed_point_mult * wv);
      lut[a2] = (wr < 0.001 * fixed_point_mult) ? 0 : (int)wr;
    }
    if (lut[63] != 0 || hipMalloc((void**)&h->nlm_lut_dev, 64 * sizeof(int)) != hipSuccess ||
        hipMemcpy(h->nlm_lut_dev, lut.data(), 64 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
      cpx_destroy(h);
      return CPX_ERR_HIP;
    }
  }
  // integer form of `bg < f - w_k` for integer bg, f (cpx_track.hip, streaming pass): keep <=> f - bg >= hi_k,
  // hi_k = floor(w_k) + 1.  Exact whenever w_k is an integer or at least 1e-6 away from one (f - w_k is then no
  // integer and its float64 rounding, < 2e-11 for f < 65536, cannot reach one).  Otherwise ("near") the kernel decides
  // f - bg == rint(w_k) with the float64 expression; hi_k = rint(w_k) + 1 then.  The entry is 2 hi_k - near_k: the kernel
  // keeps on 2 (f - bg) + 1 > entry and evaluates the float64 expression on equality (odd entries only).
  std::vector<uint32_t> thr(h->wtab_len);
  for (int k = 0; k < h->wtab_len; ++k) {
    const double wk = wt[k], m = std::nearbyint(wk);
    const bool exact = (wk == m), near = !exact && std::fabs(wk - m) < 1e-6;
    double hi = near ? m + 1.0 : std::floor(wk) + 1.0;
    if (!(hi >= 0.0)) hi = 0.0;                    // (negative weight_add: never reached by a sane config)
    if (hi > 536870912.0) hi = 536870912.0;        // beyond any f - bg: never kept (and 2 hi a positive int32)
    thr[k] = 2u * (uint32_t)hi - (near ? 1u : 0u);
  }
  if (hipMalloc((void**)&h->wthr_dev, thr.size() * sizeof(uint32_t)) != hipSuccess ||
      hipMemcpy(h->wthr_dev, thr.data(), thr.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
    cpx_destroy(h);
    return CPX_ERR_HIP;
  }
  if (hipMalloc(&h->wtab_dev, wt.size() * sizeof(double)) != hipSuccess ||
      hipMemcpy(h->wtab_dev, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      cpx::frame_kernel_attr_setup() != 0) {
    cpx_destroy(h);
    return CPX_ERR_HIP;
  }
  if (const char* env = std::getenv("CPX_TRACK_SPLIT_MIN_CLIPS")) h->split_min_clips = std::atoi(env);
  if (const char* env = std::getenv("CPX_TRACK_PACKED_STATE")) h->packed_state_ok = std::atoi(env) != 0;
  if (const char* env = std::getenv("CPX_CNN_FUSE_CONV1")) h->fuse_conv1 = std::atoi(env) != 0;
  if (const char* env = std::getenv("CPX_TRACK_PER_STEP")) h->track_per_step = std::atoi(env) != 0;
  if (const char* env = std::getenv("CPX_CNN_FUSE_SHORTCUT")) h->fuse_shortcut = std::atoi(env) != 0;
  if (const char* env = std::getenv("CPX_CNN_PLANES_HANDOVER")) h->planes_handover = std::atoi(env) != 0;
  if (const char* env = std::getenv("CPX_CNN_BLOCK_FUSION")) h->block_fusion = std::min(std::max(std::atoi(env), 0), 2);
  if (const char* env = std::getenv("CPX_CNN_MATH")) {
    if (!std::strcmp(env, "f32")) h->cnn_math = CPX_CNN_MATH_F32;
    else if (!std::strcmp(env, "bf16x3")) h->cnn_math = CPX_CNN_MATH_BF16X3;
    else if (!std::strcmp(env, "bf16x2")) h->cnn_math = CPX_CNN_MATH_BF16X2;
    else if (!std::strcmp(env, "fp16x2")) h->cnn_math = CPX_CNN_MATH_FP16X2;
  }
  *out = h;
  return CPX_OK;
}

void cpx_destroy(cpx_handle* h) {
  if (!h) return;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (cpx_cnn* c : h->cnns) cnn_free(c);
  h->cnns.clear();
  for (cpx_mog2* m : h->mog2s) mog2_free(m);
  h->mog2s.clear();
  if (h->ws) hipFree(h->ws);
  if (h->wtab_dev) hipFree(h->wtab_dev);
  if (h->wthr_dev) hipFree(h->wthr_dev);
  if (h->nlm_lut_dev) hipFree(h->nlm_lut_dev);
  if (h->sched_dev) hipFree(h->sched_dev);
  if (h->ws_assoc) hipFree(h->ws_assoc);
  if (h->ir_scratch) hipFree(h->ir_scratch);
  if (h->bf3_scratch) hipFree(h->bf3_scratch);
  if (h->cnn_ovf) hipFree(h->cnn_ovf);
  if (h->cnn_arena) hipFree(h->cnn_arena);
  if (h->ir_bitmap) hipFree(h->ir_bitmap);
  for (auto& e : h->conv_events) {
    hipEventDestroy(e.e0);
    hipEventDestroy(e.e1);
  }
  for (int i = 0; i < 2; ++i) {
    if (h->ev_front[i]) hipEventDestroy(h->ev_front[i]);
    if (h->ev_back[i]) hipEventDestroy(h->ev_back[i]);
  }
  if (h->stream2) {
    hipStreamSynchronize(h->stream2);
    hipStreamDestroy(h->stream2);
  }
  if (h->ev_median) hipEventDestroy(h->ev_median);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
}

const char* cpx_last_error(const cpx_handle* h) { return h ? h->err.c_str() : "null handle"; }

void* cpx_stream(cpx_handle* h) { return h ? (void*)h->stream : nullptr; }

// Medians of a CPX_TRACK_DEFER_MEDIANS call still in flight on stream2: what is enqueued on the handle's stream from here
// on runs behind them.  Called by every entry point that reads cpx_frame_info.thermal_median or reuses the buffers the
// median kernel reads and writes.
static int join_medians(cpx_handle* h) {
  if (h->medians_pending) {
    h->medians_pending = false;
    CPX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_median, 0));
  }
  return CPX_OK;
}

int cpx_join_medians(cpx_handle* h) {
  if (!h) return CPX_ERR_INVALID;
  return join_medians(h);
}

int cpx_release_memory(cpx_handle* h) {
  if (!h) return CPX_ERR_INVALID;
  CPX_ENTER(h);
  if (int rc = join_medians(h)) return rc;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  CPX_HIP(h, hipStreamSynchronize(h->stream2));
  if (h->cnn_arena) hipFree(h->cnn_arena);
  h->cnn_arena = nullptr;
  h->cnn_arena_floats = 0;
  if (h->bf3_scratch) hipFree(h->bf3_scratch);
  h->bf3_scratch = nullptr;
  h->bf3_scratch_bytes = 0;
  if (h->ws) hipFree(h->ws);
  h->ws = nullptr;
  h->ws_bytes = 0;
  h->last_B = 0;
  h->state_packed = false;
  if (h->ws_assoc) hipFree(h->ws_assoc);
  h->ws_assoc = nullptr;
  h->ws_assoc_bytes = 0;
  if (h->ir_scratch) hipFree(h->ir_scratch);
  h->ir_scratch = nullptr;
  h->ir_scratch_bytes = 0;
  return CPX_OK;
}

int cpx_synchronize(cpx_handle* h) {
  if (!h) return CPX_ERR_INVALID;
  if (int rc = join_medians(h)) return rc;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  return CPX_OK;
}

size_t cpx_track_workspace_bytes(const cpx_handle* h, int B, int total_frames) {
  if (!h || B <= 0) return 0;
  (void)total_frames;
  return ws_layout(h->cfg, B, true).total;
}

// Track stage for B clips.  n_prev < 0: whole clips (cpx_track_batch).  n_prev >= 0 (B == 1): the clip's first n_prev
// frames were consumed by earlier calls on this handle and its state is still in the workspace; only the frames
// [n_prev, clip_offsets[1]) are processed (cpx_track_frame).
// the state of the last track call in the layout every path but cpx_frame_kernel<true> reads (see cpx_handle::state_packed)
static int unpack_state(cpx_handle* h) {
  if (h->state_packed && h->ws && h->last_B > 0) {
    const WsLayout lp = ws_layout(h->cfg, h->last_B, h->stream_filt_state);
    char* base = (char*)h->ws;
    cpx::launch_unpack_state((uint32_t*)(base + lp.wsum), (uint16_t*)(base + lp.kcnt),
                             (size_t)h->last_B * h->cfg.width * h->cfg.height, h->stream);
    CPX_HIP(h, hipGetLastError());
  }
  h->state_packed = false;
  return CPX_OK;
}

static int track_run(cpx_handle* h, const uint16_t* frames_dev, const int32_t* clip_offsets,
                     const cpx_frame_meta* meta, int B, int n_prev, cpx_component* comps_dev,
                     cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                     float* background_dev, int flags) {
  CPX_ENTER(h);
  if (flags & ~(CPX_TRACK_KEEP_BACKGROUND | CPX_TRACK_FREEZE_ON_FFC | CPX_TRACK_FREEZE_BACKGROUND | CPX_TRACK_DEFER_MEDIANS))
    return fail(h, CPX_ERR_INVALID, "track: unknown flag");
  if (int jrc = join_medians(h)) return jrc;  // (a previous call's medians read the frames / write the records this one may reuse)
  const bool defer_medians = (flags & CPX_TRACK_DEFER_MEDIANS) != 0;
  const cpx_config& c = h->cfg;
  Schedule sc;
  int rc = build_schedule(h, clip_offsets, meta, B, &sc);
  if (rc != CPX_OK) return rc;
  const int total = sc.total, max_proc = sc.max_proc;
  const bool resume = n_prev > 0;
  int t_begin = 0;
  if (resume)
    for (int f = 0; f < n_prev; ++f) t_begin += meta[f].background_frame ? 0 : 1;
  // ---- device workspace ----
  const bool need_filt = (filtered_dev == nullptr);
  const WsLayout l = ws_layout(c, B, need_filt);
  if (resume && (l.total > h->ws_bytes || need_filt != h->stream_filt_state))
    return fail(h, CPX_ERR_INVALID, "cpx_track_frame: the stream's workspace is gone (optional outputs changed?)");
  const bool keep = !resume && (flags & CPX_TRACK_KEEP_BACKGROUND);
  if (keep) {
    // every clip needs a state to continue from: the previous call's (same layout) or a staged one
    const bool have_prev = h->ws && h->last_B == B && need_filt == h->stream_filt_state && l.total <= h->ws_bytes;
    for (int b = 0; b < B && !have_prev; ++b)
      if (!h->staged_bg.count(b)) {
        h->staged_bg.clear();  // staged for THIS call: a refused call does not leave them behind for the one after
        return fail(h, CPX_ERR_INVALID, "CPX_TRACK_KEEP_BACKGROUND: no background state for a clip (previous call had another batch size, and nothing staged; staged states dropped)");
      }
  }
  for (const auto& kv : h->staged_bg)
    if (kv.first >= B) {
      h->staged_bg.clear();  // staged for THIS call: a refused call does not leave them behind for the one after
      return fail(h, CPX_ERR_INVALID, "cpx_set_background: staged clip index outside the batch (staged states dropped)");
    }
  // a call that continues from the previous call's state reads it in the two-array layout; a fresh one starts its own
  if (resume || keep) {
    if (int urc = unpack_state(h)) return urc;
  }
  h->state_packed = false;
  h->stream_filt_state = need_filt;
  h->last_B = B;
  if (l.total > h->ws_bytes) {
    if (h->ws) hipFree(h->ws);
    h->ws = nullptr;
    h->ws_bytes = 0;
    hipError_t e = hipMalloc(&h->ws, l.total);
    if (e != hipSuccess) return fail(h, CPX_ERR_NOMEM, "workspace hipMalloc", e);
    h->ws_bytes = l.total;
  }
  rc = upload_schedule(h, sc, B);
  if (rc != CPX_OK) return rc;
  const int nproc_total = (int)sc.proc_idx.size();

  cpx::TrackArgs a{};
  a.W = c.width;
  a.H = c.height;
  a.edge = c.edge_pixels;
  a.window = c.window;
  a.cap_out = c.max_components;
  a.flags = flags;
  a.background_thresh = c.background_thresh;
  a.weight_add = c.weight_add;
  a.frames = frames_dev;
  a.clip_first = h->sched_dev;
  a.proc_off = h->sched_dev + B;
  a.proc_idx = h->sched_dev + B + (B + 1);
  a.proc_ffc = h->sched_dev + B + (B + 1) + std::max(nproc_total, 1);
  a.order = h->sched_dev + B + (B + 1) + 2 * (size_t)std::max(nproc_total, 1);
  a.wtab = h->wtab_dev;
  a.wtab_len = h->wtab_len;
  a.wthr = h->wthr_dev;
  char* base = (char*)h->ws;
  a.bg = (uint16_t*)(base + l.bg);
  a.wsum = (uint32_t*)(base + l.wsum);
  a.kcnt = (uint16_t*)(base + l.kcnt);
  // a fresh batch of at most 1023 processed frames per clip: the kept-frame counts ride in the window sums (cpx_track.hip, PK)
  a.packed_state = (h->packed_state_ok && !resume && !keep && h->staged_bg.empty() && max_proc <= 1023 && c.window <= 64) ? 1 : 0;
  a.filt_state = need_filt ? (float*)(base + l.filt) : nullptr;
  a.cstate = (cpx::ClipState*)(base + l.cstate);
  a.u8_state = (unsigned char*)(base + l.u8);
  a.carry = (cpx::FrameCarry*)(base + l.carry);
  a.bgavg = (double*)(base + l.bgavg);
  a.big_stat = c.max_components > cpx::track_lds_components() ? (uint32_t*)(base + l.big) : nullptr;
  a.nlm_flip = c.denoise ? 1 : 0;
  a.nlm_lut = h->nlm_lut_dev;
  a.comps_out = comps_dev;
  a.info_out = info_dev;
  a.labels_out = labels_dev;
  a.filtered_out = filtered_dev;

  // frames that are never processed (background frames) get frame_number = -1
  const int f_new = resume ? n_prev : 0;
  CPX_HIP(h, hipMemsetAsync(info_dev + f_new, 0xFF, (size_t)(total - f_new) * sizeof(cpx_frame_info), h->stream));
  if (!resume) cpx::launch_init(a, B, keep ? 1 : 0, h->stream);
  if (!h->staged_bg.empty()) {
    // staged states replace the seeding (or, in the middle of a stream, the model an external owner changed): both
    // ping-pong slots, the weight counters, the average
    const size_t P = (size_t)c.width * c.height;
    for (const auto& kv : h->staged_bg) {
      const int b = kv.first;
      const auto& sb = kv.second;
      for (int slot = 0; slot < 2; ++slot)
        CPX_HIP(h, hipMemcpyAsync(a.bg + ((size_t)b * 2 + slot) * P, sb.bg.data(), P * sizeof(uint16_t),
                                  hipMemcpyHostToDevice, h->stream));
      CPX_HIP(h, hipMemcpyAsync(a.kcnt + (size_t)b * P, sb.kcnt.data(), P * sizeof(uint16_t), hipMemcpyHostToDevice,
                                h->stream));
      CPX_HIP(h, hipMemcpyAsync(&a.cstate[b].bg_average, &sb.average, sizeof(double), hipMemcpyHostToDevice, h->stream));
      CPX_HIP(h, hipMemcpyAsync(a.bgavg + b, &sb.average, sizeof(double), hipMemcpyHostToDevice, h->stream));
    }
    CPX_HIP(h, hipStreamSynchronize(h->stream));  // the staged vectors die here
    h->staged_bg.clear();
  }
  // medians of all new frames first (independent of the clips' frame order; their word of the records is theirs alone,
  // the frame kernel stores around it): one workgroup per (clip, step).  Nothing orders the two kernels, but running
  // them side by side on two streams gains nothing (the call takes 123.1 ms against 122.6 ms,
  // profiles/r06_track_experiments.md): in front, on the one stream.
  if ((long long)B * (max_proc - t_begin) > 2147483647LL) return fail(h, CPX_ERR_INVALID, "batch too large for one launch");
  // CPX_TRACK_DEFER_MEDIANS: behind the frame kernel on the second stream instead (below), beside the caller's next stages
  if (!defer_medians) cpx::launch_median(a, B, t_begin, max_proc, h->stream);
  CPX_HIP(h, hipEventRecord(h->ev0, h->stream));  // (ev0 .. ev1 bracket the frame / NLM kernels)
  // (with the internal ping-pong of filtered frames the back half of step t would read what the front half of
  // step t+1 overwrites: split only when the caller keeps every filtered frame)
  const bool split = !c.denoise && !need_filt && h->split_min_clips > 0 && B >= h->split_min_clips &&
                     max_proc - t_begin > 2;
  int launches = 0;
  if (!split && !c.denoise && !h->track_per_step) {
    // one workgroup per clip walks its frames: a single launch, no phase lockstep between the clips (cpx_track.hip)
    cpx::launch_frame(a, B, t_begin, max_proc, 0, h->stream);
    launches = 1;
  } else {
    for (int t = t_begin; t < max_proc; ++t) {
      if (split) {
        // front(t) reuses the hand-over slot that back(t - 2) read
        if (t - t_begin >= 2) CPX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_back[t & 1], 0));
        cpx::launch_frame(a, B, t, t + 1, 1, h->stream);
        CPX_HIP(h, hipEventRecord(h->ev_front[t & 1], h->stream));
        CPX_HIP(h, hipStreamWaitEvent(h->stream2, h->ev_front[t & 1], 0));
        cpx::launch_frame(a, B, t, t + 1, 2, h->stream2);
        CPX_HIP(h, hipEventRecord(h->ev_back[t & 1], h->stream2));
      } else if (!c.denoise) {
        cpx::launch_frame(a, B, t, t + 1, 0, h->stream);
      } else {  // front (normalise) -> non-local means -> back (blur / threshold / label / statistics)
        cpx::launch_frame(a, B, t, t + 1, 1, h->stream);
        cpx::launch_nlm(a, B, t, h->stream);
        cpx::launch_frame(a, B, t, t + 1, 2, h->stream);
      }
      ++launches;
    }
  }
  if (split) {  // everything enqueued later on the handle's stream sees the last back halves
    CPX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_back[(max_proc - 1) & 1], 0));
    if (max_proc - t_begin >= 2) CPX_HIP(h, hipStreamWaitEvent(h->stream, h->ev_back[(max_proc - 2) & 1], 0));
  }
  CPX_HIP(h, hipEventRecord(h->ev1, h->stream));
  if (defer_medians) {
    CPX_HIP(h, hipStreamWaitEvent(h->stream2, h->ev1, 0));  // (the records' memset and the frame kernels are in front of ev1)
    cpx::launch_median(a, B, t_begin, max_proc, h->stream2);
    CPX_HIP(h, hipEventRecord(h->ev_median, h->stream2));
    h->medians_pending = true;
  }
  h->state_packed = a.packed_state != 0;
  h->last_launches = launches;
  h->timing_valid = true;
  if (background_dev) cpx::launch_export_background(a, B, background_dev, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_track_batch_ex(cpx_handle* h, const uint16_t* frames_dev, const int32_t* clip_offsets,
                       const cpx_frame_meta* meta, int B, cpx_component* comps_dev, cpx_frame_info* info_dev,
                       int32_t* labels_dev, float* filtered_dev, float* background_dev, int flags) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !clip_offsets || !meta || B <= 0 || !comps_dev || !info_dev)
    return fail(h, CPX_ERR_INVALID, "cpx_track_batch: null argument");
  h->stream_frames = -1;  // the workspace is re-initialised: an open stream ends here
  return track_run(h, frames_dev, clip_offsets, meta, B, -1, comps_dev, info_dev, labels_dev, filtered_dev,
                   background_dev, flags);
}

int cpx_track_batch(cpx_handle* h, const uint16_t* frames_dev, const int32_t* clip_offsets,
                    const cpx_frame_meta* meta, int B, cpx_component* comps_dev,
                    cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                    float* background_dev) {
  return cpx_track_batch_ex(h, frames_dev, clip_offsets, meta, B, comps_dev, info_dev, labels_dev, filtered_dev,
                            background_dev, 0);
}

int cpx_track_frame_ex(cpx_handle* h, const uint16_t* frames_dev, const cpx_frame_meta* meta, int n_prev, int n_frames,
                       cpx_component* comps_dev, cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                       float* background_dev, int flags) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !meta || !comps_dev || !info_dev || n_prev < 0 || n_frames <= n_prev)
    return fail(h, CPX_ERR_INVALID, "cpx_track_frame: bad argument");
  if (n_prev > 0 && h->stream_frames != n_prev)
    return fail(h, CPX_ERR_INVALID, "cpx_track_frame: n_prev does not match the frames this handle has consumed");
  const int32_t offs[2] = {0, n_frames};
  h->stream_frames = -1;
  if (n_prev == 0) h->stream_assoc_frames = -1;  // a new clip: its association starts fresh
  const int rc = track_run(h, frames_dev, offs, meta, 1, n_prev, comps_dev, info_dev, labels_dev, filtered_dev,
                           background_dev, flags);
  if (rc == CPX_OK) h->stream_frames = n_frames;
  return rc;
}

int cpx_track_frame(cpx_handle* h, const uint16_t* frames_dev, const cpx_frame_meta* meta, int n_prev, int n_frames,
                    cpx_component* comps_dev, cpx_frame_info* info_dev, int32_t* labels_dev, float* filtered_dev,
                    float* background_dev) {
  return cpx_track_frame_ex(h, frames_dev, meta, n_prev, n_frames, comps_dev, info_dev, labels_dev, filtered_dev,
                            background_dev, 0);
}

int cpx_set_background(cpx_handle* h, int clip, const float* background, const double* weights, double average) {
  if (!h) return CPX_ERR_INVALID;
  if (clip < 0 || !background) return fail(h, CPX_ERR_INVALID, "cpx_set_background: bad argument");
  const cpx_config& c = h->cfg;
  const int W = c.width, H = c.height, e = c.edge_pixels;
  const size_t P = (size_t)W * H;
  cpx_handle::StagedBackground sb;
  sb.bg.resize(P);
  sb.kcnt.assign(P, 0);
  sb.average = average;
  for (size_t p = 0; p < P; ++p) {
    const float v = background[p];
    if (!(v >= 0.0f && v <= 65535.0f) || v != std::floor(v))
      return fail(h, CPX_ERR_UNSUPPORTED, "cpx_set_background: the background must be integer-valued in [0, 65535]");
    sb.bg[p] = (uint16_t)v;
  }
  if (weights) {
    // a weight is the k-fold float64 accumulation of weight_add (motiondetector.py:218-222): find k, exactly
    const std::vector<double>& wt = h->wtab_host;
    const int iw = W - 2 * e, ih = H - 2 * e;
    for (int y = 0; y < ih; ++y)
      for (int x = 0; x < iw; ++x) {
        const double wv = weights[(size_t)y * iw + x];
        long k = c.weight_add > 0 ? std::lround(wv / c.weight_add) : 0;
        bool ok = false;
        for (long kk = std::max(0L, k - 1); kk <= k + 1 && kk < h->wtab_len; ++kk)
          if (wt[kk] == wv) {
            k = kk;
            ok = true;
            break;
          }
        if (!ok) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_set_background: a weight is not an accumulation of weight_add");
        sb.kcnt[(size_t)(y + e) * W + (x + e)] = (uint16_t)k;
      }
  }
  h->staged_bg[clip] = std::move(sb);
  return CPX_OK;
}

int cpx_get_background(cpx_handle* h, int clip, float* background, double* weights, double* average) {
  if (!h) return CPX_ERR_INVALID;
  if (clip < 0 || clip >= h->last_B || !h->ws) return fail(h, CPX_ERR_INVALID, "cpx_get_background: no such clip in the last track call");
  CPX_ENTER(h);
  const cpx_config& c = h->cfg;
  const int W = c.width, H = c.height, e = c.edge_pixels;
  const size_t P = (size_t)W * H;
  const WsLayout l = ws_layout(c, h->last_B, h->stream_filt_state);
  char* base = (char*)h->ws;
  if (int urc = unpack_state(h)) return urc;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  cpx::ClipState st;
  CPX_HIP(h, hipMemcpy(&st, base + l.cstate + (size_t)clip * sizeof(cpx::ClipState), sizeof(st), hipMemcpyDeviceToHost));
  if (average) *average = st.bg_average;
  if (background) {
    std::vector<uint16_t> bg(P);
    CPX_HIP(h, hipMemcpy(bg.data(), base + l.bg + ((size_t)clip * 2 + (st.n_done & 1)) * P * sizeof(uint16_t),
                         P * sizeof(uint16_t), hipMemcpyDeviceToHost));
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {  // the interior is authoritative, edges replicate it (motiondetector.py:239-244)
        const int cy = std::min(std::max(y, e), H - 1 - e), cx = std::min(std::max(x, e), W - 1 - e);
        background[(size_t)y * W + x] = (float)bg[(size_t)cy * W + cx];
      }
  }
  if (weights) {
    std::vector<uint16_t> kc(P);
    CPX_HIP(h, hipMemcpy(kc.data(), base + l.kcnt + (size_t)clip * P * sizeof(uint16_t), P * sizeof(uint16_t),
                         hipMemcpyDeviceToHost));
    const std::vector<double>& wt = h->wtab_host;
    const int iw = W - 2 * e, ih = H - 2 * e;
    for (int y = 0; y < ih; ++y)
      for (int x = 0; x < iw; ++x) {
        const int k = kc[(size_t)(y + e) * W + (x + e)];
        weights[(size_t)y * iw + x] = k < h->wtab_len ? wt[k] : (double)k * c.weight_add;
      }
  }
  return CPX_OK;
}

static int assoc_run(cpx_handle* h, const cpx_track_params* params, const int32_t* clip_offsets,
                     const cpx_frame_meta* meta, int B, int n_prev, bool fresh, const cpx_component* comps_dev,
                     const cpx_frame_info* info_dev, cpx_region* pool_dev,
                     cpx_track_record* tracks_dev, int32_t* n_tracks_dev, int32_t* status_dev,
                     cpx_region* regions_dev, int32_t* region_counts_dev) {
  if (params->max_active_tracks < 1 || params->max_tracks < 1)
    return fail(h, CPX_ERR_INVALID, "association: capacities must be positive");
  CPX_ENTER(h);
  const bool resume = n_prev > 0 && !fresh;
  int t_begin = 0;
  if (n_prev > 0)
    for (int f = 0; f < n_prev; ++f) t_begin += meta[f].background_frame ? 0 : 1;
  Schedule sc;
  int rc = build_schedule(h, clip_offsets, meta, B, &sc);
  if (rc != CPX_OK) return rc;
  rc = upload_schedule(h, sc, B);
  if (rc != CPX_OK) return rc;
  const int n = std::max((int)sc.proc_idx.size(), 1);
  const int cap = h->cfg.max_components, ma = params->max_active_tracks;
  size_t off = 0;
  const size_t o_active = off;
  off = align_up(off + (size_t)B * ma * cpx::assoc_active_bytes(), 256);
  const size_t o_regs = off;
  off = align_up(off + (size_t)B * cap * sizeof(cpx_region), 256);
  const size_t o_scores = off;
  off = align_up(off + (size_t)B * cap * ma * cpx::assoc_score_bytes(), 256);
  const size_t o_used = off;
  off = align_up(off + (size_t)B * cap, 256);
  const size_t o_resume = off;
  off = align_up(off + (size_t)B * sizeof(cpx::AssocResume), 256);
  if (resume && off > h->ws_assoc_bytes)
    return fail(h, CPX_ERR_INVALID, "cpx_associate_frame: the stream's association state is gone");
  if (off > h->ws_assoc_bytes) {
    if (h->ws_assoc) hipFree(h->ws_assoc);
    h->ws_assoc = nullptr;
    h->ws_assoc_bytes = 0;
    hipError_t e = hipMalloc(&h->ws_assoc, off);
    if (e != hipSuccess) return fail(h, CPX_ERR_NOMEM, "association workspace hipMalloc", e);
    h->ws_assoc_bytes = off;
  }
  char* base = (char*)h->ws_assoc;
  cpx::AssocArgs a{};
  a.B = B;
  a.cap = cap;
  a.params = *params;
  a.clip_first = h->sched_dev;
  a.proc_off = h->sched_dev + B;
  a.proc_idx = h->sched_dev + B + (B + 1);
  a.proc_ffc = h->sched_dev + B + (B + 1) + n;
  a.comps = comps_dev;
  a.info = info_dev;
  a.pool = pool_dev;
  a.tracks = tracks_dev;
  a.n_tracks = n_tracks_dev;
  a.status = status_dev;
  a.regions_out = regions_dev;
  a.region_counts = region_counts_dev;
  a.active = (cpx::ActiveTrack*)(base + o_active);
  a.regs = (cpx_region*)(base + o_regs);
  a.scores = (cpx::ScoreRec*)(base + o_scores);
  a.used = (unsigned char*)(base + o_used);
  a.resume = (cpx::AssocResume*)(base + o_resume);
  a.t_begin = t_begin;
  a.fresh = resume ? 0 : 1;
  cpx::launch_assoc(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_associate_batch(cpx_handle* h, const cpx_track_params* params, const int32_t* clip_offsets,
                        const cpx_frame_meta* meta, int B, const cpx_component* comps_dev,
                        const cpx_frame_info* info_dev, cpx_region* pool_dev,
                        cpx_track_record* tracks_dev, int32_t* n_tracks_dev, int32_t* status_dev,
                        cpx_region* regions_dev, int32_t* region_counts_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!params || !clip_offsets || !meta || B <= 0 || !comps_dev || !info_dev || !pool_dev || !tracks_dev ||
      !n_tracks_dev || !status_dev)
    return fail(h, CPX_ERR_INVALID, "cpx_associate_batch: null argument");
  h->stream_assoc_frames = -1;
  return assoc_run(h, params, clip_offsets, meta, B, -1, true, comps_dev, info_dev, pool_dev, tracks_dev, n_tracks_dev,
                   status_dev, regions_dev, region_counts_dev);
}

int cpx_associate_frame(cpx_handle* h, const cpx_track_params* params, const cpx_frame_meta* meta, int n_prev,
                        int n_frames, const cpx_component* comps_dev, const cpx_frame_info* info_dev,
                        cpx_region* pool_dev, cpx_track_record* tracks_dev, int32_t* n_tracks_dev,
                        int32_t* status_dev, cpx_region* regions_dev, int32_t* region_counts_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!params || !meta || !comps_dev || !info_dev || !pool_dev || !tracks_dev || !n_tracks_dev || !status_dev ||
      n_prev < 0 || n_frames <= n_prev)
    return fail(h, CPX_ERR_INVALID, "cpx_associate_frame: bad argument");
  // frames in [stream_assoc_frames, n_prev) were never handed to the association: they stay un-tracked
  const bool fresh = h->stream_assoc_frames < 0;
  if (!fresh && h->stream_assoc_frames > n_prev)
    return fail(h, CPX_ERR_INVALID, "cpx_associate_frame: n_prev is behind the frames this handle has consumed");
  const int32_t offs[2] = {0, n_frames};
  h->stream_assoc_frames = -1;
  const int rc = assoc_run(h, params, offs, meta, 1, n_prev, fresh, comps_dev, info_dev, pool_dev, tracks_dev, n_tracks_dev,
                           status_dev, regions_dev, region_counts_dev);
  if (rc == CPX_OK) h->stream_assoc_frames = n_frames;
  return rc;
}

static cpx::ClassifyArgs classify_args(const cpx_handle* h) {
  cpx::ClassifyArgs a{};
  const cpx_config& c = h->cfg;
  a.W = c.width;
  a.H = c.height;
  a.crop_x = c.edge_pixels;
  a.crop_y = c.edge_pixels;
  a.crop_w = c.width - 2 * c.edge_pixels;
  a.crop_h = c.height - 2 * c.edge_pixels;
  return a;
}

int cpx_track_limits_batch(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                           const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev,
                           const int32_t* track_offsets_dev, int n_tracks, cpx_track_limits* limits_dev) {
  return cpx_track_limits_batch_ex(h, frames_dev, filtered_dev, info_dev, refs_dev, track_offsets_dev, n_tracks,
                                   limits_dev, 0);
}

int cpx_track_limits_batch_ex(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                              const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev,
                              const int32_t* track_offsets_dev, int n_tracks, cpx_track_limits* limits_dev, int flags) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !filtered_dev || !info_dev || !refs_dev || !track_offsets_dev || !limits_dev || n_tracks < 0 ||
      (flags & ~(CPX_LIMITS_POST_PROCESS | CPX_LIMITS_THERMAL_DIFF_NORM | CPX_LIMITS_NO_DIFF_NORM | CPX_LIMITS_ALWAYS_CLIP |
                 CPX_LIMITS_SWAP_CHANNELS | CPX_LIMITS_TF_SCALING)))
    return fail(h, CPX_ERR_INVALID, "cpx_track_limits_batch: bad argument");
  if (n_tracks == 0) return CPX_OK;
  CPX_ENTER(h);
  if (int jrc = join_medians(h)) return jrc;  // (reads cpx_frame_info.thermal_median)
  cpx::ClassifyArgs a = classify_args(h);
  a.limits_flags = flags;
  a.frames = frames_dev;
  a.filtered = filtered_dev;
  a.info = info_dev;
  a.refs = refs_dev;
  a.track_offsets = track_offsets_dev;
  a.limits = limits_dev;
  cpx::launch_limits(a, n_tracks, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_crop_tile(cpx_handle* h, const uint16_t* frames_dev, const float* filtered_dev,
                  const cpx_frame_info* info_dev, const cpx_crop_req* reqs_dev, int n_reqs,
                  const cpx_track_limits* limits_dev, int frame_size, int square_width, float* out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !filtered_dev || !info_dev || !reqs_dev || !limits_dev || !out_dev || n_reqs < 0)
    return fail(h, CPX_ERR_INVALID, "cpx_crop_tile: null argument");
  if (frame_size < 1 || frame_size > 128 || square_width < 1 || square_width > 16)
    return fail(h, CPX_ERR_UNSUPPORTED, "cpx_crop_tile: frame_size must be 1..128, square_width 1..16");
  if (n_reqs == 0) return CPX_OK;
  CPX_ENTER(h);
  if (int jrc = join_medians(h)) return jrc;  // (reads cpx_frame_info.thermal_median)
  cpx::ClassifyArgs a = classify_args(h);
  a.frames = frames_dev;
  a.filtered = filtered_dev;
  a.info = info_dev;
  a.limits = const_cast<cpx_track_limits*>(limits_dev);
  a.reqs = reqs_dev;
  a.out = out_dev;
  a.frame_size = frame_size;
  a.square_width = square_width;
  cpx::launch_crop(a, n_reqs, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

// a 1x1 shortcut convolution folded into the convolution that would have read its output as the residual
struct conv_fuse {
  const float* in = nullptr;  // [N, H, W, cin]
  const float* w = nullptr;
  const float* bias = nullptr;
  int H = 0, W = 0, cin = 0, stride = 1;
};
static bool conv_can_fuse(const cpx_handle* h, const cpx_conv_desc* d);

// the modes that run the split-operand kernels (16-bit planes on the bf16 / fp16 matrix pipe)
static bool split_math(const cpx_handle* h) { return h->cnn_math != CPX_CNN_MATH_F32; }
// CPX_CNN_MATH_FP16X2: what a network's forward knows about the layer and a bare cpx_conv2d does not
struct conv_half {
  float act_scale = 1.0f;   // power of two the activated input is multiplied by before the fp16 split
  bool keep_flag = false;   // the overflow word belongs to the forward in flight (cleared once, at its start)
  int word = 0;             // which overflow word: 0 = a bare convolution's, 2 + b = block b of the forward in flight
  // producer-side split between a block's two convolutions (cpx_cnn_forward decides; ConvArgs::out_planes / in_planes)
  bool out_planes = false;  // store the output as the next layer's fp16 planes, scaled by out_act_scale
  float out_act_scale = 1.0f;
  bool in_planes = false;   // the input is in that form
  // the fp16 work of this layer was done by a fused block launch (conv_block32_kernel): only the guarded three-plane
  // rerun is launched, and no timing record is taken (the block launch has its own)
  bool rerun_only = false;
};
// the handle's overflow words: [0] the last bare convolution's / whether the last forward raised any, [1] forwards that did,
// [2 + b] block b of the forward in flight.  One word per BLOCK, not per forward: an activation out of fp16's range sends
// the rest of ITS block (the two convolutions hand fp16 planes to each other) to the bf16x3 kernels; the next block is
// back on the fp16 ones
constexpr int OVF_WORDS = 2 + 3 * CPX_WRRESNET_MAX_BLOCKS;
static int ensure_ovf_word(cpx_handle* h) {
  if (h->cnn_ovf) return CPX_OK;
  if (hipMalloc((void**)&h->cnn_ovf, OVF_WORDS * sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    return fail(h, CPX_ERR_NOMEM, "cpx_conv2d: overflow word allocation failed");
  }
  CPX_HIP(h, hipMemsetAsync(h->cnn_ovf, 0, OVF_WORDS * sizeof(int), h->stream));
  return CPX_OK;
}
// split_weights: the bf16 plane image of d->weights_dev if the caller (a cpx_cnn) keeps one, else NULL
static int conv_run(cpx_handle* h, const cpx_conv_desc* d, const void* split_weights, const conv_fuse* fuse = nullptr,
                    const conv_half* hf = nullptr) {
  if (!h) return CPX_ERR_INVALID;
  if (!d || !d->in_dev || !d->out_dev || !d->weights_dev) return fail(h, CPX_ERR_INVALID, "cpx_conv2d: null argument");
  if (d->N < 1 || d->H < 1 || d->W < 1 || d->groups < 1 || d->Cin % d->groups || d->Cout % d->groups ||
      d->ksize < 1 || d->stride < 1 || (d->in_scale_dev == nullptr) != (d->in_shift_dev == nullptr))
    return fail(h, CPX_ERR_INVALID, "cpx_conv2d: bad descriptor");
  CPX_ENTER(h);
  cpx::ConvArgs a{};
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups;
  a.ksize = d->ksize; a.stride = d->stride; a.relu = d->relu;
  if (d->pad_same) {  // TensorFlow SAME: out = ceil(in / stride), surplus padding goes to the bottom / right
    a.Ho = (d->H + d->stride - 1) / d->stride;
    a.Wo = (d->W + d->stride - 1) / d->stride;
    const int ph = std::max((a.Ho - 1) * d->stride + d->ksize - d->H, 0);
    const int pw = std::max((a.Wo - 1) * d->stride + d->ksize - d->W, 0);
    a.pad_top = ph / 2;
    a.pad_left = pw / 2;
  } else {
    if (d->H < d->ksize || d->W < d->ksize) return fail(h, CPX_ERR_INVALID, "cpx_conv2d: input smaller than kernel");
    a.Ho = (d->H - d->ksize) / d->stride + 1;
    a.Wo = (d->W - d->ksize) / d->stride + 1;
    a.pad_top = a.pad_left = 0;
  }
  a.in = d->in_dev; a.out = d->out_dev; a.weights = d->weights_dev;
  a.in_scale = d->in_scale_dev; a.in_shift = d->in_shift_dev;
  a.out_scale = d->out_scale_dev; a.out_shift = d->out_shift_dev; a.residual = d->residual_dev;
  if (fuse) {
    if (!(split_math(h) && cpx::conv_bf3_supported(a)) || a.out_scale || a.residual)
      return fail(h, CPX_ERR_INVALID, "conv_run: shortcut fusion needs the split-operand kernel, no output scale, no residual");
    a.sc_in = fuse->in; a.sc_w = fuse->w; a.sc_bias = fuse->bias;
    a.sc_H = fuse->H; a.sc_W = fuse->W; a.sc_cin = fuse->cin; a.sc_stride = fuse->stride;
  }
  cpx_handle::ConvEv ev{};
  const bool timed = h->conv_timing && !(hf && hf->rerun_only);
  if (timed) {
    ev.key = (a.Cin / a.groups) * 10000 + (a.Cout / a.groups) * 10 + a.stride + (a.ksize == 1 ? 5 : 0);
    ev.flops = 2.0 * a.N * a.Ho * a.Wo * a.Cout * (double)(a.Cin / a.groups) * a.ksize * a.ksize;
    if (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess)
      return fail(h, CPX_ERR_HIP, "cpx_conv2d: event creation failed");
    CPX_HIP(h, hipEventRecord(ev.e0, h->stream));
  }
  int rc;
  if (split_math(h) && cpx::conv_bf3_supported(a)) {
    a.planes = h->cnn_math == CPX_CNN_MATH_BF16X2 ? 2 : 3;
    // fp16x2: the two-plane layers run on fp16 planes, with the three-plane kernel launched behind as the guarded
    // rerun (it returns at once unless a scaled activation left fp16's range); every other layer as bf16x3
    // (an output that aliases the residual or the input -- an in-place add -- must not be written twice: the guarded
    // rerun would read what the fp16 pass has already stored.  Such a call runs bf16x3 directly.)
    const bool aliased = a.out == a.residual || a.out == a.in;
    const bool half = h->cnn_math == CPX_CNN_MATH_FP16X2 && cpx::conv_bf3_two_planes(a) && !aliased;
    const bool planes_out = h->cnn_math == CPX_CNN_MATH_FP16X2 && hf && hf->out_planes;
    const bool rerun = hf && hf->rerun_only;  // (any split-operand layer: the 8-channel one of a fused first block too)
    if (half || planes_out || rerun) {
      const int rco = ensure_ovf_word(h);
      if (rco != CPX_OK) return rco;
      if (!(hf && hf->keep_flag)) CPX_HIP(h, hipMemsetAsync(h->cnn_ovf, 0, sizeof(int), h->stream));
    }
    if (!split_weights) {
      const size_t need = cpx::conv_bf3_weight_bytes(a);
      if (need > h->bf3_scratch_bytes) {
        CPX_HIP(h, hipStreamSynchronize(h->stream));
        if (h->bf3_scratch) hipFree(h->bf3_scratch);
        h->bf3_scratch = nullptr;
        h->bf3_scratch_bytes = 0;
        if (hipMalloc(&h->bf3_scratch, need) != hipSuccess) {
          (void)hipGetLastError();
          return fail(h, CPX_ERR_NOMEM, "cpx_conv2d: weight scratch allocation failed");
        }
        h->bf3_scratch_bytes = need;
      }
      cpx::launch_split_weights(a, h->bf3_scratch, h->stream);
      split_weights = h->bf3_scratch;
    }
    if (half || planes_out || rerun) {
      cpx::ConvArgs ah = a;
      if (half) {
        ah.planes = 2;
        ah.half = 1;
        ah.act_scale = hf ? hf->act_scale : 1.0f;
        ah.act_unscale = 1.0f / ah.act_scale;  // (a power of two: exact)
        ah.in_planes = hf && hf->in_planes;
      }
      ah.ovf = h->cnn_ovf + (hf ? hf->word : 0);
      if (planes_out) {
        ah.out_planes = 1;
        ah.out_act_scale = hf->out_act_scale;
      }
      rc = rerun ? 0 : cpx::launch_conv_bf3(ah, split_weights, h->stream);
      a.guard = h->cnn_ovf + (hf ? hf->word : 0);
      if (rc == 0) rc = cpx::launch_conv_bf3(a, split_weights, h->stream);
    } else {
      rc = cpx::launch_conv_bf3(a, split_weights, h->stream);
    }
    if (rc == -3) {  // more tiles than the split-operand kernel's tile decomposition indexes: float32 path
      // the float32 kernel has no fused shortcut: dropping it silently would lose the block's shortcut branch
      if (fuse) return fail(h, CPX_ERR_UNSUPPORTED, "conv_run: batch too large for the fused-shortcut kernel (split the call)");
      rc = cpx::launch_conv(a, h->stream);
    }
  } else {
    if (hf && hf->rerun_only) {  // (conv1_1 behind a fused first block that computed it: only that block's rerun needs the tensor)
      const int rco = ensure_ovf_word(h);
      if (rco != CPX_OK) return rco;
      a.guard = h->cnn_ovf + hf->word;
    }
    rc = cpx::launch_conv(a, h->stream);
  }
  if (timed) {
    CPX_HIP(h, hipEventRecord(ev.e1, h->stream));
    h->conv_events.push_back(ev);
  }
  if (rc == -2) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_conv2d: no kernel for this (channels per group, stride, kernel size)");
  if (rc != 0) return fail(h, CPX_ERR_HIP, "cpx_conv2d: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_conv2d(cpx_handle* h, const cpx_conv_desc* d) { return conv_run(h, d, nullptr); }

// the 3x3 stride-1 convolution described by d runs on the split-operand kernel (which can absorb a 1x1 shortcut)
static bool conv_can_fuse(const cpx_handle* h, const cpx_conv_desc* d) {
  if (!split_math(h) || d->out_scale_dev || d->groups < 1) return false;
  cpx::ConvArgs a{};
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups; a.ksize = d->ksize; a.stride = d->stride;
  return cpx::conv_bf3_supported(a);
}

int cpx_set_cnn_math(cpx_handle* h, int mode) {
  if (!h) return CPX_ERR_INVALID;
  if (mode != CPX_CNN_MATH_F32 && mode != CPX_CNN_MATH_BF16X3 && mode != CPX_CNN_MATH_BF16X2 && mode != CPX_CNN_MATH_FP16X2)
    return fail(h, CPX_ERR_INVALID, "cpx_set_cnn_math: unknown mode");
  h->cnn_math = mode;
  return CPX_OK;
}
int cpx_get_cnn_math(const cpx_handle* h) { return h ? h->cnn_math : CPX_ERR_INVALID; }

int cpx_cnn_overflow_forwards(cpx_handle* h, int* count, int reset) {
  if (!h) return CPX_ERR_INVALID;
  if (!count) return fail(h, CPX_ERR_INVALID, "cpx_cnn_overflow_forwards: null argument");
  CPX_ENTER(h);
  *count = 0;
  if (!h->cnn_ovf) return CPX_OK;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  CPX_HIP(h, hipMemcpy(count, h->cnn_ovf + 1, sizeof(int), hipMemcpyDeviceToHost));
  if (reset) CPX_HIP(h, hipMemset(h->cnn_ovf + 1, 0, sizeof(int)));
  return CPX_OK;
}

int cpx_cnn_last_overflow(cpx_handle* h, int* overflowed) {
  if (!h) return CPX_ERR_INVALID;
  if (!overflowed) return fail(h, CPX_ERR_INVALID, "cpx_cnn_last_overflow: null argument");
  CPX_ENTER(h);
  *overflowed = 0;
  if (!h->cnn_ovf) return CPX_OK;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  CPX_HIP(h, hipMemcpy(overflowed, h->cnn_ovf, sizeof(int), hipMemcpyDeviceToHost));
  return CPX_OK;
}

int cpx_cnn_head_ex(cpx_handle* h, const cpx_head_desc* d) {
  if (!h) return CPX_ERR_INVALID;
  if (!d || !d->in_dev || !d->bn_scale_dev || !d->bn_shift_dev || !d->dense_w_dev || !d->dense_b_dev || !d->logits_dev ||
      d->N < 1 || d->HW < 1 || d->C < 1 || d->L < 1 || d->C > 8192 || d->L > 8192 || d->n_hidden < 0 ||
      d->n_hidden > CPX_HEAD_MAX_HIDDEN || (d->activation != CPX_HEAD_SIGMOID && d->activation != CPX_HEAD_SOFTMAX))
    return fail(h, CPX_ERR_INVALID, "cpx_cnn_head: bad argument");
  CPX_ENTER(h);
  cpx::HeadArgs a{};
  a.N = d->N; a.HW = d->HW; a.C = d->C; a.L = d->L;
  a.n_hidden = d->n_hidden;
  a.activation = d->activation;
  for (int k = 0; k < d->n_hidden; ++k) {
    if (!d->hidden_w_dev[k] || !d->hidden_b_dev[k] || d->hidden_sizes[k] < 1 || d->hidden_sizes[k] > 2048)
      return fail(h, CPX_ERR_INVALID, "cpx_cnn_head: bad hidden layer");
    a.hidden_sizes[k] = d->hidden_sizes[k];
    a.hidden_w[k] = d->hidden_w_dev[k];
    a.hidden_b[k] = d->hidden_b_dev[k];
  }
  a.in = d->in_dev; a.bn_scale = d->bn_scale_dev; a.bn_shift = d->bn_shift_dev;
  a.dense_w = d->dense_w_dev; a.dense_b = d->dense_b_dev; a.logits = d->logits_dev; a.probs = d->probs_dev;
  cpx::launch_head(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_cnn_head(cpx_handle* h, const float* in_dev, int N, int HW, int C, const float* bn_scale_dev,
                 const float* bn_shift_dev, const float* dense_w_dev, const float* dense_b_dev, int L,
                 float* logits_dev, float* probs_dev) {
  if (!h) return CPX_ERR_INVALID;
  cpx_head_desc d{};
  d.N = N; d.HW = HW; d.C = C; d.L = L;
  d.n_hidden = 0;
  d.activation = CPX_HEAD_SIGMOID;
  d.in_dev = in_dev; d.bn_scale_dev = bn_scale_dev; d.bn_shift_dev = bn_shift_dev;
  d.dense_w_dev = dense_w_dev; d.dense_b_dev = dense_b_dev; d.logits_dev = logits_dev; d.probs_dev = probs_dev;
  return cpx_cnn_head_ex(h, &d);
}

static int final_common(cpx_handle* h, const cpx_filter_params* params, const int32_t* clip_offsets,
                        const cpx_frame_meta* meta, int B, cpx::FinalArgs* a) {
  if (params->max_active_tracks < 1 || params->max_tracks_per_clip < 1)
    return fail(h, CPX_ERR_INVALID, "filter params: capacities must be positive");
  CPX_ENTER(h);
  Schedule sc;
  int rc = build_schedule(h, clip_offsets, meta, B, &sc);
  if (rc != CPX_OK) return rc;
  rc = upload_schedule(h, sc, B);
  if (rc != CPX_OK) return rc;
  const int n = std::max((int)sc.proc_idx.size(), 1);
  a->B = B;
  a->params = *params;
  a->max_frames = h->cfg.max_frames;
  a->clip_first = h->sched_dev;
  a->proc_off = h->sched_dev + B;
  a->proc_idx = h->sched_dev + B + (B + 1);
  a->proc_ffc = h->sched_dev + B + (B + 1) + n;
  return CPX_OK;
}

// per-clip scalar scratch of the end-of-clip kernels: double [B][2 * max_frames] + float [B][max_frames]
static int final_scratch(cpx_handle* h, int B, cpx::FinalArgs* a) {
  const size_t need = (size_t)B * h->cfg.max_frames * (2 * sizeof(double) + sizeof(float)) + 512;
  if (need > h->ws_assoc_bytes) {
    CPX_HIP(h, hipStreamSynchronize(h->stream));
    if (h->ws_assoc) hipFree(h->ws_assoc);
    h->ws_assoc = nullptr;
    h->ws_assoc_bytes = 0;
    hipError_t e = hipMalloc(&h->ws_assoc, need);
    if (e != hipSuccess) return fail(h, CPX_ERR_NOMEM, "finalize workspace hipMalloc", e);
    h->ws_assoc_bytes = need;
  }
  a->scratch_d = (double*)h->ws_assoc;
  a->scratch_f = (float*)((char*)h->ws_assoc + align_up((size_t)B * h->cfg.max_frames * 2 * sizeof(double), 256));
  return CPX_OK;
}

int cpx_finalize_tracks(cpx_handle* h, const cpx_filter_params* params, const int32_t* clip_offsets,
                        const cpx_frame_meta* meta, int B, const cpx_region* pool_dev,
                        const cpx_track_record* tracks_dev, const int32_t* n_tracks_dev,
                        cpx_track_summary* summaries_dev, int32_t* counts_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!params || !clip_offsets || !meta || B <= 0 || !pool_dev || !tracks_dev || !n_tracks_dev || !summaries_dev ||
      !counts_dev)
    return fail(h, CPX_ERR_INVALID, "cpx_finalize_tracks: null argument");
  cpx::FinalArgs a{};
  int rc = final_common(h, params, clip_offsets, meta, B, &a);
  if (rc != CPX_OK) return rc;
  rc = final_scratch(h, B, &a);
  if (rc != CPX_OK) return rc;
  a.square_width = 5;
  a.pool = pool_dev;
  a.tracks = tracks_dev;
  a.n_tracks = n_tracks_dev;
  a.summaries = summaries_dev;
  a.counts = counts_dev;
  cpx::launch_finalize(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_plan_segments(cpx_handle* h, const cpx_filter_params* params, const int32_t* clip_offsets,
                      const cpx_frame_meta* meta, int B, const cpx_region* pool_dev,
                      const cpx_track_summary* summaries_dev, const int32_t* n_tracks_dev,
                      const int32_t* prefix_dev, int square_width, cpx_region_ref* refs_dev,
                      int32_t* track_offsets_dev, cpx_crop_req* reqs_dev, int32_t* sample_track_dev,
                      int32_t* track_clip_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!params || !clip_offsets || !meta || B <= 0 || !pool_dev || !summaries_dev || !n_tracks_dev || !prefix_dev ||
      !refs_dev || !track_offsets_dev || !reqs_dev || !sample_track_dev || !track_clip_dev)
    return fail(h, CPX_ERR_INVALID, "cpx_plan_segments: null argument");
  if (square_width != 5) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_plan_segments: square_width must be 5");
  cpx::FinalArgs a{};
  int rc = final_common(h, params, clip_offsets, meta, B, &a);
  if (rc != CPX_OK) return rc;
  rc = final_scratch(h, B, &a);
  if (rc != CPX_OK) return rc;
  a.square_width = square_width;
  a.pool = pool_dev;
  a.summaries = const_cast<cpx_track_summary*>(summaries_dev);
  a.n_tracks = n_tracks_dev;
  a.prefix = prefix_dev;
  a.refs = refs_dev;
  a.track_offsets = track_offsets_dev;
  a.reqs = reqs_dev;
  a.sample_track = sample_track_dev;
  a.track_clip = track_clip_dev;
  cpx::launch_plan(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_counts_prefix(cpx_handle* h, const int32_t* counts_dev, int B, int32_t* prefix_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!counts_dev || !prefix_dev || B <= 0) return fail(h, CPX_ERR_INVALID, "cpx_counts_prefix: bad argument");
  CPX_ENTER(h);
  cpx::launch_counts_prefix(counts_dev, B, prefix_dev, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_aggregate_predictions(cpx_handle* h, const float* probs_dev, const int32_t* sample_track_dev,
                              int n_samples, const cpx_crop_req* reqs_dev, int n_tracks, int n_labels,
                              int false_positive_index, int square_width, float* scores_dev, int32_t* best_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (n_tracks == 0) return CPX_OK;
  if (!probs_dev || !sample_track_dev || !reqs_dev || !scores_dev || !best_dev || n_samples < 0 || n_tracks < 0 ||
      n_labels < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_aggregate_predictions: bad argument");
  CPX_ENTER(h);
  cpx::AggregateArgs a{};
  a.n_samples = n_samples; a.n_tracks = n_tracks; a.n_labels = n_labels; a.fp_index = false_positive_index;
  a.square_width = square_width;
  a.probs = probs_dev; a.sample_track = sample_track_dev; a.reqs = reqs_dev; a.scores = scores_dev; a.best = best_dev;
  cpx::launch_aggregate(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_conv_timing_enable(cpx_handle* h, int enable) {
  if (!h) return CPX_ERR_INVALID;
  for (auto& e : h->conv_events) {
    hipEventDestroy(e.e0);
    hipEventDestroy(e.e1);
  }
  h->conv_events.clear();
  h->conv_timing = enable != 0;
  return CPX_OK;
}

int cpx_conv_timing_report(cpx_handle* h, cpx_conv_timing* out, int cap, int* n_out) {
  if (!h || !out || !n_out || cap < 1) return CPX_ERR_INVALID;
  CPX_HIP(h, hipStreamSynchronize(h->stream));
  int n = 0;
  for (auto& e : h->conv_events) {
    float ms = 0.f;
    CPX_HIP(h, hipEventElapsedTime(&ms, e.e0, e.e1));
    int i = 0;
    for (; i < n; ++i)
      if (out[i].key == e.key) break;
    if (i == n) {
      if (n == cap) return fail(h, CPX_ERR_OVERFLOW, "cpx_conv_timing_report: more kernel variants than capacity");
      out[n].key = e.key;
      out[n].launches = 0;
      out[n].total_ms = 0.0;
      out[n].flops = 0.0;
      n += 1;
    }
    out[i].launches += 1;
    out[i].total_ms += ms;
    out[i].flops += e.flops;
  }
  *n_out = n;
  return CPX_OK;
}

int cpx_cptv_unpack(cpx_handle* h, const uint8_t* payload_dev, const int64_t* frame_offsets_dev,
                    const int32_t* bit_widths_dev, const int32_t* clip_offsets_dev, int B,
                    uint16_t* frames_out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!payload_dev || !frame_offsets_dev || !bit_widths_dev || !clip_offsets_dev || !frames_out_dev || B < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_cptv_unpack: bad argument");
  CPX_ENTER(h);
  cpx::CptvArgs a{};
  a.W = h->cfg.width;
  a.H = h->cfg.height;
  a.payload = payload_dev;
  a.frame_offsets = (const long long*)frame_offsets_dev;
  a.bit_widths = bit_widths_dev;
  a.clip_offsets = clip_offsets_dev;
  a.frames_out = frames_out_dev;
  const int rc = cpx::launch_cptv_unpack(a, B, h->stream);
  if (rc == -2) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_cptv_unpack: resolution too large for the unpack kernel");
  if (rc != 0) return fail(h, CPX_ERR_HIP, "cpx_cptv_unpack: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_cptv_inflate(cpx_handle* h, const uint8_t* in_dev, const cpx_cptv_file* files_dev, int B, uint8_t* out_dev,
                     cpx_cptv_frame_slot* slots_dev, uint8_t* header_dev, cpx_cptv_file_result* results_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!in_dev || !files_dev || !out_dev || !slots_dev || !results_dev || B < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_cptv_inflate: bad argument");
  CPX_ENTER(h);
  cpx::CptvInflateArgs a{};
  a.B = B;
  a.in = in_dev;
  a.files = files_dev;
  a.out = out_dev;
  a.slots = slots_dev;
  a.header = header_dev;
  a.results = results_dev;
  cpx::launch_cptv_inflate(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_cptv_gather_index(cpx_handle* h, const cpx_cptv_frame_slot* slots_dev, const int64_t* slot_offsets_dev,
                          const int32_t* clip_offsets_dev, int B, int64_t* frame_offsets_dev, int32_t* bit_widths_dev,
                          cpx_cptv_frame_slot* slots_out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!slots_dev || !slot_offsets_dev || !clip_offsets_dev || !frame_offsets_dev || !bit_widths_dev || B < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_cptv_gather_index: bad argument");
  CPX_ENTER(h);
  cpx::CptvGatherArgs a{};
  a.slots = slots_dev;
  a.slot_offsets = (const long long*)slot_offsets_dev;
  a.clip_offsets = clip_offsets_dev;
  a.frame_offsets = (long long*)frame_offsets_dev;
  a.bit_widths = bit_widths_dev;
  a.slots_out = slots_out_dev;
  cpx::launch_cptv_gather(a, B, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_thumb_stats_ex(cpx_handle* h, const uint16_t* frames_dev, const int32_t* labels_dev,
                       const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev, int n_refs,
                       cpx_thumb_stat* out_dev, int max_width, int max_height, int chain_capacity) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !labels_dev || !info_dev || !refs_dev || !out_dev || n_refs < 0 || max_width < 1 || max_height < 1 ||
      chain_capacity < 16 || chain_capacity > 32000)
    return fail(h, CPX_ERR_INVALID, "cpx_thumb_stats: bad argument");
  if (n_refs == 0) return CPX_OK;
  CPX_ENTER(h);
  if (int jrc = join_medians(h)) return jrc;  // (reads cpx_frame_info.thermal_median)
  cpx::ThumbArgs a{};
  a.W = h->cfg.width;
  a.H = h->cfg.height;
  a.chain_cap = chain_capacity;
  a.max_w = max_width < a.W ? max_width : a.W;
  a.max_h = max_height < a.H ? max_height : a.H;
  a.frames = frames_dev;
  a.labels = labels_dev;
  a.info = info_dev;
  a.refs = refs_dev;
  a.out = out_dev;
  const int rc = cpx::launch_thumb(a, n_refs, h->stream);
  if (rc == -2) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_thumb_stats: resolution too large for the contour kernel");
  if (rc != 0) return fail(h, CPX_ERR_HIP, "cpx_thumb_stats: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_thumb_stats(cpx_handle* h, const uint16_t* frames_dev, const int32_t* labels_dev,
                    const cpx_frame_info* info_dev, const cpx_region_ref* refs_dev, int n_refs,
                    cpx_thumb_stat* out_dev) {
  if (!h) return CPX_ERR_INVALID;
  return cpx_thumb_stats_ex(h, frames_dev, labels_dev, info_dev, refs_dev, n_refs, out_dev, h->cfg.width, h->cfg.height,
                            8192);
}

int cpx_trackless_thumb(cpx_handle* h, const uint16_t* frames_dev, int frame, int background, int32_t* out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !out_dev || frame < 0 || background < 0)
    return fail(h, CPX_ERR_INVALID, "cpx_trackless_thumb: bad argument");
  CPX_ENTER(h);
  cpx::TracklessArgs a{};
  a.W = h->cfg.width;
  a.H = h->cfg.height;
  a.frame = frame;
  a.background = background;
  a.frames = frames_dev;
  a.out = out_dev;
  const int rc = cpx::launch_trackless(a, h->stream);
  if (rc == -2) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_trackless_thumb: resolution outside the kernel's envelope");
  if (rc != 0) return fail(h, CPX_ERR_HIP, "cpx_trackless_thumb: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_trackless_thumb_batch(cpx_handle* h, const uint16_t* frames_dev, const int32_t* pairs_dev, int n, int32_t* out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !pairs_dev || !out_dev || n < 0)
    return fail(h, CPX_ERR_INVALID, "cpx_trackless_thumb_batch: bad argument");
  if (n == 0) return CPX_OK;
  CPX_ENTER(h);
  cpx::TracklessArgs a{};
  a.W = h->cfg.width;
  a.H = h->cfg.height;
  a.frames = frames_dev;
  a.out = out_dev;
  a.pairs = pairs_dev;
  a.n = n;
  const int rc = cpx::launch_trackless(a, h->stream);
  if (rc == -2) return fail(h, CPX_ERR_UNSUPPORTED, "cpx_trackless_thumb_batch: resolution outside the kernel's envelope");
  if (rc != 0) return fail(h, CPX_ERR_HIP, "cpx_trackless_thumb_batch: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

// ---- whole-network forward ------------------------------------------------------------------------------------------
struct cpx_cnn {
  cpx_handle* h = nullptr;
  cpx_wrresnet_params p{};
  std::vector<std::pair<const float*, void*>> split;  // bf16 plane images of the 3x3 stride-1 weights
  // CPX_CNN_MATH_FP16X2: the power of two each 3x3 convolution's activated input is multiplied by before the fp16 split
  // ([stage][block][a / b]; 1 until cpx_cnn_set_activation_bounds says more)
  float act_scale[3][CPX_WRRESNET_MAX_BLOCKS][2];
  cpx_cnn() {
    for (auto& st : act_scale)
      for (auto& b : st) b[0] = b[1] = 1.0f;
  }
  const void* split_of(const float* w) const {
    for (const auto& e : split)
      if (e.first == w) return e.second;
    return nullptr;
  }
};

static void cnn_free(cpx_cnn* c) {
  for (auto& e : c->split) hipFree(e.second);
  delete c;
}

static_assert(sizeof(cpx_wrresnet_block) == 56 && sizeof(cpx_wrresnet_params) == 1560,
              "cpx_wrresnet_params layout is part of the ABI");

int cpx_cnn_create(cpx_handle* h, const cpx_wrresnet_params* params, cpx_cnn** out) {
  if (!h) return CPX_ERR_INVALID;
  if (!params || !out) return fail(h, CPX_ERR_INVALID, "cpx_cnn_create: null argument");
  *out = nullptr;
  const cpx_wrresnet_params& p = *params;
  if (p.n_labels < 1 || p.blocks_per_stage < 1 || p.blocks_per_stage > CPX_WRRESNET_MAX_BLOCKS || p.groups < 1 ||
      p.in_channels < 1 || !p.conv1_w || !p.final_scale || !p.final_shift || !p.dense_w || !p.dense_b ||
      p.n_hidden < 0 || p.n_hidden > CPX_HEAD_MAX_HIDDEN ||
      (p.activation != CPX_HEAD_SIGMOID && p.activation != CPX_HEAD_SOFTMAX))
    return fail(h, CPX_ERR_INVALID, "cpx_cnn_create: bad network description");
  for (int k = 0; k < p.n_hidden; ++k)
    if (!p.hidden_w[k] || !p.hidden_b[k] || p.hidden_sizes[k] < 1 || p.hidden_sizes[k] > 2048)
      return fail(h, CPX_ERR_INVALID, "cpx_cnn_create: bad hidden dense layer");
  for (int st = 0; st < 3; ++st) {
    if (!p.shortcut_w[st]) return fail(h, CPX_ERR_INVALID, "cpx_cnn_create: missing shortcut weights");
    for (int d = 0; d < p.blocks_per_stage; ++d) {
      const cpx_wrresnet_block& b = p.block[st][d];
      if (!b.in_scale || !b.in_shift || !b.wa || !b.wb)
        return fail(h, CPX_ERR_INVALID, "cpx_cnn_create: missing block parameters");
    }
  }
  cpx_cnn* c = new (std::nothrow) cpx_cnn();
  if (!c) return fail(h, CPX_ERR_NOMEM, "cpx_cnn_create: out of memory");
  c->h = h;
  c->p = p;
  h->cnns.push_back(c);
  // the weights are constant for the life of the network: split them once (the images are used when the handle's
  // math mode is bf16x3 at forward time)
  CPX_ENTER(h);
  int c_in = p.filters[0];
  for (int st = 0; st < 3; ++st) {
    const int f = p.filters[st + 1];
    for (int d = 0; d < p.blocks_per_stage; ++d) {
      const cpx_wrresnet_block& b = p.block[st][d];
      const float* ws[2] = {b.wa, b.wb};
      for (int k = 0; k < 2; ++k) {
        cpx::ConvArgs a{};
        a.Cin = k == 0 ? c_in : f;
        a.Cout = f;
        a.groups = p.groups;
        a.ksize = 3;
        a.stride = (k == 0 && d == 0) ? st + 1 : 1;
        a.weights = ws[k];
        if (a.Cin % a.groups || a.Cout % a.groups || !cpx::conv_bf3_supported(a) || c->split_of(ws[k])) continue;
        void* img = nullptr;
        if (hipMalloc(&img, cpx::conv_bf3_weight_bytes(a)) != hipSuccess) {
          (void)hipGetLastError();
          cpx_cnn_destroy(c);
          return fail(h, CPX_ERR_NOMEM, "cpx_cnn_create: weight image allocation failed");
        }
        c->split.emplace_back(ws[k], img);
        cpx::launch_split_weights(a, img, h->stream);
      }
      c_in = f;
    }
  }
  CPX_HIP(h, hipGetLastError());
  *out = c;
  return CPX_OK;
}

void cpx_cnn_destroy(cpx_cnn* cnn) {
  if (!cnn) return;
  cpx_handle* h = cnn->h;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  h->cnns.erase(std::remove(h->cnns.begin(), h->cnns.end(), cnn), h->cnns.end());
  cnn_free(cnn);
}

int cpx_cnn_set_activation_bounds(cpx_cnn* cnn, const float* bounds, int n) {
  if (!cnn) return CPX_ERR_INVALID;
  cpx_handle* h = cnn->h;
  const cpx_wrresnet_params& p = cnn->p;
  if (!bounds || n != 3 * p.blocks_per_stage * 2)
    return fail(h, CPX_ERR_INVALID, "cpx_cnn_set_activation_bounds: expected 3 * blocks_per_stage * 2 bounds");
  for (int st = 0; st < 3; ++st)
    for (int d = 0; d < p.blocks_per_stage; ++d)
      for (int k = 0; k < 2; ++k) {
        const float b = bounds[(st * p.blocks_per_stage + d) * 2 + k];
        // the largest power of two that keeps bound * scale at or below 2^12, between 1 and 2^14; no usable bound: 1.
        // (2^12, not 2^15: sixteen times the bound still fits fp16 -- a bound from BatchNorm statistics is a guess, and
        // headroom is cheap: the low plane of every activation above 2^-3 / scale keeps all its bits either way)
        int e = 0;
        if (b > 0.0f && std::isfinite(b)) {
          int eb = 0;
          (void)std::frexp(b, &eb);  // b = f 2^eb, f in [0.5, 1): b <= 2^eb
          e = std::min(std::max(12 - eb, 0), 14);
        }
        cnn->act_scale[st][d][k] = std::ldexp(1.0f, e);
      }
  return CPX_OK;
}

int cpx_cnn_forward(cpx_cnn* cnn, const float* in_dev, int N, int H, int W, float* logits_dev, float* probs_dev) {
  if (!cnn) return CPX_ERR_INVALID;
  cpx_handle* h = cnn->h;
  if (!in_dev || !logits_dev || N < 1 || H < 1 || W < 1) return fail(h, CPX_ERR_INVALID, "cpx_cnn_forward: bad argument");
  CPX_ENTER(h);
  const cpx_wrresnet_params& p = cnn->p;
  if (h->cnn_math == CPX_CNN_MATH_FP16X2) {  // the blocks' overflow words start clear
    const int rco = ensure_ovf_word(h);
    if (rco != CPX_OK) return rco;
    CPX_HIP(h, hipMemsetAsync(h->cnn_ovf + 2, 0, (OVF_WORDS - 2) * sizeof(int), h->stream));
  }
  conv_half hf;
  hf.keep_flag = true;
  // largest activation: conv1 output (and the stage-2 tensors at stride 1)
  size_t biggest = 0;
  {
    int hh = H, ww = W;
    biggest = (size_t)N * hh * ww * p.filters[0];
    for (int st = 0; st < 3; ++st) {
      const int s = st + 1;
      hh = (hh + s - 1) / s;
      ww = (ww + s - 1) / s;
      biggest = std::max(biggest, (size_t)N * hh * ww * p.filters[st + 1]);
    }
  }
  biggest = align_up(biggest, 64);
  if (4 * biggest > h->cnn_arena_floats) {
    if (h->cnn_arena) {
      CPX_HIP(h, hipStreamSynchronize(h->stream));
      hipFree(h->cnn_arena);
    }
    h->cnn_arena = nullptr;
    h->cnn_arena_floats = 0;
    hipError_t e = hipMalloc((void**)&h->cnn_arena, 4 * biggest * sizeof(float));
    if (e != hipSuccess) return fail(h, CPX_ERR_NOMEM, "cpx_cnn_forward: activation hipMalloc", e);
    h->cnn_arena_floats = 4 * biggest;
  }
  float* act[2] = {h->cnn_arena, h->cnn_arena + biggest};
  float* mid = h->cnn_arena + 2 * biggest;
  float* sc = h->cnn_arena + 3 * biggest;
  auto conv = [&](const float* in, float* out, const float* w, int hh, int ww, int cin, int cout, int k, int stride,
                  int same, int relu, const float* in_scale, const float* in_shift, const float* out_scale,
                  const float* out_shift, const float* residual) {
    cpx_conv_desc d{};
    d.N = N; d.H = hh; d.W = ww; d.Cin = cin; d.Cout = cout; d.groups = p.groups; d.ksize = k; d.stride = stride;
    d.pad_same = same; d.relu = relu;
    d.in_dev = in; d.out_dev = out; d.weights_dev = w; d.in_scale_dev = in_scale; d.in_shift_dev = in_shift;
    d.out_scale_dev = out_scale; d.out_shift_dev = out_shift; d.residual_dev = residual;
    return conv_run(h, &d, cnn->split_of(w), nullptr, &hf);
  };
  // conv1_1.  fp16x2 with the stage-2 first block fused: that block's kernel computes this layer while it stages its patch
  // (conv_block32_kernel<true, true>) and the launch here becomes the block's guarded rerun's -- unless the block turns out not
  // to be fusable, in which case it is launched in front of it as ever
  auto conv1 = [&]() {
    return conv(in_dev, act[0], p.conv1_w, H, W, p.in_channels, p.filters[0], 3, 1, 1, 0, nullptr, nullptr, nullptr, p.conv1_b,
                nullptr);
  };
  bool c1_pending = h->cnn_math == CPX_CNN_MATH_FP16X2 && h->block_fusion >= 2 && h->fuse_shortcut && h->fuse_conv1 &&
                    p.groups == 2 && p.in_channels == 2 && p.filters[0] == 16 && p.blocks_per_stage >= 1;
  bool c1_fused = false;
  int rc = CPX_OK;
  if (!c1_pending) rc = conv1();
  if (rc != CPX_OK) return rc;
  float* cur = act[0];
  int flip = 0, c_in = p.filters[0], hh = H, ww = W;
  for (int st = 0; st < 3; ++st) {
    const int f = p.filters[st + 1];
    for (int d = 0; d < p.blocks_per_stage; ++d) {
      const cpx_wrresnet_block& b = p.block[st][d];
      const int s = d == 0 ? st + 1 : 1;  // wr_block(stride = stage index), wr_resnet.py:27-30
      const int ho = (hh + s - 1) / s, wo = (ww + s - 1) / s;
      // fp16x2: where the first convolution's kernel can store fp16 planes and the second one's can stage them, `mid`
      // travels as the second convolution's scaled planes (same bytes as float32) and its staging is a copy
      bool planes_pair = false;
      if (h->cnn_math == CPX_CNN_MATH_FP16X2 && h->planes_handover) {
        cpx::ConvArgs pa{}, pb{};
        pa.N = N; pa.H = hh; pa.W = ww; pa.Ho = ho; pa.Wo = wo; pa.Cin = c_in; pa.Cout = f; pa.groups = p.groups; pa.ksize = 3;
        pa.stride = s; pa.pad_top = pa.pad_left = 1;
        pb = pa;
        pb.H = ho; pb.W = wo; pb.Cin = f; pb.stride = 1;
        planes_pair = c_in % p.groups == 0 && f % p.groups == 0 && cpx::conv_bf3_can_store_planes(pa) &&
                      cpx::conv_bf3_two_planes(pb) && cpx::conv_bf3_can_load_planes(pb);
      }
      hf.word = 2 + st * p.blocks_per_stage + d;
      // fp16x2: a block whose two convolutions are stride-1 with 32 channels per group (stage 2 past its first block) is
      // ONE launch -- `mid` stays in LDS (conv_block32_kernel); the two guarded three-plane launches follow as its rerun
      hf.rerun_only = false;
      // ... the stage's first block too (8 input channels per group; its 1x1 shortcut inside the second convolution)
      const bool first8 = d == 0 && s == 1 && c_in / p.groups == 8 && h->block_fusion >= 2 && h->fuse_shortcut;
      if (h->cnn_math == CPX_CNN_MATH_FP16X2 && h->block_fusion && s == 1 && ((d != 0 && c_in == f) || first8) && b.in_scale && cnn->split_of(b.wa) &&
          cnn->split_of(b.wb) && c_in % p.groups == 0 && f % p.groups == 0) {
        cpx::ConvArgs ca{}, cb{};
        ca.N = N; ca.H = hh; ca.W = ww; ca.Ho = hh; ca.Wo = ww; ca.Cin = f; ca.Cout = f; ca.groups = p.groups; ca.ksize = 3; ca.stride = 1;
        ca.relu = 1; ca.pad_top = ca.pad_left = 1; ca.planes = 2; ca.half = 1; ca.ovf = h->cnn_ovf + hf.word;
        cb = ca;
        ca.Cin = c_in;
        ca.in = cur; ca.out = mid; ca.weights = b.wa; ca.in_scale = b.in_scale; ca.in_shift = b.in_shift;
        ca.out_scale = b.a_scale; ca.out_shift = b.a_shift;
        ca.act_scale = cnn->act_scale[st][d][0]; ca.act_unscale = 1.0f / ca.act_scale;
        cb.in = mid; cb.out = act[flip ^ 1]; cb.weights = b.wb; cb.out_shift = b.bb;
        if (first8) {
          cb.sc_in = cur; cb.sc_w = p.shortcut_w[st]; cb.sc_bias = p.shortcut_b[st];
          cb.sc_H = hh; cb.sc_W = ww; cb.sc_cin = c_in; cb.sc_stride = 1;
        } else {
          cb.residual = cur;
        }
        cb.act_scale = cnn->act_scale[st][d][1]; cb.act_unscale = 1.0f / cb.act_scale;
        const bool c1_try = first8 && c1_pending && st == 0 && d == 0;
        if (c1_try) {
          ca.c1_in = in_dev; ca.c1_w = p.conv1_w; ca.c1_b = p.conv1_b;
        }
        if (cpx::conv_block32_supported(ca, cb)) {
          cpx_handle::ConvEv ev{};
          if (h->conv_timing) {
            // ("stride 4": a fused block; both convolutions' products -- and conv1_1's when it is computed inside --, the shortcut's not counted)
            ev.key = (c_in / p.groups) * 10000 + 32 * 10 + 4;
            ev.flops = 2.0 * N * hh * ww * f * ((double)(c_in / p.groups) + (double)(f / p.groups)) * 9;
            if (c1_try) ev.flops += 2.0 * N * hh * ww * c_in * (double)(p.in_channels / p.groups) * 9;
            if (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess)
              return fail(h, CPX_ERR_HIP, "cpx_cnn_forward: event creation failed");
            CPX_HIP(h, hipEventRecord(ev.e0, h->stream));
          }
          int rb = cpx::launch_conv_block32(ca, cb, cnn->split_of(b.wa), cnn->split_of(b.wb), h->stream);
          if (rb != 0 && c1_try) {  // not with conv1_1 inside: the layer as a launch of its own, then the block as before
            c1_pending = false;
            rc = conv1();
            if (rc != CPX_OK) return rc;
            ca.c1_in = ca.c1_w = ca.c1_b = nullptr;
            rb = cpx::launch_conv_block32(ca, cb, cnn->split_of(b.wa), cnn->split_of(b.wb), h->stream);
          }
          if (rb == 0) {
            hf.rerun_only = true;
            c1_fused = c1_try && c1_pending;
            if (h->conv_timing) {
              CPX_HIP(h, hipEventRecord(ev.e1, h->stream));
              h->conv_events.push_back(ev);
            }
          } else {
            if (h->conv_timing) { hipEventDestroy(ev.e0); hipEventDestroy(ev.e1); }
            if (rb != -2 && rb != -3) return fail(h, CPX_ERR_HIP, "cpx_cnn_forward: block kernel configuration failed");
          }
        }
      }
      if (c1_pending) {  // conv1_1: in front of a first block that did not take it, or guarded, as the head of that block's rerun
        c1_pending = false;
        const bool was = hf.rerun_only;
        hf.rerun_only = c1_fused;
        rc = conv1();
        hf.rerun_only = was;
        if (rc != CPX_OK) return rc;
      }
      if (hf.rerun_only) planes_pair = false;  // (the rerun hands float32 over)
      hf.act_scale = cnn->act_scale[st][d][0];
      hf.out_planes = planes_pair;
      hf.out_act_scale = cnn->act_scale[st][d][1];
      hf.in_planes = false;
      rc = conv(cur, mid, b.wa, hh, ww, c_in, f, 3, s, 1, 1, b.in_scale, b.in_shift, b.a_scale, b.a_shift, nullptr);
      if (rc != CPX_OK) return rc;
      hf.act_scale = cnn->act_scale[st][d][1];
      hf.out_planes = false;
      hf.in_planes = planes_pair;
      const float* res = cur;
      bool fused = false;
      if (d == 0) {
        // the 1x1 shortcut of a stage's first block: folded into the block's second convolution when that one runs
        // on the split-operand kernel (saves writing and re-reading the shortcut tensor), a launch of its own otherwise
        cpx_conv_desc probe{};
        probe.Cin = f; probe.Cout = f; probe.groups = p.groups; probe.ksize = 3; probe.stride = 1;
        // (the kernels' fused shortcut walks K in fours -- conv_bf3w_kernel -- or in twos: a block input with 2, 6, 10 ...
        // channels per group keeps the shortcut as a launch of its own rather than depending on which kernel takes the layer)
        fused = h->fuse_shortcut && conv_can_fuse(h, &probe) && (c_in / p.groups) % 4 == 0;
        if (!fused) {
          rc = conv(cur, sc, p.shortcut_w[st], hh, ww, c_in, f, 1, s, 0, 0, nullptr, nullptr, nullptr, p.shortcut_b[st],
                    nullptr);
          if (rc != CPX_OK) return rc;
          res = sc;
        }
      }
      flip ^= 1;
      if (fused) {
        conv_fuse fu;
        fu.in = cur; fu.w = p.shortcut_w[st]; fu.bias = p.shortcut_b[st];
        fu.H = hh; fu.W = ww; fu.cin = c_in; fu.stride = s;
        cpx_conv_desc dd{};
        dd.N = N; dd.H = ho; dd.W = wo; dd.Cin = f; dd.Cout = f; dd.groups = p.groups; dd.ksize = 3; dd.stride = 1;
        dd.pad_same = 1; dd.relu = 1;
        dd.in_dev = mid; dd.out_dev = act[flip]; dd.weights_dev = b.wb; dd.out_shift_dev = b.bb;
        rc = conv_run(h, &dd, cnn->split_of(b.wb), &fu, &hf);
      } else {
        rc = conv(mid, act[flip], b.wb, ho, wo, f, f, 3, 1, 1, 1, nullptr, nullptr, nullptr, b.bb, res);
      }
      if (rc != CPX_OK) return rc;
      cur = act[flip];
      hh = ho;
      ww = wo;
      c_in = f;
    }
  }
  cpx_head_desc hd{};
  hd.N = N; hd.HW = hh * ww; hd.C = c_in; hd.L = p.n_labels;
  hd.n_hidden = p.n_hidden;
  hd.activation = p.activation;
  for (int k = 0; k < p.n_hidden; ++k) {
    hd.hidden_sizes[k] = p.hidden_sizes[k];
    hd.hidden_w_dev[k] = p.hidden_w[k];
    hd.hidden_b_dev[k] = p.hidden_b[k];
  }
  hd.in_dev = cur; hd.bn_scale_dev = p.final_scale; hd.bn_shift_dev = p.final_shift;
  hd.dense_w_dev = p.dense_w; hd.dense_b_dev = p.dense_b; hd.logits_dev = logits_dev; hd.probs_dev = probs_dev;
  rc = cpx_cnn_head_ex(h, &hd);
  if (rc == CPX_OK && h->cnn_math == CPX_CNN_MATH_FP16X2)
    cpx::launch_count_overflow(h->cnn_ovf, 3 * p.blocks_per_stage, h->stream);
  if (rc == CPX_OK && h->cnn_math == CPX_CNN_MATH_FP16X2 && std::getenv("CPX_CNN_DEBUG_OVF")) {
    // diagnostic (synchronises): which blocks of this forward left fp16's range
    int words[OVF_WORDS];
    if (hipStreamSynchronize(h->stream) == hipSuccess &&
        hipMemcpy(words, h->cnn_ovf, sizeof(words), hipMemcpyDeviceToHost) == hipSuccess && words[0]) {
      std::fprintf(stderr, "cpx_cnn_forward: N = %d, fp16 overflow in blocks", N);
      for (int k = 0; k < 3 * p.blocks_per_stage; ++k)
        if (words[2 + k]) std::fprintf(stderr, " %d.%d", k / p.blocks_per_stage + 2, k % p.blocks_per_stage);
      std::fprintf(stderr, "\n");
    }
  }
  return rc;
}

int cpx_ir_delta_variance(cpx_handle* h, const uint8_t* cur_dev, const uint8_t* prev_dev, int width, int height,
                          const int32_t* rects_dev, int n, double* var_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!cur_dev || !prev_dev || width < 1 || height < 1 || n < 0 || (n > 0 && (!rects_dev || !var_dev)))
    return fail(h, CPX_ERR_INVALID, "cpx_ir_delta_variance: bad argument");
  if (n == 0) return CPX_OK;
  CPX_ENTER(h);
  cpx::IrVarArgs a{};
  a.W = width; a.H = height; a.n = n;
  a.cur = cur_dev; a.prev = prev_dev; a.rects = rects_dev; a.out = var_dev;
  cpx::launch_ir_delta_variance(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_ir_resize_area(cpx_handle* h, const uint8_t* src_dev, int n, int width, int height, int factor, uint8_t* dst_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!src_dev || !dst_dev || n < 0 || width < 1 || height < 1 || factor < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_ir_resize_area: bad argument");
  if (factor > 16 || width % factor || height % factor)
    return fail(h, CPX_ERR_UNSUPPORTED, "cpx_ir_resize_area: the factor must divide both sides (integer-ratio INTER_AREA only)");
  if (n == 0) return CPX_OK;
  CPX_ENTER(h);
  cpx::launch_ir_resize_area(src_dev, dst_dev, n, width, height, factor, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_ir_merge(cpx_handle* h, const cpx_component* comps_dev, const int32_t* counts_dev, int n, int cap_in, int cap_out,
                 const uint8_t* cur_dev, const uint8_t* prev_dev, int width, int height, int frame_number, int out_stride,
                 cpx_component* out_comps_dev, cpx_frame_info* out_info_dev, int32_t* status_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!comps_dev || !counts_dev || !cur_dev || !out_comps_dev || !status_dev || n < 0 || cap_in < 1 || cap_out < 1 ||
      cap_out > 1024 || width < 1 || height < 1 || frame_number < 0 || out_stride < 1 || frame_number >= out_stride)
    return fail(h, CPX_ERR_INVALID, "cpx_ir_merge: bad argument");
  if (n == 0) return CPX_OK;
  CPX_ENTER(h);
  cpx::IrMergeArgs a{};
  a.W = width; a.H = height; a.n = n; a.cap_in = cap_in; a.cap_out = cap_out;
  a.frame_number = frame_number; a.out_stride = out_stride;
  a.comps = comps_dev; a.counts = counts_dev; a.cur = cur_dev; a.prev = prev_dev;
  a.out_comps = out_comps_dev; a.out_info = out_info_dev; a.status = status_dev;
  cpx::launch_ir_merge(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_ir_frame_statistics(cpx_handle* h, const uint8_t* frames_dev, const uint8_t* masks_dev, int n, int pixels,
                            uint32_t* hist_dev, cpx_ir_frame_stats* out_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!frames_dev || !hist_dev || !out_dev || n < 0 || pixels < 1)
    return fail(h, CPX_ERR_INVALID, "cpx_ir_frame_statistics: bad argument");
  if (n == 0) return CPX_OK;
  CPX_ENTER(h);
  CPX_HIP(h, hipMemsetAsync(hist_dev, 0, (size_t)n * 256 * sizeof(uint32_t), h->stream));
  CPX_HIP(h, hipMemsetAsync(out_dev, 0, (size_t)n * sizeof(cpx_ir_frame_stats), h->stream));
  cpx::IrStatsArgs a{};
  a.n = n; a.pixels = pixels;
  a.vec16 = pixels % 16 == 0 && reinterpret_cast<uintptr_t>(frames_dev) % 16 == 0 &&
            (!masks_dev || reinterpret_cast<uintptr_t>(masks_dev) % 16 == 0);
  a.frames = frames_dev; a.masks = masks_dev; a.hist = hist_dev; a.out = out_dev;
  cpx::launch_ir_frame_stats(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

// ---- IR background model ---------------------------------------------------------------------------------------------
struct cpx_mog2 {
  cpx_handle* h = nullptr;
  int n_streams = 0, width = 0, height = 0, history = 0, nframes = 0;
  float var_threshold = 16.0f;
  size_t n = 0;
  float* state = nullptr;        // weight | var | mean, each [5][n]
  unsigned char* modes = nullptr;
};

static void mog2_free(cpx_mog2* m) {
  if (m->state) hipFree(m->state);
  if (m->modes) hipFree(m->modes);
  delete m;
}

int cpx_mog2_create(cpx_handle* h, int n_streams, int width, int height, int history, float var_threshold,
                    cpx_mog2** out) {
  if (!h) return CPX_ERR_INVALID;
  if (!out || n_streams < 1 || width < 1 || height < 1 || !(var_threshold > 0.0f))
    return fail(h, CPX_ERR_INVALID, "cpx_mog2_create: bad argument");
  *out = nullptr;
  CPX_ENTER(h);
  cpx_mog2* m = new (std::nothrow) cpx_mog2();
  if (!m) return fail(h, CPX_ERR_NOMEM, "cpx_mog2_create: out of memory");
  m->h = h;
  m->n_streams = n_streams;
  m->width = width;
  m->height = height;
  m->history = history > 0 ? history : 500;
  m->var_threshold = var_threshold;
  m->n = (size_t)n_streams * width * height;
  if (hipMalloc(reinterpret_cast<void**>(&m->state), 15 * m->n * sizeof(float)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&m->modes), m->n) != hipSuccess) {
    (void)hipGetLastError();
    mog2_free(m);
    return fail(h, CPX_ERR_NOMEM, "cpx_mog2_create: state allocation failed");
  }
  CPX_HIP(h, hipMemsetAsync(m->state, 0, 15 * m->n * sizeof(float), h->stream));
  CPX_HIP(h, hipMemsetAsync(m->modes, 0, m->n, h->stream));
  h->mog2s.push_back(m);
  *out = m;
  return CPX_OK;
}

void cpx_mog2_destroy(cpx_mog2* m) {
  if (!m) return;
  cpx_handle* h = m->h;
  hipSetDevice(h->device);
  hipStreamSynchronize(h->stream);
  h->mog2s.erase(std::remove(h->mog2s.begin(), h->mog2s.end(), m), h->mog2s.end());
  mog2_free(m);
}

static cpx::Mog2Args mog2_args(const cpx_mog2* m) {
  cpx::Mog2Args a{};
  a.n = m->n;
  a.var_threshold = m->var_threshold;
  a.background_ratio = 0.9f;
  a.var_threshold_gen = 9.0f;
  a.var_init = 15.0f;
  a.var_min = 4.0f;
  a.var_max = 75.0f;
  a.weight = m->state;
  a.var = m->state + 5 * m->n;
  a.mean = m->state + 10 * m->n;
  a.modes = m->modes;
  return a;
}

int cpx_mog2_apply(cpx_mog2* m, const uint8_t* frames_dev, double learning_rate, uint8_t* fgmask_dev) {
  if (!m) return CPX_ERR_INVALID;
  cpx_handle* h = m->h;
  if (!frames_dev || !fgmask_dev) return fail(h, CPX_ERR_INVALID, "cpx_mog2_apply: null argument");
  CPX_ENTER(h);
  m->nframes += 1;
  const double rate = (learning_rate >= 0 && m->nframes > 1) ? learning_rate
                                                            : 1.0 / std::min(2 * m->nframes, m->history);
  cpx::Mog2Args a = mog2_args(m);
  a.alphaT = (float)rate;
  a.alpha1 = 1.0f - a.alphaT;
  a.prune = (float)(-rate * 0.05f);  // -learningRate * fCT, fCT a float member as in the reference implementation
  a.frames = frames_dev;
  a.mask = fgmask_dev;
  cpx::launch_mog2_apply(a, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_mog2_background(cpx_mog2* m, uint8_t* out_dev) {
  if (!m) return CPX_ERR_INVALID;
  cpx_handle* h = m->h;
  if (!out_dev) return fail(h, CPX_ERR_INVALID, "cpx_mog2_background: null argument");
  CPX_ENTER(h);
  cpx::launch_mog2_background(mog2_args(m), out_dev, h->stream);
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_ir_detect(cpx_handle* h, const uint8_t* images_dev, int n_frames, int width, int height, int threshold,
                  int max_components, cpx_component* comps_dev, int32_t* counts_dev, int32_t* status_dev,
                  int32_t* labels_dev) {
  if (!h) return CPX_ERR_INVALID;
  if (!images_dev || !comps_dev || !counts_dev || !status_dev || n_frames < 1 || max_components < 1 || threshold < 0 ||
      threshold > 255)
    return fail(h, CPX_ERR_INVALID, "cpx_ir_detect: bad argument");
  if (!cpx::ir_supported(width, height))
    return fail(h, CPX_ERR_UNSUPPORTED, "cpx_ir_detect: width must be a multiple of 64 and width x height at most 640 x 480");
  CPX_ENTER(h);
  cpx::IrArgs a{};
  a.W = width;
  a.H = height;
  a.threshold = threshold;
  a.max_components = max_components;
  a.images = images_dev;
  a.comps = comps_dev;
  a.counts = counts_dev;
  a.status = status_dev;
  a.labels = labels_dev;
  // one slot per frame that can be resident at once (at most one workgroup of this LDS size per CU pair)
  a.n_slots = n_frames < 256 ? n_frames : 256;
  a.slot_bytes = cpx::ir_slot_bytes(width, height);
  const size_t need = a.slot_bytes * (size_t)a.n_slots;
  if (need > h->ir_scratch_bytes) {
    CPX_HIP(h, hipStreamSynchronize(h->stream));
    if (h->ir_scratch) hipFree(h->ir_scratch);
    h->ir_scratch = nullptr;
    h->ir_scratch_bytes = 0;
    if (hipMalloc(reinterpret_cast<void**>(&h->ir_scratch), need) != hipSuccess) {
      (void)hipGetLastError();
      return fail(h, CPX_ERR_NOMEM, "cpx_ir_detect: scratch allocation failed");
    }
    h->ir_scratch_bytes = need;
  }
  if (!h->ir_bitmap && hipMalloc(reinterpret_cast<void**>(&h->ir_bitmap), 32) != hipSuccess) {
    (void)hipGetLastError();
    return fail(h, CPX_ERR_NOMEM, "cpx_ir_detect: scratch allocation failed");
  }
  CPX_HIP(h, hipMemsetAsync(h->ir_bitmap, 0, 32, h->stream));
  a.slots = h->ir_scratch;
  a.slot_bitmap = h->ir_bitmap;
  if (cpx::launch_ir_detect(a, n_frames, h->stream) != 0)
    return fail(h, CPX_ERR_HIP, "cpx_ir_detect: kernel configuration failed");
  CPX_HIP(h, hipGetLastError());
  return CPX_OK;
}

int cpx_last_kernel_timing(cpx_handle* h, float* total_ms, int* launches) {
  if (!h || !total_ms || !launches) return CPX_ERR_INVALID;
  if (!h->timing_valid) return fail(h, CPX_ERR_INVALID, "no batch has been run");
  CPX_HIP(h, hipEventSynchronize(h->ev1));
  CPX_HIP(h, hipEventElapsedTime(total_ms, h->ev0, h->ev1));
  *launches = h->last_launches;
  return CPX_OK;
}

}  // extern "C"
