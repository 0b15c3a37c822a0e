// cpx_assoc.hip -- association stage: one GPU lane per clip walks the clip's
// processed frames in order (the work is scalar and sequential inside a clip,
// embarrassingly parallel across clips).  The arithmetic lives in
// cpx_assoc_core.h.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "cpx_assoc_core.h"
#include "cpx_final_core.h"
#include "cpx_kernels.h"

namespace cpx {

__global__ __launch_bounds__(64) void cpx_assoc_kernel(AssocArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int pbase = a.proc_off[b];
  const int nproc = a.proc_off[b + 1] - pbase;
  const int first = a.clip_first[b];
  AssocClip c;
  c.p = &a.params;
  c.cap = a.cap;
  c.max_active = a.params.max_active_tracks;
  c.max_tracks = a.params.max_tracks;
  c.pool = a.pool + (size_t)first * c.max_active;
  c.active = a.active + (size_t)b * c.max_active;
  c.n_active = 0;
  c.tracks = a.tracks + (size_t)b * c.max_tracks;
  c.n_tracks = 0;
  c.next_id = 1;  // Track._track_id is reset per Clip (clip.py:57-59)
  c.regs = a.regs + (size_t)b * a.cap;
  c.scores = a.scores + (size_t)b * a.cap * c.max_active;
  c.used = a.used + (size_t)b * a.cap;
  c.status = 0;
  if (!a.fresh) {  // incremental call: the clip's association state comes from the previous call
    const AssocResume r = a.resume[b];
    c.n_active = r.n_active;
    c.n_tracks = r.n_tracks;
    c.next_id = r.next_id;
    c.status = r.status;
  }
  for (int t = a.t_begin; t < nproc; ++t) {
    const int fidx = a.proc_idx[pbase + t];
    const cpx_frame_info& fi = a.info[fidx];
    int nreg = 0;
    if (a.proc_ffc[pbase + t]) {
      c.n_active = 0;  // cliptrackextractor.py:239-240
    } else {
      if (fi.status != 0) c.status = fi.status;
      const int ncomp = fi.status == 0 ? fi.n_components : 0;
      nreg = build_regions(c, a.comps + (size_t)fidx * a.cap, ncomp, t, t > 0);
      if (a.regions_out)
        for (int i = 0; i < nreg; ++i) a.regions_out[(size_t)fidx * a.cap + i] = c.regs[i];
      assoc_frame(c, nreg, t);
    }
    if (a.region_counts) a.region_counts[fidx] = nreg;
  }
  // a clip that ran out of table space (components of a frame, simultaneous or total tracks) has no usable track list:
  // it reports none, so that the stages behind spend nothing on it, and its status says why -- the caller runs it again
  // on larger tables (cpx.engine.TrackEngine.track_clip_grown)
  a.n_tracks[b] = c.status != 0 ? 0 : c.n_tracks;
  a.status[b] = c.status;
  AssocResume r;
  r.n_active = c.n_active;
  r.n_tracks = c.n_tracks;
  r.next_id = c.next_id;
  r.status = c.status;
  a.resume[b] = r;
}

// end of clip: trim / statistics / rejects / score order + the size of the classification plan
__global__ __launch_bounds__(64) void cpx_finalize_kernel(FinalArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int pbase = a.proc_off[b];
  const int first = a.clip_first[b];
  const int ma = a.params.max_active_tracks, mt = a.params.max_tracks_per_clip;
  FinalScratch sc;
  sc.d = a.scratch_d + (size_t)b * 2 * a.max_frames;
  sc.f = a.scratch_f + (size_t)b * a.max_frames;
  finalize_clip(a.params, a.pool + (size_t)first * ma, a.tracks + (size_t)b * mt, a.n_tracks[b], a.proc_ffc + pbase,
                a.square_width, a.summaries + (size_t)b * mt, a.counts + 4 * b, sc);
}

// exclusive prefix sums of the per-clip work counts [B][4] (kept tracks, region refs, samples, spare) -> [B + 1][4],
// row B = the totals: one workgroup, a wave scan per 1024 clips with the carry handed on -- what sizes and addresses
// the plan pass, on the handle's stream (it was a torch cumsum + subtract + cast: three at::native launches per step)
__global__ __launch_bounds__(1024) void cpx_counts_prefix_kernel(const int* __restrict__ counts, int B, int* __restrict__ prefix) {
  __shared__ int s_wave[16][4];
  __shared__ int s_carry[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 4) s_carry[tid] = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int b = base + tid;
    int v[4], inc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = b < B ? counts[4 * b + k] : 0;
      inc[k] = v[k];
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc[k], d, 64);
        if (lane >= d) inc[k] += o;
      }
      if (lane == 63) s_wave[wave][k] = inc[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int before = s_carry[k];
      for (int w = 0; w < wave; ++w) before += s_wave[w][k];
      if (b < B) prefix[4 * b + k] = before + inc[k] - v[k];
    }
    __syncthreads();
    if (tid < 4) {
      int tot = s_carry[tid];
      for (int w = 0; w < 16; ++w) tot += s_wave[w][tid];
      s_carry[tid] = tot;
    }
    __syncthreads();
  }
  if (tid < 4) prefix[4 * B + tid] = s_carry[tid];
}

__global__ __launch_bounds__(64) void cpx_plan_kernel(FinalArgs a) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int pbase = a.proc_off[b];
  const int first = a.clip_first[b];
  const int ma = a.params.max_active_tracks, mt = a.params.max_tracks_per_clip;
  plan_clip(a.params, a.pool + (size_t)first * ma, a.summaries + (size_t)b * mt, a.n_tracks[b], a.proc_ffc + pbase,
            a.proc_idx + pbase, a.square_width, b, a.prefix + 4 * b, a.refs, a.track_offsets, a.reqs, a.sample_track,
            a.track_clip, reinterpret_cast<unsigned char*>(a.scratch_d + (size_t)b * 2 * a.max_frames), b == a.B - 1);
}

// One lane walks one clip, and the clips of a wave take different paths: a wave costs the union of its lanes' paths,
// and a lone wave per CU hides no latency.  So few clips share a wave (two at 4096 clips: 2048 waves, two per SIMD)
// instead of sixty-four (64 waves on a 1024-SIMD chip): assoc / finalize / plan 5.3 / 5.5 / 3.8 -> 4.1 / 3.3 / 2.4 ms
// per 4096 clips x 270 frames (what is left is one clip's own serial walk).
static inline int lanes_per_wave(int B) {
  int l = B / 2048;
  return l < 1 ? 1 : (l > 64 ? 64 : l);
}

void launch_finalize(const FinalArgs& a, hipStream_t s) {
  const int l = lanes_per_wave(a.B);
  hipLaunchKernelGGL(cpx_finalize_kernel, dim3((a.B + l - 1) / l), dim3(l), 0, s, a);
}
void launch_counts_prefix(const int* counts, int B, int* prefix, hipStream_t s) {
  hipLaunchKernelGGL(cpx_counts_prefix_kernel, dim3(1), dim3(1024), 0, s, counts, B, prefix);
}
void launch_plan(const FinalArgs& a, hipStream_t s) {
  const int l = lanes_per_wave(a.B);
  hipLaunchKernelGGL(cpx_plan_kernel, dim3((a.B + l - 1) / l), dim3(l), 0, s, a);
}

size_t assoc_active_bytes() { return sizeof(ActiveTrack); }
size_t assoc_score_bytes() { return sizeof(ScoreRec); }

void launch_assoc(const AssocArgs& a, hipStream_t s) {
  const int l = lanes_per_wave(a.B);
  hipLaunchKernelGGL(cpx_assoc_kernel, dim3((a.B + l - 1) / l), dim3(l), 0, s, a);
}

}  // namespace cpx
