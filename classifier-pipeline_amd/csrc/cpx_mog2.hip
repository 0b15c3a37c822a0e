// cpx_mog2.hip -- background model of the IR tracker: the per-pixel Gaussian-mixture update of
// cv2.createBackgroundSubtractorMOG2(history, varThreshold, detectShadows=False).apply(frame, None, learning_rate) and
// getBackgroundImage(), as the reference's CVBackground drives them (track/cliptracker.py:561-613).  The algorithm is
// OpenCV's (modules/video/src/bgfg_gaussmix2.cpp; Zivkovic 2004 / 2006), which is not vendored in the reference:
// restated from the published update, parity against cv2 itself UNPINNED (oracle/mog2_oracle.c says the same).
//
// One thread per pixel of every stream; the mixture (5 modes x weight / variance / mean, float32) lives in HBM as
// mode-major planes so that every load and store of a wave is one contiguous 256 bytes: 61 B of state per pixel are
// read and written per frame next to the 1 B of input and 1 B of mask -- HBM-bound, 124 algorithmic bytes per pixel.
// The five modes sit in registers; the data-dependent loops of the reference (bubble the matched mode up, prune,
// insert a new mode) are unrolled with compile-time indices and predicated, in the reference's operation order, float
// by float (-ffp-contract=off).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "cpx_kernels.h"

namespace cpx {

namespace {
constexpr int K = 5;

__device__ __forceinline__ void swap3(float& w0, float& v0, float& m0, float& w1, float& v1, float& m1) {
  float t;
  t = w0; w0 = w1; w1 = t;
  t = v0; v0 = v1; v1 = t;
  t = m0; m0 = m1; m1 = t;
}
}  // namespace

__global__ __launch_bounds__(256) void cpx_mog2_apply_kernel(Mog2Args a) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float wgt[K], var[K], mean[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    wgt[k] = a.weight[(size_t)k * a.n + i];
    var[k] = a.var[(size_t)k * a.n + i];
    mean[k] = a.mean[(size_t)k * a.n + i];
  }
  const float data = (float)a.frames[i];
  const float alphaT = a.alphaT, alpha1 = a.alpha1, prune = a.prune;
  bool background = false, fits = false;
  int nmodes = a.modes[i];
  float total = 0.0f;
#pragma unroll
  for (int mode = 0; mode < K; ++mode) {
    if (mode < nmodes) {  // nmodes shrinks while modes are pruned, exactly as the reference's loop condition sees it
      float weight = alpha1 * wgt[mode] + prune;
      int swap_count = 0;
      if (!fits) {
        const float v = var[mode];
        const float d = mean[mode] - data;
        const float dist2 = d * d;
        if (total < a.background_ratio && dist2 < a.var_threshold * v) background = true;
        if (dist2 < a.var_threshold_gen * v) {
          fits = true;
          weight += alphaT;
          const float k = alphaT / weight;
          mean[mode] -= k * d;
          float varnew = v + k * (dist2 - v);
          varnew = varnew > a.var_min ? varnew : a.var_min;
          varnew = varnew < a.var_max ? varnew : a.var_max;
          var[mode] = varnew;
          bool moving = true;  // the matched mode moves up past lighter ones
#pragma unroll
          for (int j = mode; j > 0; --j) {
            if (moving && !(weight < wgt[j - 1])) {
              ++swap_count;
              swap3(wgt[j], var[j], mean[j], wgt[j - 1], var[j - 1], mean[j - 1]);
            } else {
              moving = false;
            }
          }
        }
      }
      if (weight < -prune) {
        weight = 0.0f;
        --nmodes;
      }
#pragma unroll
      for (int j = 0; j <= mode; ++j)
        if (j == mode - swap_count) wgt[j] = weight;
      total += weight;
    }
  }
  float inv = 0.0f;
  if (fabsf(total) > FLT_EPSILON) inv = 1.0f / total;
#pragma unroll
  for (int mode = 0; mode < K; ++mode)
    if (mode < nmodes) wgt[mode] *= inv;
  if (!fits && alphaT > 0.0f) {
    const int slot = nmodes == K ? K - 1 : nmodes++;  // replace the weakest or append
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (j == slot) {
        wgt[j] = nmodes == 1 ? 1.0f : alphaT;
        mean[j] = data;
        var[j] = a.var_init;
      } else if (nmodes != 1 && j < nmodes - 1) {
        wgt[j] *= alpha1;
      }
    }
    bool moving = true;
#pragma unroll
    for (int j = K - 1; j > 0; --j) {
      if (j <= nmodes - 1) {
        if (moving && !(alphaT < wgt[j - 1])) {
          swap3(wgt[j], var[j], mean[j], wgt[j - 1], var[j - 1], mean[j - 1]);
        } else {
          moving = false;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    a.weight[(size_t)k * a.n + i] = wgt[k];
    a.var[(size_t)k * a.n + i] = var[k];
    a.mean[(size_t)k * a.n + i] = mean[k];
  }
  a.modes[i] = (unsigned char)nmodes;
  a.mask[i] = background ? 0 : 255;
}

// getBackgroundImage: weighted mean of the heaviest modes up to the background ratio, round-to-nearest-even, saturated
__global__ __launch_bounds__(256) void cpx_mog2_background_kernel(Mog2Args a, unsigned char* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int nmodes = a.modes[i];
  float mean_val = 0.0f, total = 0.0f;
  bool open = true;
#pragma unroll
  for (int mode = 0; mode < K; ++mode) {
    if (open && mode < nmodes) {
      const float w = a.weight[(size_t)mode * a.n + i];
      mean_val += w * a.mean[(size_t)mode * a.n + i];
      total += w;
      if (total > a.background_ratio) open = false;
    }
  }
  float inv = 0.0f;
  if (fabsf(total) > FLT_EPSILON) inv = 1.0f / total;
  mean_val *= inv;
  const float r = rintf(mean_val);
  out[i] = (unsigned char)(r < 0.0f ? 0.0f : r > 255.0f ? 255.0f : r);
}

void launch_mog2_apply(const Mog2Args& a, hipStream_t s) {
  hipLaunchKernelGGL(cpx_mog2_apply_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
}
void launch_mog2_background(const Mog2Args& a, unsigned char* out, hipStream_t s) {
  hipLaunchKernelGGL(cpx_mog2_background_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a, out);
}

}  // namespace cpx
