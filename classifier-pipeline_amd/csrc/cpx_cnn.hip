// cpx_cnn.hip -- WR-ResNet forward on gfx950 matrix cores (reference architecture:
// ml_tools/resnet/wr_resnet.py:5-98; head ml_tools/kerasmodel.py:308-350).
//
// Grouped KxK convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32 in /
// f32 accumulate: the logits must match the exported TF model to 1e-3):
//   M = output pixels (one 8x16 tile per workgroup, 32 pixels per wave)
//   N = output channels of one group (32 per MFMA tile, NTN tiles per wave)
//   K = KS*KS*Cin_g, walked as (cin chunk of KC) x (tap) x (pair of channels)
// Activations are NHWC f32.  The pre-activation BatchNorm + ReLU of a block is applied
// while the input patch is staged into LDS (padding stays exactly 0 afterwards, as in
// TF); bias, the following BatchNorm (folded on the host into scale / shift), the
// residual add and ReLU are applied from the accumulators before the single store.
#include <hip/hip_runtime.h>

#include "cpx_kernels.h"

// input channels staged per chunk for the stride-1 3x3 layers of stage 2 / 3 / 4.  Small chunks keep the
// workgroup's LDS at 25-30 KB so that 4 workgroups share a CU and hide each other's staging / epilogue
// latency: measured 63 (32/32/16) -> 79 (32/16/8) -> 92 TFLOP/s (16/8/4) for the whole forward; with the
// register prefetch of the next chunk (below) and two bands per workgroup in stage 2: 100 TFLOP/s (8/8/4)
#ifndef CPX_CONV_KC_S2
#define CPX_CONV_KC_S2 8
#endif
#ifndef CPX_CONV_KC_S3
#define CPX_CONV_KC_S3 8
#endif
#ifndef CPX_CONV_KC_S4
#define CPX_CONV_KC_S4 4
#endif
// 8-row bands of output pixels per workgroup (M tiles per wave) for the same layers: with 2 bands a weight
// fragment read from LDS feeds two MFMAs, the weight chunk is staged once per 256 pixels and the halo shrinks
#ifndef CPX_CONV_NTM_S2
#define CPX_CONV_NTM_S2 2
#endif
#ifndef CPX_CONV_NTM_S3
#define CPX_CONV_NTM_S3 1
#endif
#ifndef CPX_CONV_NTM_S4
#define CPX_CONV_NTM_S4 1
#endif
#ifndef CPX_CONV_TW_S4
#define CPX_CONV_TW_S4 32  // band width of the layers that produce the 27 x 27 maps
#endif

namespace cpx {

namespace {

// uniform base + 32-bit unsigned byte offset: `global_load v, v_off, s[base]` instead of 64-bit address arithmetic per
// access (everything addressed this way stays inside one sample or one weight block: < 2^32 bytes)
template <typename T>
__device__ __forceinline__ const T* at_off(const T* base, unsigned bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + bytes);
}
// ReLU of a float as an integer maximum: one instruction (fmaxf(x, 0) on a value of unknown origin is canonicalised first)
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
// a native vector, not HIP's float4 struct: struct copies become memcpy calls that keep a staging array in scratch
typedef float f32x4 __attribute__((ext_vector_type(4)));

// a band of output pixels is 128 pixels = 4 waves x 32: TW columns x TB = 128 / TW rows, wave w owns the rows
// [w * 32 / TW, (w + 1) * 32 / TW) of every band.  TW = 16 (8 x 16 bands) everywhere except the 27 x 27 maps of
// stage 4, where 4 x 32 bands waste 19 % of the tile pixels instead of 29 %
constexpr int CT = 256;

template <int KC, int NTN, int S, int KS, int NTM, int TW>
__global__ __launch_bounds__(CT) void conv_mfma_kernel(ConvArgs a) {
  // (the guarded rerun of a fp16x2 layer that has no split-operand kernel -- the stride-3 convolution: nothing to do unless the
  // layer left fp16's range)
  if (a.guard != nullptr && *a.guard == 0) return;
  constexpr int TB = 128 / TW;        // rows of a band
  constexpr int WR = 32 / TW;         // rows of a wave's 32-pixel tile
  constexpr int TH = TB * NTM;        // output tile of a workgroup: NTM bands
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr int KP = KC + 1;          // padded channel stride of a patch pixel (bank spread)
  constexpr int COG = 32 * NTN;       // output channels per group
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* s_patch = lds;                      // [PH*PW][KP]
  constexpr int PATCH_FLOATS = (PH * PW * KP + 3) / 4 * 4;
  float* s_w = lds + PATCH_FLOATS;           // [KS*KS][KC][COG], 16-byte aligned

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.Wo + TW - 1) / TW, tiles_y = (a.Ho + TH - 1) / TH;
  int bid = blockIdx.x;
#ifndef CPX_CONV_NO_XCD_REMAP
  // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): give every XCD a contiguous run of
  // tiles so that neighbouring tiles, which share their halo rows / columns, hit the same L2
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
#endif
  const int txi = bid % tiles_x;
  bid /= tiles_x;
  const int tyi = bid % tiles_y;
  const int n = bid / tiles_y;
  const int g = blockIdx.y;
  const int cin_g = a.Cin / a.groups;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int iy0 = oy0 * S - a.pad_top, ix0 = ox0 * S - a.pad_left;
  const float* in_n = a.in + (size_t)n * a.H * a.W * a.Cin + (size_t)g * cin_g;
  const float* wg = a.weights + (size_t)g * KS * KS * cin_g * COG;

  f32x16 acc[NTM][NTN];
#pragma unroll
  for (int m = 0; m < NTM; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

  // this lane's pixel inside the wave tile and its k half
  const int pi = lane & 31, kh = lane >> 5;
  const int prow = WR * wave + pi / TW, pcol = pi % TW;
  const int a_base = ((prow * S) * PW + pcol * S) * KP + kh;
  const int b_base = kh * COG + (lane & 31);

  // staging geometry: a thread always serves the same group of 4 channels (256 % (KC/4) == 0), so the
  // BatchNorm scale / shift of the prologue sit in registers for a whole chunk; 16-byte global loads.
  // The loads of chunk c+1 are issued before the MFMA loop of chunk c and land in registers while the
  // matrix cores work; they are written to LDS (BN + ReLU applied, padding kept exactly 0) after the loop.
  constexpr int C4 = KC / 4;                 // float4 groups per pixel
  constexpr int NITEM = PH * PW * C4;        // float4 items of a patch chunk
  constexpr int NP = (NITEM + CT - 1) / CT;  // ... per thread
  constexpr int WV = KC * COG / 4;           // float4 per tap of a weight chunk [tap][KC][COG]
  constexpr int NWI = (KS * KS * WV + CT - 1) / CT;
  const int my_c4 = tid % C4;
  f32x4 pre_p[NP], pre_w[NWI];
  float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
  // iteration cc = -KC only fetches chunk 0; iteration cc >= 0 commits chunk cc, fetches cc + KC, multiplies cc
  for (int cc = -KC; cc < cin_g; cc += KC) {
    if (cc >= 0) {
      // ---- registers -> LDS: patch chunk (BN + ReLU prologue; zero padding stays 0) and weight chunk ----
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int item = tid + i * CT;
        if (item < NITEM) {
          const int px = item / C4;
          const int py = px / PW, pxx = px - py * PW;
          const int iy = iy0 + py, ix = ix0 + pxx;
          f32x4 v = pre_p[i];
          const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          if (a.in_scale) {
            v.x = fmaxf(v.x * psc.x + psh.x, 0.0f);
            v.y = fmaxf(v.y * psc.y + psh.y, 0.0f);
            v.z = fmaxf(v.z * psc.z + psh.z, 0.0f);
            v.w = fmaxf(v.w * psc.w + psh.w, 0.0f);
          }
          v.x = inside ? v.x : 0.0f;  // padding is zero after the prologue, as TensorFlow pads the activated tensor
          v.y = inside ? v.y : 0.0f;
          v.z = inside ? v.z : 0.0f;
          v.w = inside ? v.w : 0.0f;
          float* d = s_patch + px * KP + 4 * my_c4;
          d[0] = v.x;
          d[1] = v.y;
          d[2] = v.z;
          d[3] = v.w;
        }
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = tid + i * CT;
        if (item < KS * KS * WV) {
          const int tap = item / WV, r = item - tap * WV;
          *reinterpret_cast<f32x4*>(s_w + tap * KC * COG + 4 * r) = pre_w[i];
        }
      }
      __syncthreads();
    }
    if (cc + KC < cin_g) {
      // ---- global -> registers for the next chunk (in flight during the MFMA loop below) ----
      const int cn = cc + KC;
      {  // without a prologue any readable 16 bytes do (never used): no branch around the loads
        const int ch = g * cin_g + cn + 4 * my_c4;
        psc = *reinterpret_cast<const float4*>(a.in_scale ? a.in_scale + ch : a.weights);
        psh = *reinterpret_cast<const float4*>(a.in_scale ? a.in_shift + ch : a.weights);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int item = min(tid + i * CT, NITEM - 1);
        const int px = item / C4;
        const int py = px / PW, pxx = px - py * PW;
        const int iy = iy0 + py, ix = ix0 + pxx;
        // branch-free: a clamped address is always loaded (conditional loads split the block and make the compiler
        // wait for every outstanding load at each join); out-of-image pixels are zeroed at commit
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        pre_p[i] = *reinterpret_cast<const f32x4*>(at_off(in_n, (unsigned)((cy * a.W + cx) * a.Cin + cn + 4 * my_c4) << 2));
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = min(tid + i * CT, KS * KS * WV - 1);
        const int tap = item / WV, r = item - tap * WV;
        pre_w[i] = *reinterpret_cast<const f32x4*>(at_off(wg, (unsigned)((tap * cin_g + cn) * COG + 4 * r) << 2));
      }
    }
    if (cc >= 0) {
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const float* ap = s_patch + a_base + (ky * PW + kx) * KP;
        const float* bp = s_w + tap * KC * COG + b_base;
#pragma unroll
        for (int k2 = 0; k2 < KC / 2; ++k2) {
          float av[NTM], bv[NTN];
#pragma unroll
          for (int m = 0; m < NTM; ++m) av[m] = ap[m * (TB * S * PW * KP) + 2 * k2];
#pragma unroll
          for (int t = 0; t < NTN; ++t) bv[t] = bp[(2 * k2) * COG + t * 32];
#pragma unroll
          for (int m = 0; m < NTM; ++m)
#pragma unroll
            for (int t = 0; t < NTN; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[t], acc[m][t], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: affine (bias / folded BN) from the accumulators, then through LDS so that residual
  // loads and stores are 16 bytes per lane (8 lanes = one pixel's 128 contiguous bytes): 4x fewer, wider
  // memory instructions than the accumulator layout (one channel per lane, 4 bytes) would give ----
  float* out_n = a.out + (size_t)n * a.Ho * a.Wo * a.Cout;
  const float* res_n = a.residual ? a.residual + (size_t)n * a.Ho * a.Wo * a.Cout : nullptr;
  float* s_tile = lds + wave * (32 * 32);  // [32 pixels][32 channels] per wave; patch / weights are dead now
#pragma unroll
  for (int m = 0; m < NTM; ++m) {
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const int ch = g * COG + t * 32 + (lane & 31);
      const float os = a.out_scale ? a.out_scale[ch] : 1.0f;
      const float ob = a.out_shift ? a.out_shift[ch] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // pixel index inside the wave tile
        s_tile[i * 32 + (lane & 31)] = acc[m][t][r] * os + ob;
      }
      // each wave transposes through its own 4 KB (the patch / weights are dead after the loop's last barrier) and
      // the LDS operations of one wave complete in order: no workgroup barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // two loops, not one with a conditional residual load inside: a load in the loop makes the compiler wait for
      // the PREVIOUS store before every store (cpx_cnn_bf3.hip)
      if (res_n) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int f = it * 64 + lane;         // float4 index inside the tile
          const int i = f >> 3, c4 = f & 7;
          const int oy = oy0 + m * TB + WR * wave + i / TW, ox = ox0 + i % TW;
          if (oy < a.Ho && ox < a.Wo) {
            float4 v = *reinterpret_cast<const float4*>(s_tile + i * 32 + 4 * c4);
            const unsigned o = (unsigned)((oy * a.Wo + ox) * a.Cout + g * COG + t * 32 + 4 * c4) << 2;  // inside one sample
            const float4 rv = *reinterpret_cast<const float4*>(at_off(res_n, o));
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            if (a.relu) {
              v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
            }
            *reinterpret_cast<float4*>(at_off(out_n, o)) = v;
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int f = it * 64 + lane;
          const int i = f >> 3, c4 = f & 7;
          const int oy = oy0 + m * TB + WR * wave + i / TW, ox = ox0 + i % TW;
          if (oy < a.Ho && ox < a.Wo) {
            float4 v = *reinterpret_cast<const float4*>(s_tile + i * 32 + 4 * c4);
            const unsigned o = (unsigned)((oy * a.Wo + ox) * a.Cout + g * COG + t * 32 + 4 * c4) << 2;  // inside one sample
            if (a.relu) {
              v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
            }
            *reinterpret_cast<float4*>(at_off(out_n, o)) = v;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// first layer: 2 -> 16 channels, one input channel per group (conv1_1, wr_resnet.py:12-20): HBM-bound
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  const size_t total = (size_t)a.N * a.Ho * a.Wo * a.groups;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(idx % a.groups);
    size_t p = idx / a.groups;
    const int ox = (int)(p % a.Wo);
    p /= a.Wo;
    const int oy = (int)(p % a.Ho);
    const int n = (int)(p / a.Ho);
    float acc[16];
    for (int co = 0; co < cout_g; ++co) acc[co] = 0.0f;
    const float* in_n = a.in + (size_t)n * a.H * a.W * a.Cin + g * cin_g;
    const float* wg = a.weights + (size_t)g * a.ksize * a.ksize * cin_g * cout_g;
    for (int ky = 0; ky < a.ksize; ++ky)
      for (int kx = 0; kx < a.ksize; ++kx) {
        const int iy = oy * a.stride - a.pad_top + ky, ix = ox * a.stride - a.pad_left + kx;
        if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W) continue;
        for (int ci = 0; ci < cin_g; ++ci) {
          float v = in_n[((size_t)iy * a.W + ix) * a.Cin + ci];
          if (a.in_scale) v = fmaxf(v * a.in_scale[g * cin_g + ci] + a.in_shift[g * cin_g + ci], 0.0f);
          const float* wr = wg + ((size_t)(ky * a.ksize + kx) * cin_g + ci) * cout_g;
          for (int co = 0; co < cout_g; ++co) acc[co] += v * wr[co];
        }
      }
    const size_t ob = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + g * cout_g;
    for (int co = 0; co < cout_g; ++co) {
      const int ch = g * cout_g + co;
      float v = acc[co] * (a.out_scale ? a.out_scale[ch] : 1.0f) + (a.out_shift ? a.out_shift[ch] : 0.0f);
      if (a.residual) v += a.residual[ob + co];
      if (a.relu) v = fmaxf(v, 0.0f);
      a.out[ob + co] = v;
    }
  }
}

// conv1_1 as the network uses it (2 input channels, 2 groups of 8 output channels, 3x3, stride 1, SAME, bias
// only): one thread per pixel computes all 16 output channels -- float2 input loads, four 16-byte stores per
// pixel (64 contiguous bytes), weights through uniform (scalar) loads.  HBM-bound: 8 B read, 64 B written per pixel.
__global__ __launch_bounds__(256) void conv1_kernel(ConvArgs a) {
  // (guarded: behind a fused first block that computed this layer itself, the tensor is only needed by the block's rerun)
  if (a.guard != nullptr && *a.guard == 0) return;
  __shared__ float4 s_out[4 * 256];
  const size_t total = (size_t)a.N * a.Ho * a.Wo;
  // the loop condition is wave-uniform (a wave's first pixel): in the last wave of a launch whose pixel count is not a
  // multiple of 64 the lanes past `total` compute a clamped (discarded) pixel and still take part in the store loop
  // below, which writes slots of OTHER lanes' pixels
  const int lane0 = threadIdx.x & 63;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx - lane0 < total; idx += (size_t)gridDim.x * blockDim.x) {
    // (32-bit index arithmetic: launch_conv refuses a launch of 2^32 pixels or more; the 64-bit divisions cost more vector
    // instructions per pixel than the convolution itself)
    unsigned p = (unsigned)(idx < total ? idx : total - 1);
    const unsigned prow = p / (unsigned)a.Wo;
    const int ox = (int)(p - prow * (unsigned)a.Wo);
    const int n = (int)(prow / (unsigned)a.Ho);
    const int oy = (int)(prow - (unsigned)n * (unsigned)a.Ho);
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    const float* in_n = a.in + (size_t)n * a.H * a.W * 2;
    // all nine loads first and unconditionally (clamped address, padding zeroed afterwards): a load behind
    // `if (inside)` is waited for before the next one is issued -- nine dependent trips to memory per pixel
    float2 tap[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy - a.pad_top + ky, ix = ox - a.pad_left + kx;
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        tap[ky * 3 + kx] = *reinterpret_cast<const float2*>(in_n + ((size_t)cy * a.W + cx) * 2);
      }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy - a.pad_top + ky, ix = ox - a.pad_left + kx;
        const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        float2 v = tap[ky * 3 + kx];
        v.x = inside ? v.x : 0.0f;
        v.y = inside ? v.y : 0.0f;
        const float* w0 = a.weights + (ky * 3 + kx) * 8;       // group 0: [tap][1][8]
        const float* w1 = a.weights + 72 + (ky * 3 + kx) * 8;  // group 1
#pragma unroll
        for (int c = 0; c < 8; ++c) {  // (fused multiply-adds: half the vector instructions of this vector-bound kernel)
          acc[c] = __fmaf_rn(v.x, w0[c], acc[c]);
          acc[8 + c] = __fmaf_rn(v.y, w1[c], acc[8 + c]);
        }
      }
    // the 64 bytes of a pixel go through the wave's 4 KB of LDS so that a store instruction writes 1 KB of
    // consecutive addresses (a lane storing its own pixel's four float4 scatters 16-byte pieces 64 bytes apart:
    // four partial writes per cache line)
    float4* tile = s_out + (threadIdx.x >> 6) * 256;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 r;
      r.x = acc[4 * q + 0] + a.out_shift[4 * q + 0];
      r.y = acc[4 * q + 1] + a.out_shift[4 * q + 1];
      r.z = acc[4 * q + 2] + a.out_shift[4 * q + 2];
      r.w = acc[4 * q + 3] + a.out_shift[4 * q + 3];
      tile[lane * 4 + q] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // the wave's pixels are consecutive: idx0 .. idx0 + 63 (the tail of the last wave is cut by `total`)
    const size_t idx0 = idx - lane;
    float4* o = reinterpret_cast<float4*>(a.out + idx0 * 16);
    const size_t left = (total - idx0) * 4;  // float4 slots that exist behind idx0
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = q * 64 + lane;
      if ((size_t)f < left) o[f] = tile[f];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// final_bn -> ReLU -> GlobalAveragePooling2D -> Dense(n_labels) (+ sigmoid): one workgroup per sample
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a, int vmax) {
  extern __shared__ __attribute__((aligned(16))) float s_feat[];  // [vmax] x 2: layer input / output, ping-pong
  __shared__ float s_red[2];
  const int n = blockIdx.x;
  const float* x = a.in + (size_t)n * a.HW * a.C;
  float* cur = s_feat;
  float* nxt = s_feat + vmax;
  for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
    const float sc = a.bn_scale[c], sh = a.bn_shift[c];
    float s = 0.0f;
    for (int p = 0; p < a.HW; ++p) s += fmaxf(x[(size_t)p * a.C + c] * sc + sh, 0.0f);
    cur[c] = s / (float)a.HW;
  }
  __syncthreads();
  int width = a.C;
  for (int k = 0; k < a.n_hidden; ++k) {  // Dense(size, relu) layers of hyperparams.dense_sizes (kerasmodel.py:337-339)
    const int out = a.hidden_sizes[k];
    const float* w = a.hidden_w[k];
    for (int j = threadIdx.x; j < out; j += blockDim.x) {
      float s = a.hidden_b[k][j];
      for (int c = 0; c < width; ++c) s += cur[c] * w[(size_t)c * out + j];
      nxt[j] = fmaxf(s, 0.0f);
    }
    __syncthreads();
    float* t = cur;
    cur = nxt;
    nxt = t;
    width = out;
  }
  for (int l = threadIdx.x; l < a.L; l += blockDim.x) {
    float s = a.dense_b[l];
    for (int c = 0; c < width; ++c) s += cur[c] * a.dense_w[(size_t)c * a.L + l];
    a.logits[(size_t)n * a.L + l] = s;
    nxt[l] = s;
    if (a.probs && a.activation == CPX_HEAD_SIGMOID) a.probs[(size_t)n * a.L + l] = 1.0f / (1.0f + expf(-s));
  }
  if (a.probs && a.activation == CPX_HEAD_SOFTMAX) {  // softmax as Keras evaluates it: exp(x - max) / sum
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = nxt[0];
      for (int l = 1; l < a.L; ++l) m = fmaxf(m, nxt[l]);
      float z = 0.0f;
      for (int l = 0; l < a.L; ++l) z += expf(nxt[l] - m);
      s_red[0] = m;
      s_red[1] = z;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < a.L; l += blockDim.x) a.probs[(size_t)n * a.L + l] = expf(nxt[l] - s_red[0]) / s_red[1];
  }
}

template <int KC, int NTN, int S, int KS, int NTM, int TW>
static int launch_conv_t(const ConvArgs& a, hipStream_t s) {
  constexpr int TB = 128 / TW;
  constexpr int TH = TB * NTM;
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  size_t lds = (((size_t)PH * PW * (KC + 1) + 3) / 4 * 4 + (size_t)KS * KS * KC * 32 * NTN) * sizeof(float);
  if (lds < 4 * 32 * 32 * sizeof(float)) lds = 4 * 32 * 32 * sizeof(float);  // epilogue tiles
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_mfma_kernel<KC, NTN, S, KS, NTM, TW>), lds_ready, 160 * 1024 - 1024)) return -1;
  const int tiles = ((a.Wo + TW - 1) / TW) * ((a.Ho + TH - 1) / TH);
  hipLaunchKernelGGL((conv_mfma_kernel<KC, NTN, S, KS, NTM, TW>), dim3(tiles * a.N, a.groups), dim3(CT), lds, s, a);
  return 0;
}

}  // namespace

int launch_conv(const ConvArgs& a, hipStream_t s) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  if (cin_g == 1 && cout_g == 8 && a.groups == 2 && a.ksize == 3 && a.stride == 1 && !a.in_scale && !a.out_scale &&
      a.out_shift && !a.residual && !a.relu) {
    const size_t total = (size_t)a.N * a.Ho * a.Wo;
    if (total >= (1ull << 32)) return -3;  // (conv1_kernel indexes pixels in 32 bits)
    const int blocks = (int)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
    hipLaunchKernelGGL(conv1_kernel, dim3(blocks), dim3(256), 0, s, a);
    return 0;
  }
  if (cin_g < 8) {
    if (cout_g > 16) return -2;
    const size_t total = (size_t)a.N * a.Ho * a.Wo * a.groups;
    const int blocks = (int)((total + 255) / 256 > 65535 * 4 ? 65535 * 4 : (total + 255) / 256);
    hipLaunchKernelGGL(conv_direct_kernel, dim3(blocks), dim3(256), 0, s, a);
    return 0;
  }
#define CPX_CONV_CASE(KC, NTN, S, KS, NTM, TW)                                               \
  if (cout_g == 32 * NTN && a.stride == S && a.ksize == KS && (cin_g % KC) == 0 && cin_g >= KC) \
    return launch_conv_t<KC, NTN, S, KS, NTM, TW>(a, s);
  // (channels per group, stride, kernel) combinations of WR-ResNet-22-4 with groups = 2
  if (cin_g == 8) { CPX_CONV_CASE(8, 1, 1, 3, 1, 16) CPX_CONV_CASE(8, 1, 1, 1, 1, 16) }
  CPX_CONV_CASE(CPX_CONV_KC_S2, 1, 1, 3, CPX_CONV_NTM_S2, 16)
  CPX_CONV_CASE(8, 2, 2, 3, 1, 16)
  CPX_CONV_CASE(16, 2, 2, 1, 1, 16)
  CPX_CONV_CASE(CPX_CONV_KC_S3, 2, 1, 3, CPX_CONV_NTM_S3, 16)
  CPX_CONV_CASE(8, 4, 3, 3, 1, CPX_CONV_TW_S4)
  CPX_CONV_CASE(8, 4, 3, 1, 1, CPX_CONV_TW_S4)
  CPX_CONV_CASE(CPX_CONV_KC_S4, 4, 1, 3, CPX_CONV_NTM_S4, CPX_CONV_TW_S4)
#undef CPX_CONV_CASE
  return -2;
}

// fp16x2 bookkeeping at the end of a forward: ovf[2 ..] are the overflow words of the forward's blocks; ovf[0] = whether
// any of them was raised (what cpx_cnn_last_overflow reads), ovf[1] counts the forwards in which one was
__global__ void count_overflow_kernel(int* ovf, int n_words) {
  if (threadIdx.x == 0) {
    int any = 0;
    for (int k = 0; k < n_words; ++k) any |= ovf[2 + k];
    ovf[0] = any;
    if (any) ovf[1] += 1;
  }
}
void launch_count_overflow(int* ovf, int n_words, hipStream_t s) {
  hipLaunchKernelGGL(count_overflow_kernel, dim3(1), dim3(64), 0, s, ovf, n_words);
}

void launch_head(const HeadArgs& a, hipStream_t s) {
  int vmax = a.C > a.L ? a.C : a.L;
  for (int k = 0; k < a.n_hidden; ++k) vmax = a.hidden_sizes[k] > vmax ? a.hidden_sizes[k] : vmax;
  hipLaunchKernelGGL(head_kernel, dim3(a.N), dim3(256), (size_t)2 * vmax * sizeof(float), s, a, vmax);
}

}  // namespace cpx
