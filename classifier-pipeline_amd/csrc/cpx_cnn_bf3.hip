// cpx_cnn_bf3.hip -- the 3x3 stride-1 convolutions of WR-ResNet with float32 operands on the bf16 matrix pipe.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate on gfx950.  A float32 value splits EXACTLY into three
// bf16 terms (x = x0 + x1 + x2: round to nearest, subtract, repeat; 3 x 8 significant bits plus the signs of the
// remainders cover the 24-bit significand), every bf16 x bf16 product is exact in float32, and the accumulator is
// float32 as before.  Of the nine cross products the six with i + j <= 2 are issued; the three dropped ones are below
// 2^-26 of the product, i.e. below the rounding of a float32 accumulation step, so the result is float32 arithmetic
// in every respect the 1e-3 logit tolerance (and the 2e-4 of tests/test_cnn_gpu.py) can see -- measured against a
// float64 convolution it is as close as the float32 MFMA kernel (tests/test_cnn_gpu.py::test_bf16x3_is_f32_accurate).
// Six v_mfma_f32_32x32x16_bf16 (32 cycles each) do the work of eight 32x32x2_f32 (64 cycles each): 2.67x.
//
// Same implicit GEMM as cpx_cnn.hip (M = a band of 128 output pixels per workgroup, 32 per wave; N = 32-channel
// tiles of one group; K walked as 16-channel chunk x tap), same fused prologue (BatchNorm + ReLU while staging, zero
// padding kept exactly 0) and epilogue (affine, residual, ReLU, one store).  What differs is the operand staging:
//   patch   LDS [plane 0..2][k half 0..1][pixel] of 16-byte entries (8 bf16 = this pixel's channels 8h..8h+7):
//           an A fragment is one ds_read_b128 per plane, consecutive lanes on consecutive entries (conflict-free)
//   weights pre-split on the device into the same entry format, [chunk][plane][tap][k half][column]; staging is
//           a straight copy and a B fragment is one ds_read_b128 per plane
// Activations stay float32 in HBM; the split happens on the way into LDS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "cpx_kernels.h"

namespace cpx {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// a native vector, not HIP's uint4 struct: struct copies become memcpy calls that keep a staging array in scratch
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
// uniform base + 32-bit unsigned byte offset: the form the compiler turns into `global_load v, v_off, s[base]` (one
// offset register per lane instead of 64-bit address arithmetic per access; everything addressed here stays inside
// one sample or one weight image: < 2^32 bytes)
template <typename T>
__device__ __forceinline__ const T* at_off(const T* base, unsigned bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + bytes);
}
// element offset of pixel (y, x) in an NHWC sample: both products stay below 2^24 (y * W + x < 2^17 pixels, x C <= 256
// channels), so the full-rate 24-bit multiply does (a 32-bit v_mul_lo_u32 / v_mad_u64_u32 takes four issue slots)
__device__ __forceinline__ unsigned pix_off(int y, int x, int W, int C) {
  return __umul24(__umul24((unsigned)y, (unsigned)W) + (unsigned)x, (unsigned)C);
}
// ReLU of a float as an integer maximum: one instruction (fmaxf(x, 0) on a value of unknown origin is canonicalised first)
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }
// two float32 -> their three bf16 planes (packed pairs)
__device__ __forceinline__ void split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
  p0 = pack2(a, b);
  // (opaque: seeing through the pack, the compiler converts `a` a second time just to shift it -- one vector
  // instruction more per level and pair; the halves of the packed word are what is wanted)
  asm volatile("" : "+v"(p0));
  const float ra = a - bf_lo(p0), rb = b - bf_hi(p0);  // exact
  p1 = pack2(ra, rb);
  asm volatile("" : "+v"(p1));
  const float sa = ra - bf_lo(p1), sb = rb - bf_hi(p1);  // exact
  p2 = pack2(sa, sb);
}

// two float32 -> TWO bf16 planes, each rounded to nearest (v_cvt_pk_bf16_f32): hi + lo carries 16 significand bits,
// |x - hi - lo| <= 2^-16 |x| (CPX_CNN_MATH_BF16X2: three products per K step instead of six)
__device__ __forceinline__ unsigned pack2_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ void split_pair2(float a, float b, unsigned& p0, unsigned& p1) {
  p0 = pack2_rne(a, b);
  const float ra = a - bf_lo(p0), rb = b - bf_hi(p0);  // exact
  p1 = pack2_rne(ra, rb);
}

// CPX_CNN_MATH_FP16X2: two float32 -> two fp16 planes, each rounded to nearest (v_cvt_pk_f16_f32): 11 + 11 significand
// bits and the sign of the remainder, |x - hi - lo| <= 2^-22 |x| while lo is a normal fp16 (|x| >= 2^-3 after the
// caller's scaling) and <= 2^-25 absolute below that (v_mfma_*_f16 keeps subnormal inputs: scratch/fp16_probe.hip)
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned pack2_h(float a, float b) {
  f16x2 v = {(_Float16)a, (_Float16)b};  // v_cvt_pk_f16_f32: round to nearest even
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split_pair2_h(float a, float b, unsigned& p0, unsigned& p1) {
  p0 = pack2_h(a, b);
  // lo = fp16(x - hi): the difference is exact in float32, so ONE mixed-precision fused multiply-add per element (x * 1.0 - hi,
  // float32 inside, rounded to nearest fp16 once, into the low / high half of p1) is the same value as convert-back, subtract,
  // convert -- three instructions per pair instead of six
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(p1) : "v"(a), "v"(p0));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p1) : "v"(b), "v"(p0));
}
// the two-plane split of the chosen kind (H: fp16, else bf16)
template <bool H>
__device__ __forceinline__ void split2(float a, float b, unsigned& p0, unsigned& p1) {
  if (H) split_pair2_h(a, b, p0, p1);
  else split_pair2(a, b, p0, p1);
}
// one plane product on the matrix pipe: 16-byte fragments as they come out of LDS
template <bool H>
__device__ __forceinline__ f32x4 mfma16(u32x4 w, u32x4 x, f32x4 c) {
  if (H) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
template <bool H>
__device__ __forceinline__ f32x16 mfma32(u32x4 x, u32x4 w, f32x16 c) {
  if (H) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, w), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), c, 0, 0, 0);
}
constexpr float F16_MAX = 65504.0f;
// producer-side split (ConvArgs::out_planes): four consecutive channels of one pixel, activated and already multiplied by
// the consumer's act_scale, as the consumer stages them: [hi 0..3 | lo 0..3] in the 16 bytes the float32 values would take
__device__ __forceinline__ u32x4 planes_of(f32x4 v) {
  unsigned h0, h1, l0, l1;
  split_pair2_h(v.x, v.y, h0, l0);
  split_pair2_h(v.z, v.w, h1, l1);
  return u32x4{h0, h1, l0, l1};
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float max_abs4(f32x4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// entry test shared by the split-operand kernels: a fp16 layer is skipped once the network's overflow word is set (its
// three-plane rerun follows), the rerun is skipped while it is clear
#define BF3_ENTRY_GUARD(H_)                          \
  if (H_) {                                          \
    if (*a.ovf != 0) return;                         \
  } else if (a.guard != nullptr && *a.guard == 0) {  \
    return;                                          \
  }

constexpr int KC = 16;  // channels per K step of the bf16 MFMA = per staged chunk

// NB bands of 128 output pixels per workgroup, CT threads: four waves share a band (32 pixels each); with CT = 512
// the second set of four waves takes every other band, so NTM = NB / (CT / 256) bands per wave
// q = n / d for n < 2^22, d < 2^12 with a host-computed M = floor(2^42 / d) + 1: two scalar multiplies instead of
// the ~25-instruction reciprocal sequence a runtime division costs (four of them per workgroup otherwise)
struct TileDiv {
  unsigned long long m_nsplit, m_tx, m_ty;
  int nsplit, tiles_x, tiles_y;
  int run;  // conv_bf3w_kernel: tiles per workgroup along x (tiles_x then counts runs)
  int total;  // PERSIST: workgroup-sized units of work along x (the grid then is smaller and each workgroup walks its share)
};
// PERSIST (last template parameter of the three kernels): the form the guarded bf16x3 rerun of a fp16x2 layer is launched
// in.  A rerun that has nothing to do must cost nothing: 409,600 workgroups that load one word and leave take 174 us
// (two resident per CU: each costs a dispatch round trip), 1,024 take 2.5 us (profiles/r05_fp16_probe.txt) -- so the
// rerun is a small grid whose workgroups walk the tiles, bid0 = blockIdx.x, + gridDim.x, ... < td.total, with a barrier
// between tiles (the LDS images are reused).  The ordinary launches keep one tile per workgroup: the loop folds away.
#define BF3_TILE_LOOP_BEGIN for (int bid0 = blockIdx.x;;) {
#define BF3_TILE_LOOP_END        \
  if (!PERSIST) break;           \
  bid0 += gridDim.x;             \
  if (bid0 >= td.total) break;   \
  __syncthreads();               \
  }
__device__ __forceinline__ int div_magic(int n, unsigned long long m) { return (int)(((unsigned long long)n * m) >> 42); }

// (two workgroups per CU is what the LDS footprint allows: the register budget is pinned to match)
// C8: the layer with 8 input channels per group (one 16-byte k half per pixel).  A K = 16 step then pairs two TAPS
// instead of two channel halves: lanes 0-31 (k half 0) read tap 2 s, lanes 32-63 tap 2 s + 1 of the same 8-channel
// entry array -- five steps for the nine taps (the tenth tap has zero weights) instead of nine half-empty ones.
// PL: bf16 planes per operand (3: the exact split; 2: CPX_CNN_MATH_BF16X2 -- see conv_bf3w_kernel; not with C8)
// H: the two planes are fp16 (CPX_CNN_MATH_FP16X2: ConvArgs::half; PL == 2 only)
template <int NTN, int S, int NB, int TW, int CT, bool C8, int PL = 3, bool H = false, bool PERSIST = false>
// (the strided forms fill the LDS with one workgroup: their waves may use the registers of the absent second one)
__global__ __launch_bounds__(CT) __attribute__((amdgpu_waves_per_eu(S > 1 ? CT / 256 : (CT >= 1024 ? 4 : CT / 128), S > 1 ? CT / 256 : (CT >= 1024 ? 4 : CT / 128)))) void conv_bf3_kernel(ConvArgs a, const uint4* __restrict__ wimg, TileDiv td) {
  constexpr int KS = 3;
  constexpr int WSETS = CT / 256;    // sets of four waves
  constexpr int NTM = NB / WSETS;    // bands per wave
  static_assert(NTM * WSETS == NB, "bands must divide among the wave sets");
  constexpr int TB = 128 / TW;   // rows of a band
  constexpr int WR = 32 / TW;    // rows of a wave's 32-pixel tile
  constexpr int TH = TB * NB;    // output rows of a workgroup
  constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
  constexpr int NPX = PH * PW;
  constexpr int COGW = 32 * NTN;  // output channels of this workgroup
  constexpr int KH = C8 ? 1 : 2;      // 8-channel entries (k halves) stored per patch pixel
  constexpr int QPP = C8 ? 2 : 4;     // 16-byte pieces of float32 input per patch pixel and chunk
  constexpr int NSTEP = C8 ? 5 : 9;   // K = 16 steps per chunk
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  static_assert(PL == 3 || (PL == 2 && !C8), "two planes: not for the tap-paired 8-channel form");
  static_assert(!H || PL == 2, "fp16 planes come in twos");
  BF3_ENTRY_GUARD(H)
  uint4* s_patch = lds4;                // [PL][KH][NPX]
  uint4* s_w = lds4 + PL * KH * NPX;    // [PL][NSTEP][2][COGW]

  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wset = tid >> 8;
  BF3_TILE_LOOP_BEGIN
  int bid = bid0;
  if (!PERSIST && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);  // XCD-contiguous tiles (cpx_cnn.hip)
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  int q = div_magic(bid, td.m_nsplit);
  const int ns = bid - q * td.nsplit;  // the column slices of one tile run next to each other: they share the patch in L2
  bid = q;
  q = div_magic(bid, td.m_tx);
  const int txi = bid - q * td.tiles_x;
  bid = q;
  q = div_magic(bid, td.m_ty);
  const int tyi = bid - q * td.tiles_y;
  const int n = q;
  const int g = blockIdx.y;
  const int oy0 = tyi * TH, ox0 = txi * TW;
  const int iy0 = oy0 * S - a.pad_top, ix0 = ox0 * S - a.pad_left;
  const float* in_n = a.in + (size_t)n * a.H * a.W * a.Cin + (size_t)g * cin_g;
  const int nchunks = C8 ? 1 : cin_g / KC;
  // weight image: [g][chunk][3][NSTEP][2][cout_g] entries
  const uint4* wg = wimg + (size_t)g * nchunks * (2 * PL * NSTEP) * cout_g + (size_t)ns * COGW;
  // a tile whose patch lies inside the image needs no padding tests (uniform)
  const bool interior = iy0 >= 0 && iy0 + PH <= a.H && ix0 >= 0 && ix0 + PW <= a.W;
  // a tile whose OUTPUT lies inside the map needs no clamps / bounds tests in the residual preload and the stores, and
  // the offsets of a lane's values then differ by wave-uniform amounts: one per-lane base + scalar terms (uniform)
  const bool full_tile = oy0 + TH <= a.Ho && ox0 + TW <= a.Wo;

  // The residual goes INTO the accumulators before the first product (possible when the output is not scaled):
  // its loads are in flight under the whole tile instead of stalling the epilogue, and they cost no registers.
  // out = (res + sum of products) + bias: the same terms as before in another float32 order.
  const bool res_in_acc = a.residual != nullptr && a.out_scale == nullptr;
  f32x16 acc[NTM][NTN];
  if (res_in_acc) {
    const float* res_n = a.residual + (size_t)n * a.Ho * a.Wo * a.Cout + (g * cout_g + ns * COGW);  // (uniform)
    const int rcol = lane & 31;
    float rs[NTN];  // (H) the accumulators hold act_scale * w_scale[channel] times the sum: so must the residual
#pragma unroll
    for (int t = 0; t < NTN; ++t) rs[t] = H ? a.w_scale[g * cout_g + ns * COGW + t * 32 + rcol] * a.act_scale : 1.0f;
    if (full_tile) {
#pragma unroll
      for (int m = 0; m < NTM; ++m) {
        // register r holds pixel ic(r) + 4 (lane >> 5) of the wave tile; TW is a multiple of 8, so the lane part never
        // carries into the row: row / column split into a constant of r and the lane's share
        const unsigned lb = (pix_off(oy0 + (wset + m * WSETS) * TB + WR * wave, ox0 + 4 * (lane >> 5), a.Wo, a.Cout) + (unsigned)rcol) << 2;
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ic = (r & 3) + 8 * (r >> 2);
            const unsigned ub = (unsigned)(((ic / TW) * a.Wo + (ic % TW)) * a.Cout + t * 32) << 2;  // (scalar)
            acc[m][t][r] = *at_off(res_n, lb + ub);
            if (H) acc[m][t][r] *= rs[t];
          }
      }
    } else {
#pragma unroll
      for (int m = 0; m < NTM; ++m)
#pragma unroll
        for (int t = 0; t < NTN; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // pixel of the wave tile this register holds
            const int oy = min(oy0 + (wset + m * WSETS) * TB + WR * wave + i / TW, a.Ho - 1), ox = min(ox0 + i % TW, a.Wo - 1);
            acc[m][t][r] = *at_off(res_n, (pix_off(oy, ox, a.Wo, a.Cout) + (unsigned)(t * 32 + rcol)) << 2);
            if (H) acc[m][t][r] *= rs[t];
          }
    }
  } else {
#pragma unroll
    for (int m = 0; m < NTM; ++m)
#pragma unroll
      for (int t = 0; t < NTN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;
  }

  const int pi = lane & 31, kh = lane >> 5;
  const int prow = WR * wave + pi / TW, pcol = pi % TW;
  const int a_base = (C8 ? 0 : kh * NPX) + ((prow + wset * TB) * S) * PW + pcol * S;  // band m of this wave: wset + m * WSETS
  const int b_base = kh * COGW + (lane & 31);

  // staging: an item is ONE 16-byte piece (4 channels) of a patch pixel's 16-channel chunk, and the four pieces of a
  // pixel sit on adjacent lanes, so that a wave's load instruction covers whole contiguous 64-byte runs -- a quarter
  // of the cache-line requests the (pixel, 8 channels)-per-lane form made, which kept the vector-memory issue busy for
  // thousands of cycles per chunk (in-kernel s_memtime stamps, scratch/patches/README.md).  BatchNorm + ReLU + split
  // happen here; a piece fills the low or high 8 bytes of an LDS entry.
  constexpr int NITEM = NPX * QPP;
  constexpr int NP = (NITEM + CT - 1) / CT;
  constexpr int NW = 2 * PL * NSTEP * COGW;   // uint4 entries of a weight chunk
  constexpr int NWI = (NW + CT - 1) / CT;
  const int my_q = tid & (QPP - 1);           // float32 input: CT is a multiple of 4, so a thread keeps its piece
  u32x4 pre_p[NP];
  u32x4 pre_w[NWI];
  f32x4 psc, psh;
  for (int cc = -1; cc < nchunks; ++cc) {
    if (cc >= 0) {
      // ---- registers -> LDS: BatchNorm + ReLU prologue, split into bf16 planes ----
      float vmax = 0.0f;  // (H) largest scaled magnitude this thread stages
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int item = tid + i * CT;
        if (item < NITEM) {
          const int px = C8 ? item >> 1 : item >> 2;
          bool inside = true;
          if (!interior) {
            const int py = px / PW, pxx = px - py * PW;
            const int iy = iy0 + py, ix = ix0 + pxx;
            // the loads are unconditional (clamped addresses, below): padding pixels are zeroed here, after the
            // prologue, exactly as TensorFlow pads the activated tensor
            inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
          }
          {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(pre_p[i][j]);
            if (a.in_scale) {  // (H: scale and shift carry act_scale)
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(v[j], psc[j], psh[j]), 0.0f);
            } else if (H) {
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] *= a.act_scale;
            }
            if (!interior) {  // (uniform: most tiles skip the selects)
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
            }
            if (H) vmax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), vmax);
            unsigned q0[2], q1[2], q2[2];
            if (PL == 3) {
              split_pair(v[0], v[1], q0[0], q1[0], q2[0]);
              split_pair(v[2], v[3], q0[1], q1[1], q2[1]);
            } else {
              split2<H>(v[0], v[1], q0[0], q1[0]);
              split2<H>(v[2], v[3], q0[1], q1[1]);
            }
            // channels 4 q .. 4 q + 3 of the chunk: entry of k half q >> 1, its low or high 8 bytes
            uint2* sp2 = reinterpret_cast<uint2*>(s_patch);
            const int e2 = ((my_q >> 1) * NPX + px) * 2 + (my_q & 1);
            sp2[(0 * KH * NPX) * 2 + e2] = make_uint2(q0[0], q0[1]);
            sp2[(1 * KH * NPX) * 2 + e2] = make_uint2(q1[0], q1[1]);
            if (PL == 3) sp2[(2 * KH * NPX) * 2 + e2] = make_uint2(q2[0], q2[1]);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = tid + i * CT;
        if (item < NW) reinterpret_cast<u32x4*>(s_w)[item] = pre_w[i];
      }
      if (H && vmax > F16_MAX) atomicOr(a.ovf, 1);  // out of fp16's range: the three-plane kernel reruns the layer
    }
    // (the next chunk's loads are issued BEFORE the barrier: their registers are free once the commit above has read
    // them, and the time a wave waits for the others at the barrier then counts towards hiding the loads' latency)
    if (cc + 1 < nchunks) {
      // ---- global -> registers for the next chunk (in flight during the MFMA loop below) ----
      const int cn = C8 ? 0 : (cc + 1) * KC;
      {  // without a prologue any readable 16 bytes do (never used): no branch around the loads
        const int ch = g * cin_g + cn;  // (uniform; the lane's piece goes into the offset)
        const float* scp = a.in_scale ? a.in_scale + ch : reinterpret_cast<const float*>(wimg);
        const float* shp = a.in_scale ? a.in_shift + ch : reinterpret_cast<const float*>(wimg);
        unsigned qoff = (unsigned)my_q << 4;
        asm volatile("" : "+v"(qoff));
        psc = *reinterpret_cast<const f32x4*>(at_off(scp, qoff));
        psh = *reinterpret_cast<const f32x4*>(at_off(shp, qoff));
        if (H) {  // relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k), exactly
          psc *= a.act_scale;
          psh *= a.act_scale;
        }
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int item = min(tid + i * CT, NITEM - 1);
        const int px = C8 ? item >> 1 : item >> 2;
        const int py = px / PW, pxx = px - py * PW;
        const int iy = iy0 + py, ix = ix0 + pxx;
        // branch-free: a clamped address is always loaded (conditional loads split the block and make the
        // compiler wait for all outstanding loads at every join); out-of-image pixels are zeroed at commit
        const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1);
        pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, a.W, a.Cin) + (unsigned)(cn + 4 * my_q)) << 2));
      }
      const uint4* wc = wg + (size_t)(cc + 1) * (2 * PL * NSTEP) * cout_g;
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = min(tid + i * CT, NW - 1);
        const int pth = item / COGW, col = item - pth * COGW;
        unsigned woff = (unsigned)(pth * cout_g + col) << 4;
        asm volatile("" : "+v"(woff));  // (kept 32-bit: hoisted out of the chunk loop it becomes a 64-bit pair per load)
        pre_w[i] = *at_off(reinterpret_cast<const u32x4*>(wc), woff);
      }
    }
    if (cc >= 0) {
      __syncthreads();
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        // the patch offset of this step's tap: the same for the whole wave, or (C8) tap 2 st for k half 0 and tap
        // 2 st + 1 for k half 1 (the tenth tap does not exist: its weights are zero, any address will do)
        int koff;
        if (C8) {
          const int tl = 2 * st + kh, tc = tl > 8 ? 8 : tl;
          const int ky = (tc * 11) >> 5;  // tc / 3 for tc <= 8
          koff = ky * PW + (tc - 3 * ky);
        } else {
          const int ky = st / KS, kx = st - ky * KS;
          koff = ky * PW + kx;
        }
        u32x4 av[NTM][PL], bv[NTN][PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
#pragma unroll
          for (int m = 0; m < NTM; ++m)
            av[m][p] = __builtin_bit_cast(u32x4, s_patch[p * KH * NPX + a_base + m * (WSETS * TB * S * PW) + koff]);
#pragma unroll
          for (int t = 0; t < NTN; ++t)
            bv[t][p] = __builtin_bit_cast(u32x4, s_w[(p * NSTEP + st) * 2 * COGW + b_base + t * 32]);
        }
#pragma unroll
        for (int m = 0; m < NTM; ++m)
#pragma unroll
          for (int t = 0; t < NTN; ++t) {
            // smallest terms first: x1*y1, x0*y2, x2*y0, x0*y1, x1*y0, x0*y0
            if (PL == 3) {
              acc[m][t] = mfma32<false>(av[m][1], bv[t][1], acc[m][t]);
              acc[m][t] = mfma32<false>(av[m][0], bv[t][PL - 1], acc[m][t]);
              acc[m][t] = mfma32<false>(av[m][PL - 1], bv[t][0], acc[m][t]);
            }
            acc[m][t] = mfma32<H>(av[m][0], bv[t][1], acc[m][t]);
            acc[m][t] = mfma32<H>(av[m][1], bv[t][0], acc[m][t]);
            acc[m][t] = mfma32<H>(av[m][0], bv[t][0], acc[m][t]);
          }
      }
      __syncthreads();
    }
  }

  // ---- fused 1x1 shortcut of a stage's first block: out += conv1x1(block input, stride) -- K = the block input's
  // channels per group, on the float32 MFMA (same accumulator layout), operands straight from global memory ----
  if (a.sc_in) {
    const int sc_cg = a.sc_cin / a.groups;
    const float* wsc = a.sc_w + (size_t)g * sc_cg * cout_g + ns * COGW + (lane & 31);
    float ss[NTN];  // (H) into scaled accumulators: the shortcut's weights take the column's scale
#pragma unroll
    for (int t = 0; t < NTN; ++t) ss[t] = H ? a.w_scale[g * cout_g + ns * COGW + t * 32 + (lane & 31)] * a.act_scale : 1.0f;
#pragma unroll
    for (int m = 0; m < NTM; ++m) {
      const int oy = min(oy0 + (wset + m * WSETS) * TB + prow, a.Ho - 1), ox = min(ox0 + pcol, a.Wo - 1);
      const float* psc_in = a.sc_in + (((size_t)n * a.sc_H + oy * a.sc_stride) * a.sc_W + ox * a.sc_stride) * a.sc_cin +
                            g * sc_cg + kh;
      for (int k2 = 0; k2 < sc_cg; k2 += 2) {
        const float av = psc_in[k2];
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
          float wv = wsc[(size_t)(k2 + kh) * cout_g + t * 32];
          if (H) wv *= ss[t];
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc[m][t], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue (as cpx_cnn.hip): affine from the accumulators, through LDS for 16-byte residual loads / stores.
  // Each wave transposes through its own 4 KB of LDS (the patch / weights are dead after the loop's last barrier),
  // and LDS operations of one wave complete in order, so no workgroup barrier is needed here ----
  float* out_n = a.out + (size_t)n * a.Ho * a.Wo * a.Cout;
  const float* res_n = (a.residual && !res_in_acc) ? a.residual + (size_t)n * a.Ho * a.Wo * a.Cout : nullptr;
  float* s_tile = reinterpret_cast<float*>(lds4) + (wset * 4 + wave) * (32 * 32);
  const int ch0 = g * cout_g + ns * COGW;
#pragma unroll
  for (int m = 0; m < NTM; ++m) {
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const int ch = ch0 + t * 32 + (lane & 31);
      float os = a.out_scale ? a.out_scale[ch] : 1.0f;
      if (H) os *= a.w_unscale[ch] * a.act_unscale;  // (powers of two: exact)
      float ob = (a.out_shift ? a.out_shift[ch] : 0.0f) + (a.sc_in && a.sc_bias ? a.sc_bias[ch] : 0.0f);
      if (a.out_planes) {  // the consumer's range scale rides on the affine: relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k)
        os *= a.out_act_scale;
        ob *= a.out_act_scale;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);  // pixel index inside the wave tile
        s_tile[i * 32 + (lane & 31)] = acc[m][t][r] * os + ob;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // two loops, not one with a conditional residual load inside: a load in the loop makes the compiler wait for
      // vmcnt(0) before every store, i.e. for the PREVIOUS store to complete -- four serialized round trips per tile
      // (seen in the ISA and as 5-8 k cycles of epilogue in the stamps).  Without loads the four stores go out back
      // to back.
      if (res_n) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int f = it * 64 + lane;
          const int i = f >> 3, c4 = f & 7;
          const int oy = oy0 + (wset + m * WSETS) * TB + WR * wave + i / TW, ox = ox0 + i % TW;
          if (oy < a.Ho && ox < a.Wo) {
            float4 v = *reinterpret_cast<const float4*>(s_tile + i * 32 + 4 * c4);
            const int o = (int)pix_off(oy, ox, a.Wo, a.Cout) + ch0 + t * 32 + 4 * c4;  // inside one sample: < 2^31
            const float4 rv = *reinterpret_cast<const float4*>(at_off(res_n, (unsigned)o << 2));
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            if (a.relu) {
              v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
            }
            *reinterpret_cast<float4*>(at_off(out_n, (unsigned)o << 2)) = v;
          }
        }
      } else if (full_tile) {
        // lane f of store `it` writes pixel 8 it + (lane >> 3), channels 4 (lane & 7) ..: a per-lane base + a scalar
        const unsigned lb = (pix_off(oy0 + (wset + m * WSETS) * TB + WR * wave, ox0 + (lane >> 3), a.Wo, a.Cout) +
                             (unsigned)(ch0 + t * 32 + 4 * (lane & 7))) << 2;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          float4 v = *reinterpret_cast<const float4*>(s_tile + (it * 64 + lane) * 4);
          const unsigned ub = (unsigned)((((it * 8) / TW) * a.Wo + ((it * 8) % TW)) * a.Cout) << 2;  // (scalar)
          if (a.relu) {
            v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          }
          if (a.out_planes) {  // (uniform) the next layer's fp16 planes instead of float32
            const f32x4 vv = {v.x, v.y, v.z, v.w};
            if (max_abs4(vv) > F16_MAX) atomicOr(a.ovf, 1);
            *reinterpret_cast<u32x4*>(at_off(out_n, lb + ub)) = planes_of(vv);
          } else {
            *reinterpret_cast<float4*>(at_off(out_n, lb + ub)) = v;
          }
        }
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int f = it * 64 + lane;
          const int i = f >> 3, c4 = f & 7;
          const int oy = oy0 + (wset + m * WSETS) * TB + WR * wave + i / TW, ox = ox0 + i % TW;
          if (oy < a.Ho && ox < a.Wo) {
            float4 v = *reinterpret_cast<const float4*>(s_tile + i * 32 + 4 * c4);
            const int o = (int)pix_off(oy, ox, a.Wo, a.Cout) + ch0 + t * 32 + 4 * c4;  // inside one sample: < 2^31
            if (a.relu) {
              v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
            }
            if (a.out_planes) {
              const f32x4 vv = {v.x, v.y, v.z, v.w};
              if (max_abs4(vv) > F16_MAX) atomicOr(a.ovf, 1);
              *reinterpret_cast<u32x4*>(at_off(out_n, (unsigned)o << 2)) = planes_of(vv);
            } else {
              *reinterpret_cast<float4*>(at_off(out_n, (unsigned)o << 2)) = v;
            }
          }
        }
      }
      if (m + 1 < NTM || t + 1 < NTN) {  // the wave's LDS tile is reused by the next accumulator tile
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }
  BF3_TILE_LOOP_END
}

// ---------------------------------------------------------------------------------------------------------------------
// Flattened tiling for small maps.  The 27 x 27 maps of stage 4 (54 x 54 at frame size 64) fill 4 x 32 bands to only
// 81 %; here a workgroup takes 128 CONSECUTIVE output positions of a sample in row-major order (729 = 5.7 bands:
// 95 %), stages the input rows they touch at full width, and every lane derives its pixel from its position.  Same
// staging format, MFMA loop and epilogue as conv_bf3_kernel; the output offset of a position is simply p * Cout.
// PL: bf16 planes per operand (3: the exact split, six products; 2: CPX_CNN_MATH_BF16X2, three -- see conv_bf3w_kernel)
// H: the two planes are fp16 (CPX_CNN_MATH_FP16X2; see conv_bf3_kernel)
// CT: threads = 64 x the 32-position wave tiles of a workgroup's band (256: 128 positions; 512: 256 positions -- a staged
// weight chunk then feeds twice the products: at 128 positions three workgroups per CU pull 37 KB of weights each per 16-channel
// chunk, ~50 B per clock and CU out of L2, which is about what an XCD's L2 delivers)
template <int NTN, int NPXC, int PL, bool H = false, bool PERSIST = false, int CT = 256>
__global__ __launch_bounds__(CT) void conv_bf3flat_kernel(ConvArgs a, const uint4* __restrict__ wimg, TileDiv td) {
  constexpr int KS = 3, BAND = CT / 2;
  constexpr int COGW = 32 * NTN;
  static_assert(!H || PL == 2, "fp16 planes come in twos");
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  BF3_ENTRY_GUARD(H)
  uint4* s_patch = lds4;                 // [PL][2][NPXC]
  uint4* s_w = lds4 + 2 * PL * NPXC;     // [PL][9][2][COGW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  BF3_TILE_LOOP_BEGIN
  int bid = bid0;
  if (!PERSIST && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  int q = div_magic(bid, td.m_nsplit);
  const int ns = bid - q * td.nsplit;
  bid = q;
  q = div_magic(bid, td.m_tx);
  const int ti = bid - q * td.tiles_x;  // band of 128 positions inside the sample
  const int n = q;
  const int g = blockIdx.y;
  const int M = a.Ho * a.Wo;
  const int p0 = ti * BAND;
  const int y_first = p0 / a.Wo, y_last = min(p0 + BAND - 1, M - 1) / a.Wo;
  const int PW = a.W + 2, PH = y_last - y_first + 3;  // stride 1, 3 x 3, SAME: one halo row / column each side
  const int npx = PH * PW;
  const int iy0 = y_first - a.pad_top, ix0 = -a.pad_left;
  const float* in_n = a.in + (size_t)n * a.H * a.W * a.Cin + (size_t)g * cin_g;
  const int nchunks = cin_g / KC;
  const uint4* wg = wimg + (size_t)g * nchunks * (18 * PL) * cout_g + (size_t)ns * COGW;

  const bool res_in_acc = a.residual != nullptr && a.out_scale == nullptr;  // see conv_bf3_kernel
  f32x16 acc[NTN];
  if (res_in_acc) {
    const float* res_n = a.residual + (size_t)n * M * a.Cout + (g * cout_g + ns * COGW);  // (uniform)
    const int rcol = lane & 31;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const float rs = H ? a.w_scale[g * cout_g + ns * COGW + t * 32 + rcol] * a.act_scale : 1.0f;  // (see conv_bf3_kernel)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        acc[t][r] = *at_off(res_n, (__umul24((unsigned)min(p0 + wave * 32 + i, M - 1), (unsigned)a.Cout) + (unsigned)(t * 32 + rcol)) << 2);
        if (H) acc[t][r] *= rs;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  }

  const int kh = lane >> 5;
  const int pl = min(p0 + wave * 32 + (lane & 31), M - 1);  // this lane's output position (clamped: never stored)
  const int ly = pl / a.Wo, lx = pl - ly * a.Wo;
  const int a_base = kh * NPXC + (ly - y_first) * PW + lx;
  const int b_base = kh * COGW + (lane & 31);

  constexpr int NP = (2 * NPXC + CT - 1) / CT;
  constexpr int NW = 18 * PL * COGW;
  constexpr int NWI = (NW + CT - 1) / CT;
  const int my_h = tid & 1;
  int item_py[NP], item_px[NP], item_e[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int item = min(tid + i * CT, 2 * npx - 1);  // surplus threads repeat the last item (same value, same slot)
    const int px = item >> 1;
    item_py[i] = px / PW;
    item_px[i] = px - item_py[i] * PW;
    item_e[i] = my_h * NPXC + px;
  }
  f32x4 pre_p[NP][2];
  u32x4 pre_w[NWI];
  f32x4 psc[2], psh[2];
  for (int cc = -1; cc < nchunks; ++cc) {
    if (cc >= 0) {
      // ---- registers -> LDS: BatchNorm + ReLU prologue, padding zeroed, split into bf16 planes ----
      float vmax = 0.0f;  // (H) largest scaled magnitude this thread stages
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int iy = iy0 + item_py[i], ix = ix0 + item_px[i];
        const bool inside = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pre_p[i][j >> 2][j & 3];
        if (a.in_scale) {  // (H: scale and shift carry act_scale)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = fmaxf(__fmaf_rn(v[j], psc[j >> 2][j & 3], psh[j >> 2][j & 3]), 0.0f);
        } else if (H) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] *= a.act_scale;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = inside ? v[j] : 0.0f;
        if (H) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) vmax = fmaxf(fmaxf(fabsf(v[j]), fabsf(v[j + 1])), vmax);
        }
        unsigned q0[4], q1[4], q2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (PL == 3) split_pair(v[2 * j], v[2 * j + 1], q0[j], q1[j], q2[j]);
          else split2<H>(v[2 * j], v[2 * j + 1], q0[j], q1[j]);
        }
        u32x4* sp4 = reinterpret_cast<u32x4*>(s_patch);
        sp4[0 * 2 * NPXC + item_e[i]] = u32x4{q0[0], q0[1], q0[2], q0[3]};
        sp4[1 * 2 * NPXC + item_e[i]] = u32x4{q1[0], q1[1], q1[2], q1[3]};
        if (PL == 3) sp4[2 * 2 * NPXC + item_e[i]] = u32x4{q2[0], q2[1], q2[2], q2[3]};
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = tid + i * CT;
        if (item < NW) reinterpret_cast<u32x4*>(s_w)[item] = pre_w[i];
      }
      if (H && vmax > F16_MAX) atomicOr(a.ovf, 1);  // out of fp16's range: the three-plane kernel reruns the layer
      __syncthreads();
    }
    if (cc + 1 < nchunks) {
      // ---- global -> registers for the next chunk (in flight during the MFMA loop below), branch-free ----
      const int cn = (cc + 1) * KC;
      {
        const int ch = g * cin_g + cn;  // (uniform; the lane's half goes into the offset)
        const float* scp = a.in_scale ? a.in_scale + ch : reinterpret_cast<const float*>(wimg);
        const float* shp = a.in_scale ? a.in_shift + ch : reinterpret_cast<const float*>(wimg);
        unsigned hoff = (unsigned)my_h << 5;
        asm volatile("" : "+v"(hoff));
        psc[0] = *reinterpret_cast<const f32x4*>(at_off(scp, hoff));
        psc[1] = *reinterpret_cast<const f32x4*>(at_off(scp, hoff + 16));
        psh[0] = *reinterpret_cast<const f32x4*>(at_off(shp, hoff));
        psh[1] = *reinterpret_cast<const f32x4*>(at_off(shp, hoff + 16));
        if (H) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            psc[k] *= a.act_scale;
            psh[k] *= a.act_scale;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int cy = min(max(iy0 + item_py[i], 0), a.H - 1), cx = min(max(ix0 + item_px[i], 0), a.W - 1);
        const unsigned soff = (pix_off(cy, cx, a.W, a.Cin) + (unsigned)(cn + 8 * my_h)) << 2;
        pre_p[i][0] = *reinterpret_cast<const f32x4*>(at_off(in_n, soff));
        pre_p[i][1] = *reinterpret_cast<const f32x4*>(at_off(in_n, soff + 16));
      }
      const u32x4* wc = reinterpret_cast<const u32x4*>(wg + (size_t)(cc + 1) * (18 * PL) * cout_g);
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = min(tid + i * CT, NW - 1);
        const int pth = item / COGW, col = item - pth * COGW;
        unsigned woff = (unsigned)(pth * cout_g + col) << 4;
        asm volatile("" : "+v"(woff));  // (see conv_bf3_kernel)
        pre_w[i] = *at_off(wc, woff);
      }
    }
    if (cc >= 0) {
#pragma unroll
      for (int tap = 0; tap < KS * KS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        u32x4 av[PL], bv[NTN][PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
          av[p] = __builtin_bit_cast(u32x4, s_patch[p * 2 * NPXC + a_base + ky * PW + kx]);
#pragma unroll
          for (int t = 0; t < NTN; ++t)
            bv[t][p] = __builtin_bit_cast(u32x4, s_w[(p * 9 + tap) * 2 * COGW + b_base + t * 32]);
        }
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
          if (PL == 3) {
            acc[t] = mfma32<false>(av[1], bv[t][1], acc[t]);
            acc[t] = mfma32<false>(av[0], bv[t][PL - 1], acc[t]);
            acc[t] = mfma32<false>(av[PL - 1], bv[t][0], acc[t]);
          }
          acc[t] = mfma32<H>(av[0], bv[t][1], acc[t]);
          acc[t] = mfma32<H>(av[1], bv[t][0], acc[t]);
          acc[t] = mfma32<H>(av[0], bv[t][0], acc[t]);
        }
      }
      __syncthreads();
    }
  }

  if (a.sc_in) {  // fused 1x1 shortcut (see conv_bf3_kernel)
    const int sc_cg = a.sc_cin / a.groups;
    const float* wsc = a.sc_w + (size_t)g * sc_cg * cout_g + ns * COGW + (lane & 31);
    const float* psc_in = a.sc_in + (((size_t)n * a.sc_H + ly * a.sc_stride) * a.sc_W + lx * a.sc_stride) * a.sc_cin +
                          g * sc_cg + kh;
    float ss[NTN];  // (H: see conv_bf3_kernel)
#pragma unroll
    for (int t = 0; t < NTN; ++t) ss[t] = H ? a.w_scale[g * cout_g + ns * COGW + t * 32 + (lane & 31)] * a.act_scale : 1.0f;
    for (int k2 = 0; k2 < sc_cg; k2 += 2) {
      const float av = psc_in[k2];
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        float wv = wsc[(size_t)(k2 + kh) * cout_g + t * 32];
        if (H) wv *= ss[t];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: as conv_bf3_kernel; the NHWC offset of flattened position p is p * Cout ----
  float* out_n = a.out + (size_t)n * M * a.Cout;
  const float* res_n = (a.residual && !res_in_acc) ? a.residual + (size_t)n * M * a.Cout : nullptr;
  float* s_tile = reinterpret_cast<float*>(lds4) + wave * (32 * 32);
  const int ch0 = g * cout_g + ns * COGW;
#pragma unroll
  for (int t = 0; t < NTN; ++t) {
    const int ch = ch0 + t * 32 + (lane & 31);
    float os = a.out_scale ? a.out_scale[ch] : 1.0f;
    if (H) os *= a.w_unscale[ch] * a.act_unscale;  // (powers of two: exact)
    const float ob = (a.out_shift ? a.out_shift[ch] : 0.0f) + (a.sc_in && a.sc_bias ? a.sc_bias[ch] : 0.0f);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      s_tile[i * 32 + (lane & 31)] = acc[t][r] * os + ob;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (res_n) {  // (two loops: see conv_bf3_kernel)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int f = it * 64 + lane;
        const int i = f >> 3, c4 = f & 7;
        const int p = p0 + wave * 32 + i;
        if (p < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(s_tile + i * 32 + 4 * c4);
          const int o = (int)__umul24((unsigned)p, (unsigned)a.Cout) + ch0 + t * 32 + 4 * c4;  // inside one sample: < 2^31
          v += *reinterpret_cast<const f32x4*>(at_off(res_n, (unsigned)o << 2));
          if (a.relu) {
            v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          }
          *reinterpret_cast<f32x4*>(at_off(out_n, (unsigned)o << 2)) = v;
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int f = it * 64 + lane;
        const int i = f >> 3, c4 = f & 7;
        const int p = p0 + wave * 32 + i;
        if (p < M) {
          f32x4 v = *reinterpret_cast<const f32x4*>(s_tile + i * 32 + 4 * c4);
          const int o = (int)__umul24((unsigned)p, (unsigned)a.Cout) + ch0 + t * 32 + 4 * c4;  // inside one sample: < 2^31
          if (a.relu) {
            v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          }
          *reinterpret_cast<f32x4*>(at_off(out_n, (unsigned)o << 2)) = v;
        }
      }
    }
    if (t + 1 < NTN) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  BF3_TILE_LOOP_END
}

// ---------------------------------------------------------------------------------------------------------------------
// conv_bf3w_kernel: the stride-1 3x3 layers with 32 / 64 channels per group on v_mfma_f32_16x16x32_bf16.
//
// Why a second MFMA shape: the split-operand loop is bound by the clock the chip holds under it, not by issue slots
// (profiles/r01_conv_power_probe.md), and the chip holds a higher clock on the 16x16x32 form: scratch/mfma_shape_probe.hip
// (this kernel's inner loop alone, operands re-read from LDS, random planes) runs 284 TFLOP/s at 1.70 GHz with
// 32x32x16 and 327 TFLOP/s at 1.97 GHz with 16x16x32, at equal cycles (profiles/r04_mfma_shape_probe.txt).
//
// K = 32 per instruction = the 32 channels of one tap, so the staged chunk is 32 channels: the patch (18 x 18 pixels of a
// 16 x 16 tile) is committed once per 32 channels, the weights in sub-chunks of one kernel ROW (3 taps x 32 channels x
// 32 columns x 3 planes = 18 KB) so that patch + weights stay at 79 KB and two workgroups share a CU.
//   patch   LDS [plane][quarter pair][pixel][2] of 16-byte entries (8 channels): the 32 bytes of a pixel's quarter
//           pair are adjacent, which makes ds_read_b128 conflict-free in its lane groups ({0-3,12-15,20-27}, ...: one
//           k quarter's pixels 0-3 / 12-15 and the next quarter's 4-11 tile a 256-byte bank period exactly)
//   weights LDS [plane][kx][quarter][column]; device image [g][chunk][ky][plane][kx][quarter][cout_g]
// The roles of the operands are swapped (A = weights: M = output channels; B = pixels: N = pixels), so a lane's four
// accumulator values are four CONSECUTIVE channels of one pixel: the residual preload, the affine and the store are
// one 16-byte access per accumulator tile straight from the registers -- no transpose through LDS, no epilogue barrier.
// A wave owns two tile rows (2 x 16 pixels) x 32 columns = 2 x 2 accumulator tiles; 12 ds_read_b128 feed 24 MFMAs.
constexpr int KW = 32;
#ifndef CPX_BF3W_RUN
#define CPX_BF3W_RUN 1  // tiles a workgroup walks along x (launch_bf3w)
#endif
#ifndef CPX_BF3W_LDSBN
#define CPX_BF3W_LDSBN 0
#endif
constexpr int W_TW = 16, W_TH = 16, W_PW = 18, W_NPX = 18 * 18;
constexpr int W_NPXP = 326;  // pixels per (plane, quarter pair) region: 326 * 32 B = 64 mod 128, the two regions' ds_write_b64 lanes then use different banks
constexpr int w_patch_entries(int planes) { return planes * 2 * W_NPXP * 2; }
constexpr int w_rows_resident(int planes) { return planes == 2 ? 3 : 1; }
constexpr int w_wsub_entries(int planes) { return w_rows_resident(planes) * planes * 3 * 4 * 32; }  // a weight sub-chunk
// WALK: the workgroup walks a run of td.run tiles along x (else exactly one tile: the loop and the per-use
// laundering of the staging bases fold away); LDSBN: BatchNorm scale / shift read back from LDS at each patch commit
// instead of living in eight registers; NH: 32-column slices of the group's output channels the workgroup computes from
// ONE staged patch (NH = 2 for 64 columns per group: the patch of a tile is loaded, normalised and split once instead
// of once per slice -- the slices' weights alternate through the same 18 KB, twice the accumulators)
// NG: GROUPS the workgroup walks on its tile, one after the other (experiment, -DCPX_BF3W_NG=2: the second group's patch
// and first weights are in flight under the first group's products, as the next chunk's are in a layer with 64
// channels per group, and the tile's index arithmetic is paid once for both -- 171 vs 183 TFLOP/s on stage 2: not shipped)
// PL: bf16 planes per operand.  3 = the exact split (six products per K step: every term down to 2^-24 of the float32
// product); 2 = CPX_CNN_MATH_BF16X2: both operands as hi + lo rounded to nearest (16 significand bits, relative error
// <= 2^-16 each), three products w0 x1 + w1 x0 + w0 x0 -- half the matrix work, two thirds of the staging and LDS.
// H (PL == 2): the two planes are fp16 -- CPX_CNN_MATH_FP16X2: 11 + 11 significand bits (2^-22 per operand against 2^-16),
// three products on v_mfma_f32_16x16x32_f16 at the bf16 form's rate; the operands are scaled by powers of two into
// fp16's range and the accumulators scaled back in the epilogue (ConvArgs::half)
template <bool WALK, bool LDSBN, int NH, int NG, int PL, bool H = false, bool PERSIST = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_bf3w_kernel(ConvArgs a, const uint4* __restrict__ wimg, TileDiv td) {
  static_assert(NG == 1 || (!WALK && !LDSBN && NH == 1), "the group walk is built for the one-slice, one-tile form");
  static_assert(PL == 2 || PL == 3, "two or three bf16 planes per operand");
  static_assert(!H || (PL == 2 && LDSBN), "fp16 planes come in twos (and with the BatchNorm parameters in LDS)");
  BF3_ENTRY_GUARD(H)
  // WR: kernel rows of a chunk's weights resident in LDS at a time.  Three planes: one (patch + one row = 79 KB, two
  // workgroups per CU).  Two planes leave room for all three (41 + 36 KB): one weight commit and two barriers per chunk
  // and column slice instead of three and six.
  constexpr int WR = w_rows_resident(PL);
  constexpr int W_PATCH = PL * 2 * W_NPXP * 2;  // entries
  constexpr int W_WROW = PL * 3 * 4 * 32;       // entries of one kernel row's weights (32 columns)
  constexpr int W_WSUB = WR * W_WROW;           // entries of a weight sub-chunk
  constexpr int CT = 512;
  // per-thread staging / output indices are re-derived from the thread index at each use (WALK: they would be carried
  // across the tile loop; NH > 1: sixteen more accumulator registers leave no room to keep them across the products --
  // kept, the compiler spills eight of them to scratch per tile: +22 % HBM traffic on the stage-3 launches, measured)
  constexpr bool LAUNDER = WALK || NH > 1 || NG > 1 || PL == 2;  // (PL == 2: twenty prefetch registers for the three weight rows)
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  uint4* s_patch = lds4;
  uint4* s_w = lds4 + W_PATCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  BF3_TILE_LOOP_BEGIN
  int bid = bid0;
  if (!PERSIST && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);  // XCD-contiguous tiles (cpx_cnn.hip)
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  int qd = div_magic(bid, td.m_nsplit);
  const int ns = bid - qd * td.nsplit;
  bid = qd;
  qd = div_magic(bid, td.m_tx);
  const int txr = bid - qd * td.tiles_x;  // run of td.run consecutive tiles of one tile row (td.tiles_x counts runs)
  bid = qd;
  qd = div_magic(bid, td.m_ty);
  const int tyi = bid - qd * td.tiles_y;
#ifdef CPX_BF3W_ALIAS_N  // experiment (scratch/patches/README.md): every sample reads and writes the first CPX_BF3W_ALIAS_N ones -- same work, no HBM traffic
  const int n = qd & (CPX_BF3W_ALIAS_N - 1);
#else
  const int n = qd;
#endif
  const int g0 = blockIdx.y * NG;
  const int oy0 = tyi * W_TH;
  const int iy0 = oy0 - a.pad_top;
  const int run = WALK ? td.run : 1;
  const int tx_end = WALK ? min((txr + 1) * run, (a.Wo + W_TW - 1) / W_TW) : txr + 1;
  const float* in_n = a.in + (size_t)n * a.H * a.W * a.Cin;
  const int nch = cin_g / KW;
  const uint4* wg = wimg + (size_t)ns * (32 * NH);
  const int a_base = ((q >> 1) * W_NPXP + (2 * wave) * W_PW + i16) * 2 + (q & 1);
  const int b_base = q * 32 + i16;
  // Staging items: ONE 16-byte piece (4 channels) of a patch pixel's 32-channel chunk, the eight pieces of a pixel on
  // adjacent lanes (a wave's load instruction covers whole 128-byte pixel runs): item tid + 512 i = pixel
  // (tid >> 3) + 64 i, piece tid & 7.  2592 items = 5 per thread and a sixth for threads 0..31: an even share of the
  // BatchNorm / split work per SIMD (a row-aligned mapping that spares the division was measured 10 % slower:
  // three SIMDs then carry six items per wave and one carries three).
  constexpr int NITEM = W_NPX * 8;
  constexpr int NP = (NITEM + CT - 1) / CT;
  constexpr int NWI = (W_WSUB + CT - 1) / CT;
  const int my_q8 = tid & 7;
  const int st_e2_0 = ((my_q8 >> 2) * W_NPXP + (tid >> 3)) * 4 + (my_q8 & 3);  // uint2 slot of item 0, plane 0; an item further = 64 pixels
  u32x4 pre_p[NP];
  u32x4 pre_w[NWI];
  // BatchNorm scale / shift of the group's input channels: staged once per workgroup (read back at each patch commit
  // instead of living in eight registers across the phases)
  f32x4* s_bn = reinterpret_cast<f32x4*>(lds4 + W_PATCH + W_WSUB);
  // (!LDSBN; with LDSBN they are never written and their copies below never read: initialising them costs eight
  // instructions per tile for nothing)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wuninitialized"
  f32x4 psc_r, psh_r;
  if (LDSBN && a.in_scale) {
    if (tid < cin_g) {  // (H: relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k), exactly)
      reinterpret_cast<float*>(s_bn)[tid] = a.in_scale[g0 * cin_g + tid] * (H ? a.act_scale : 1.0f);
      reinterpret_cast<float*>(s_bn)[cin_g + tid] = a.in_shift[g0 * cin_g + tid] * (H ? a.act_scale : 1.0f);
    }
    __syncthreads();
  }
  // (the per-thread bases are laundered at every use: left visible, the compiler keeps one address register per
  // staged item alive -- and stepping -- across all phases, and the prefetched pieces themselves go to scratch)
#define BF3W_ISSUE_W(G_, C_, R_, HF_)                                                                     \
  {                                                                                                       \
    const uint4* wc = wg + (size_t)((((G_) * nch + (C_)) * 3 + (R_)) * (12 * PL)) * cout_g + (HF_) * 32;  \
    unsigned woff = (__umul24((unsigned)(tid >> 5), (unsigned)cout_g) + (unsigned)(tid & 31)) << 4;       \
    if (LAUNDER) asm volatile("" : "+v"(woff));                                                           \
    _Pragma("unroll") for (int i = 0; i < NWI; ++i) {                                                     \
      /* item tid + 512 i = 16 i rows further down; the last slice (rows 32..35) exists for tid < 128 only */ \
      const unsigned wo = (i == NWI - 1 && (W_WSUB % CT) != 0) ? (tid < (W_WSUB % CT) ? woff + ((unsigned)(16 * i * cout_g) << 4) : woff) : woff + ((unsigned)(16 * i * cout_g) << 4); \
      pre_w[i] = *at_off(reinterpret_cast<const u32x4*>(wc), wo);                                         \
    }                                                                                                     \
  }
  // (branch-free, clamped addresses: see conv_bf3_kernel; out-of-image pixels are zeroed at commit)
#define BF3W_ISSUE_P(G_, C_, IX0_)                                                                        \
  {                                                                                                       \
    int t8 = tid >> 3;                                                                                    \
    if (LAUNDER) asm volatile("" : "+v"(t8));                                                             \
    int my_q8 = tid & 7;                                                                                  \
    if (LAUNDER) asm volatile("" : "+v"(my_q8));                                                          \
    if (!LDSBN) {                                                                                         \
      const int ch = (G_) * cin_g + (C_) * KW;                                                            \
      const float* scp = a.in_scale ? a.in_scale + ch : reinterpret_cast<const float*>(wimg);             \
      const float* shp = a.in_scale ? a.in_shift + ch : reinterpret_cast<const float*>(wimg);             \
      unsigned qoff = (unsigned)my_q8 << 4;                                                               \
      asm volatile("" : "+v"(qoff));                                                                      \
      psc_r = *reinterpret_cast<const f32x4*>(at_off(scp, qoff));                                         \
      psh_r = *reinterpret_cast<const f32x4*>(at_off(shp, qoff));                                         \
    }                                                                                                     \
    const unsigned coff = (unsigned)((G_) * cin_g + (C_) * KW + 4 * my_q8);                               \
    _Pragma("unroll") for (int i = 0; i < NP; ++i) {                                                      \
      const int px = min(t8 + 64 * i, W_NPX - 1);                                                         \
      /* px / 18 for px < 324; 24-bit multiplies by hand: a laundered index has no known range, and the 32-bit   \
         v_mul_lo_u32 the compiler then picks takes four issue slots */                                      \
      const int py = (int)(__umul24((unsigned)px, 3641u) >> 16), pxx = __mul24(py, -W_PW) + px;           \
      const int cy = min(max(iy0 + py, 0), a.H - 1), cx = min(max((IX0_) + pxx, 0), a.W - 1);             \
      pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, a.W, a.Cin) + coff) << 2)); \
    }                                                                                                     \
  }
  BF3W_ISSUE_W(g0, 0, 0, 0)
  BF3W_ISSUE_P(g0, 0, txr * run * W_TW - a.pad_left)
  // The workgroup walks a run of tiles along x: the next tile's patch and first weights are in flight under the
  // current tile's products exactly as the next chunk's are, so only the first tile of a run waits for its loads
  // with nothing else to do (the other workgroup of the CU aside).
  for (int txi = txr * run; txi < tx_end; ++txi) {
  const int ox0 = txi * W_TW, ix0 = ox0 - a.pad_left;
  const bool interior = iy0 >= 0 && iy0 + 18 <= a.H && ix0 >= 0 && ix0 + 18 <= a.W;
  const bool full_tile = oy0 + W_TH <= a.Ho && ox0 + W_TW <= a.Wo;

  // this lane's two pixels (tile rows 2 wave, 2 wave + 1; column i16) and its channel quad inside a 16-column tile
  int ch_l;
  unsigned opix[2];  // element offsets of the (clamped) pixels in the output map
  bool ovalid[2];
  auto out_pixels = [&](const int g) __attribute__((always_inline)) {
    int t = tid;
    if (LAUNDER) asm volatile("" : "+v"(t));
    const int i16_ = t & 15, q_ = (t >> 4) & 3, wave_ = t >> 6;
    ch_l = g * cout_g + ns * (32 * NH) + 4 * q_;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int oy = oy0 + 2 * wave_ + pt, ox = ox0 + i16_;
      ovalid[pt] = full_tile || (oy < a.Ho && ox < a.Wo);
      opix[pt] = pix_off(min(oy, a.Ho - 1), min(ox, a.Wo - 1), a.Wo, a.Cout) + (unsigned)ch_l;
    }
  };
  const bool res_in_acc = a.residual != nullptr && a.out_scale == nullptr;  // see conv_bf3_kernel
  f32x4 acc[NH][2][2];
  auto init_acc = [&](const int g) __attribute__((always_inline)) {  // a group's accumulators: zero, or its residual
  out_pixels(g);
  if (res_in_acc) {
    const float* res_n = a.residual + (size_t)n * a.Ho * a.Wo * a.Cout;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          acc[hf][ct][pt] = *reinterpret_cast<const f32x4*>(at_off(res_n, (opix[pt] + 32u * hf + 16u * ct) << 2));
    if (H) {  // the accumulators hold act_scale * w_scale[channel] times the sum: so must the residual
#pragma unroll
      for (int hc = 0; hc < 2 * NH; ++hc) {
        const f32x4 rs = *reinterpret_cast<const f32x4*>(a.w_scale + ch_l + 16 * hc) * a.act_scale;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[hc >> 1][hc & 1][pt] *= rs;
      }
    }
  } else {
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) acc[hf][ct][pt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  }
  };
  // a group's fused shortcut, affine, residual, ReLU and stores
  auto finish = [&](const int g) __attribute__((always_inline)) {

  // ---- fused 1x1 shortcut (see conv_bf3_kernel), on v_mfma_f32_16x16x4_f32: A = weights [column][k], B = pixels ----
  if (a.sc_in) {
    const int sc_cg = a.sc_cin / a.groups;
    const float* wsc = a.sc_w + ((size_t)g * sc_cg + q) * cout_g + ns * (32 * NH) + i16;
    const float* pin[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const int oy = min(oy0 + 2 * wave + pt, a.Ho - 1), ox = min(ox0 + i16, a.Wo - 1);
      pin[pt] = a.sc_in + (((size_t)n * a.sc_H + oy * a.sc_stride) * a.sc_W + ox * a.sc_stride) * a.sc_cin + g * sc_cg + q;
    }
    float ss[NH][2];  // (H) into scaled accumulators: the shortcut's weights take their column's scale
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        ss[hf][ct] = H ? a.w_scale[g * cout_g + ns * (32 * NH) + 32 * hf + 16 * ct + i16] * a.act_scale : 1.0f;
    for (int k4 = 0; k4 < sc_cg; k4 += 4) {
      float xs[2], ws[NH][2];
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) xs[pt] = pin[pt][k4];
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          ws[hf][ct] = wsc[(size_t)k4 * cout_g + 32 * hf + 16 * ct];
          if (H) ws[hf][ct] *= ss[hf][ct];
        }
#pragma unroll
      for (int hf = 0; hf < NH; ++hf)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
            acc[hf][ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[hf][ct], xs[pt], acc[hf][ct][pt], 0, 0, 0);
    }
  }

  // ---- epilogue: affine, residual, ReLU and one 16-byte store per accumulator tile, straight from the registers ----
  float* out_n = a.out + (size_t)n * a.Ho * a.Wo * a.Cout;
  const float* res_n = (a.residual && !res_in_acc) ? a.residual + (size_t)n * a.Ho * a.Wo * a.Cout : nullptr;
  if (LAUNDER) out_pixels(g);  // (re-derived: see LAUNDER)
#pragma unroll
  for (int hc = 0; hc < 2 * NH; ++hc) {
    const int hf = hc >> 1, ct = hc & 1;
    const int ch = ch_l + 16 * hc;
    f32x4 os = {1.0f, 1.0f, 1.0f, 1.0f}, ob = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.out_scale) os = *reinterpret_cast<const f32x4*>(a.out_scale + ch);
    if (H) os *= *reinterpret_cast<const f32x4*>(a.w_unscale + ch) * a.act_unscale;  // (powers of two: exact)
    if (a.out_shift) ob = *reinterpret_cast<const f32x4*>(a.out_shift + ch);
    if (a.sc_in && a.sc_bias) ob += *reinterpret_cast<const f32x4*>(a.sc_bias + ch);
    if (a.out_planes) {  // the consumer's range scale rides on the affine: relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k)
      os *= a.out_act_scale;
      ob *= a.out_act_scale;
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      f32x4 v = acc[hf][ct][pt];
      if (H || a.out_scale || a.out_planes) {  // (uniform branches kept as branches: see the patch commit)
        asm volatile("");
        v = v * os;
      }
      v += ob;
      const unsigned o = (opix[pt] + 16u * hc) << 2;
      if (res_n) v += *reinterpret_cast<const f32x4*>(at_off(res_n, o));
      if (a.relu) {
        asm volatile("");
        v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
      }
      if (a.out_planes) {  // (uniform) the next layer's fp16 planes instead of float32: its staging is a copy
        asm volatile("");
        if (max_abs4(v) > F16_MAX) atomicOr(a.ovf, 1);
        if (ovalid[pt]) *reinterpret_cast<u32x4*>(at_off(out_n, o)) = planes_of(v);
      } else {
        if (ovalid[pt]) *reinterpret_cast<f32x4*>(at_off(out_n, o)) = v;
      }
    }
  }
  };
  init_acc(g0);

  for (int cc = 0; cc < NG * nch; ++cc) {
  const int gi = (NG > 1 && cc >= nch) ? 1 : 0;  // (NG <= 2)
  const int c = cc - gi * nch, g = g0 + gi;
  {
    // phases of a chunk: (kernel row, column slice) with one row resident, (column slice) with all three
    constexpr int NPH = (3 / WR) * NH;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
    {
      const int r0 = WR == 3 ? 0 : ph / NH, hf = WR == 3 ? ph : ph % NH;
      if (cc > 0 || ph > 0 || txi > txr * run) __syncthreads();  // every wave has read the previous phase's fragments
      // ---- registers -> LDS: the kernel row's weights (a straight copy) ----
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int item = tid + i * CT;
        if (item < W_WSUB) reinterpret_cast<u32x4*>(s_w)[item] = pre_w[i];
      }
      if (ph == 0) {
        // ---- the chunk's patch: BatchNorm + ReLU prologue, padding zeroed, split into bf16 planes ----
        {
          f32x4 psc = psc_r, psh = psh_r;
#pragma clang diagnostic pop
          int q8c = my_q8;
          if (LAUNDER) {
            q8c = tid & 7;
            asm volatile("" : "+v"(q8c));
          }
          if (LDSBN && a.in_scale) {
            psc = s_bn[c * 8 + q8c];
            psh = s_bn[(cin_g >> 2) + c * 8 + q8c];
          }
          int t8 = tid >> 3;
          if (LAUNDER) asm volatile("" : "+v"(t8));
          int st_e2 = st_e2_0;
          if (LAUNDER) st_e2 = (int)(__umul24((unsigned)(q8c >> 2), (unsigned)W_NPXP) + (unsigned)t8) * 4 + (q8c & 3);
          float vmax = 0.0f;  // (H) largest scaled magnitude this thread stages
#pragma unroll
          for (int i = 0; i < NP; ++i) {
            if (i * CT + CT <= NITEM || tid < NITEM - i * CT) {
#ifdef CPX_BF3W_FAKE_SPLIT  // experiment (scratch/patches/README.md): the staged pieces go to LDS as they are -- wrong results, the cost of BatchNorm + split gone
              {
                uint2* sp2 = reinterpret_cast<uint2*>(s_patch) + st_e2 + i * (64 * 4);
                sp2[(0 * 2 * W_NPXP) * 4] = make_uint2(pre_p[i][0], pre_p[i][1]);
                sp2[(1 * 2 * W_NPXP) * 4] = make_uint2(pre_p[i][2], pre_p[i][3]);
                if (PL == 3) sp2[(2 * 2 * W_NPXP) * 4] = make_uint2(pre_p[i][1], pre_p[i][2]);
                continue;
              }
#endif
              if (H && a.in_planes) {  // (uniform) the producer stored this layer's planes: [hi 0..3 | lo 0..3] per piece
                asm volatile("");
                u32x4 pv = pre_p[i];
                if (!interior) {
                  asm volatile("");
                  const int px = t8 + 64 * i;
                  const int py = (int)(__umul24((unsigned)px, 3641u) >> 16), pxx = __mul24(py, -W_PW) + px;
                  const bool inside = (unsigned)(iy0 + py) < (unsigned)a.H && (unsigned)(ix0 + pxx) < (unsigned)a.W;
#pragma unroll
                  for (int j = 0; j < 4; ++j) pv[j] = inside ? pv[j] : 0u;
                }
                uint2* sp2 = reinterpret_cast<uint2*>(s_patch) + st_e2 + i * (64 * 4);
                sp2[(0 * 2 * W_NPXP) * 4] = make_uint2(pv[0], pv[1]);
                sp2[(1 * 2 * W_NPXP) * 4] = make_uint2(pv[2], pv[3]);
                continue;
              }
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = __uint_as_float(pre_p[i][j]);
              // (the empty asm keeps these uniform branches branches: if-converted, every item pays the BatchNorm and the
              // padding selects -- 12 vector instructions -- whether the layer has a prologue / the tile a border or not)
              if (a.in_scale) {  // (H: scale and shift carry act_scale)
                asm volatile("");
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(v[j], psc[j], psh[j]), 0.0f);
              } else if (H) {
                asm volatile("");
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= a.act_scale;
              }
              if (H) vmax = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), vmax);
              if (!interior) {  // (uniform: most tiles skip the selects)
                asm volatile("");
                const int px = t8 + 64 * i;
                const int py = (int)(__umul24((unsigned)px, 3641u) >> 16), pxx = __mul24(py, -W_PW) + px;
                const bool inside = (unsigned)(iy0 + py) < (unsigned)a.H && (unsigned)(ix0 + pxx) < (unsigned)a.W;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
              }
              unsigned q0[2], q1[2], q2[2];
              if (PL == 3) {
                split_pair(v[0], v[1], q0[0], q1[0], q2[0]);
                split_pair(v[2], v[3], q0[1], q1[1], q2[1]);
              } else {
                split2<H>(v[0], v[1], q0[0], q1[0]);
                split2<H>(v[2], v[3], q0[1], q1[1]);
              }
              // channels 4 q8 .. 4 q8 + 3 of the chunk: quarter pair q8 >> 2, 8-byte slot q8 & 3 of the pixel's 32 bytes
              uint2* sp2 = reinterpret_cast<uint2*>(s_patch) + st_e2 + i * (64 * 4);
              sp2[(0 * 2 * W_NPXP) * 4] = make_uint2(q0[0], q0[1]);
              sp2[(1 * 2 * W_NPXP) * 4] = make_uint2(q1[0], q1[1]);
              if (PL == 3) sp2[(2 * 2 * W_NPXP) * 4] = make_uint2(q2[0], q2[1]);
            }
          }
          // out of fp16's range (padding pixels' clamped loads included: conservative): the three-plane kernel reruns the layer
          if (H && vmax > F16_MAX) atomicOr(a.ovf, 1);
        }
      }
      // ---- global -> registers for what comes next (in flight under this phase's products) ----
      // (the chunk after this one: the same group's next 32 channels, or the next group's first)
      const int gn = (c + 1 < nch) ? g : g + 1, cn = (c + 1 < nch) ? c + 1 : 0;
      if (ph + 1 < NPH) {
        BF3W_ISSUE_W(g, c, (WR == 3 ? 0 : (ph + 1) / NH), (WR == 3 ? ph + 1 : (ph + 1) % NH))
      } else if (cc + 1 < NG * nch) {
        BF3W_ISSUE_W(gn, cn, 0, 0)
      } else if (txi + 1 < tx_end) {
        BF3W_ISSUE_W(g0, 0, 0, 0)
      }
      if (ph == 0) {
        if (cc + 1 < NG * nch) {
          BF3W_ISSUE_P(gn, cn, ix0)
        } else if (txi + 1 < tx_end) {
          BF3W_ISSUE_P(g0, 0, ix0 + W_TW)
        }
      }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < WR; ++rr) {
      const int r = r0 + rr;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        u32x4 xv[2][PL], wv[2][PL];
#pragma unroll
        for (int p = 0; p < PL; ++p) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt)
            xv[pt][p] = __builtin_bit_cast(u32x4, s_patch[p * (4 * W_NPXP) + a_base + ((pt + r) * W_PW + kx) * 2]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) wv[ct][p] = __builtin_bit_cast(u32x4, s_w[rr * W_WROW + (p * 3 + kx) * 128 + b_base + 16 * ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            // smallest terms first, as conv_bf3_kernel (x = activation planes, w = weight planes)
            if (PL == 3) {
              acc[hf][ct][pt] = mfma16<false>(wv[ct][1], xv[pt][1], acc[hf][ct][pt]);
              acc[hf][ct][pt] = mfma16<false>(wv[ct][PL - 1], xv[pt][0], acc[hf][ct][pt]);
              acc[hf][ct][pt] = mfma16<false>(wv[ct][0], xv[pt][PL - 1], acc[hf][ct][pt]);
            }
            acc[hf][ct][pt] = mfma16<H>(wv[ct][1], xv[pt][0], acc[hf][ct][pt]);
            acc[hf][ct][pt] = mfma16<H>(wv[ct][0], xv[pt][1], acc[hf][ct][pt]);
            acc[hf][ct][pt] = mfma16<H>(wv[ct][0], xv[pt][0], acc[hf][ct][pt]);
          }
      }
      }  // resident kernel rows
    }
    }  // phases of the chunk
  }
  if (NG > 1 && c + 1 == nch && cc + 1 < NG * nch) {  // a group but the last is done: its stores, the next one's accumulators
    finish(g);
    init_acc(g + 1);
  }
  }  // chunks of the tile's groups
  finish(g0 + NG - 1);
  }  // tiles of the run
  BF3_TILE_LOOP_END
}
#undef BF3W_ISSUE_W
#undef BF3W_ISSUE_P

// ---------------------------------------------------------------------------------------------------------------------
// conv_block32_kernel: a whole residual block of stage 2 (two stride-1 3x3 convolutions, 32 channels per group in and
// out, wr_resnet.py wr_block) in ONE launch, CPX_CNN_MATH_FP16X2 only.
//
// Why: with three products per K step the stage-2 launches of conv_bf3w_kernel are bound by HBM, not by the matrix pipe
// (4.1 TB/s, profiles/r05_cnn_probe_handover.txt): per sample and block the first convolution reads `cur` and writes
// `mid` (2 x 6.55 MB), the second reads `mid` and `cur` (the residual) and writes the block's output (3 x 6.55 MB).
// Here `mid` never leaves the CU: per 16 x 16 output tile the workgroup stages the 20 x 20 input patch once, computes the
// 18 x 18 tile of `mid` the second convolution needs (1.27 x the first convolution's products: the halo is recomputed),
// applies the folded BatchNorm + ReLU and the second convolution's range scale to the accumulators and writes them to LDS
// as the second convolution's fp16 planes -- over the patch, which is dead by then -- and runs the second convolution from
// there.  HBM sees `cur` in (halo re-reads come from L2) and the output out: 13.1 MB instead of 32.8.
//
// LDS: [patch 20 x 20, later mid 18 x 18: 51,456 B][weights a: 36,864 B][weights b: 36,864 B][BatchNorm: 256 B] = 125 KB:
// one workgroup per CU, so the workgroup is persistent -- both weight images are loaded once, the tiles are walked with
// the next tile's patch in flight (registers) under the second convolution's products.  Layouts, fragment addressing and
// the order of the products are conv_bf3w_kernel's (PL = 2, H): the results are bit-identical to the two launches.
// First convolution: 324 pixels of mid = 21 groups of 16 (the last one 4 pixels), wave w takes groups w, w + 8, w + 16.
constexpr int B_PW = 20, B_NPX = 400;
constexpr int B_NPXP = 402;  // 402 * 32 B = 64 mod 128 (see W_NPXP)
constexpr int B_R0 = 2 * 2 * B_NPXP * 2;    // entries of the patch region (the mid image, 2 * 2 * W_NPXP * 2, fits inside)
constexpr int B_WIMG = 3 * 2 * 3 * 4 * 32;  // entries of one convolution's fp16 image of one group
// C8: the stage's FIRST block -- its first convolution takes 8 input channels per group (16 -> 64 channels), its second one adds
// the 1x1 shortcut of the block's input instead of a residual.  What changes: the patch is one 16-byte entry per pixel and
// plane ([plane][pixel], 8 channels); a K = 32 step of the first convolution pairs FOUR TAPS (lane quarter q of step t reads tap
// 4 t + q: three steps for the nine taps, the last three quarters of the third read a zero entry; weight image
// [t][plane][quarter][column], split_weights8h_kernel); the second convolution's accumulators start at zero and take the
// shortcut on v_mfma_f32_16x16x4_f32 before the epilogue, as conv_bf3w_kernel's do.
constexpr int B8_NPXP = B_NPX + 2;             // entries per plane of the 8-channel patch (a region of its own): 400 pixels, one zero entry, one spare
constexpr int B8_WIMG = 3 * 2 * 4 * 32;        // entries of the first convolution's fp16 image of one group (C8)
// C1 (with C8, round 6): the network's first convolution (conv1_1: 3x3, ONE input channel and 8 output channels per group, bias)
// is computed here too, while the patch is staged -- a thread takes a patch pixel, reads the nine raw input values around it and
// forms the pixel's 8 channels of the block's input with conv1_kernel's own multiply-adds in its order (the same float32 values),
// instead of loading them: the 64 B per pixel conv1_kernel wrote and this kernel read back (31 GB per step of 9,659 samples: a
// 5.3 ms launch and the patch loads of this one) never exist.  The 1x1 shortcut takes its operand -- the block's input at the
// output pixels -- from a float32 copy of the tile's centre in LDS (two buffers: the next tile is staged under this tile's
// second convolution).  A rerun after an fp16 overflow needs the tensor in memory: conv1_kernel is launched behind, guarded.
constexpr int B_C1W = 96;                 // floats: conv1's weights [9][8] and bias [8] of one group (+ padding)
constexpr int B_C1C = 256 * 8;            // floats per buffer of the centre copy: 16 x 16 pixels x 8 channels
template <bool C8, bool C1 = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_block32_kernel(ConvArgs a, ConvArgs b, const uint4* __restrict__ wa, const uint4* __restrict__ wb, TileDiv td) {
  static_assert(!C1 || C8, "conv1 is fused into the stage's first block only");
  if (*a.ovf != 0) return;  // (the block's guarded three-plane launches follow)
  constexpr int WA_IMG = C8 ? B8_WIMG : B_WIMG;
  constexpr int CING = C8 ? 8 : 32;  // input channels per group of the first convolution
  constexpr int CT = 512;
  constexpr int W_WROW = 2 * 3 * 4 * 32;
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  uint4* s_r0 = lds4;
  uint4* s_wa = lds4 + B_R0;
  uint4* s_wb = s_wa + WA_IMG;
  f32x4* s_bn = reinterpret_cast<f32x4*>(s_wb + B_WIMG);
  // the patch: the region mid takes over later -- or (C8: 13 KB, and a zero entry that must survive) a region of its own
  uint4* s_px = C8 ? reinterpret_cast<uint4*>(s_bn) + 16 : s_r0;
  float* s_c1w = reinterpret_cast<float*>(s_px + 2 * B8_NPXP);  // (C1) [9][8] weights, [8] bias
  float* s_c1c = s_c1w + B_C1W;                                  // (C1) [2][256][8] the block's input at the tile's 16 x 16 output pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int g = blockIdx.y;
  const int C = b.Cout;    // channels of mid and of the output (the block's input has a.Cin: the same unless C8)
  const int CI = a.Cin;
  for (int i = tid; i < B_WIMG; i += CT) {
    if (i < WA_IMG) reinterpret_cast<u32x4*>(s_wa)[i] = reinterpret_cast<const u32x4*>(wa)[g * WA_IMG + i];
    reinterpret_cast<u32x4*>(s_wb)[i] = reinterpret_cast<const u32x4*>(wb)[g * B_WIMG + i];
  }
  if (tid < CING) {  // relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k), exactly
    reinterpret_cast<float*>(s_bn)[tid] = a.in_scale[g * CING + tid] * a.act_scale;
    reinterpret_cast<float*>(s_bn)[CING + tid] = a.in_shift[g * CING + tid] * a.act_scale;
  }
  if (C8 && tid < 2) reinterpret_cast<u32x4*>(s_px)[tid * B8_NPXP + B_NPX] = u32x4{0u, 0u, 0u, 0u};  // the zero entry of either plane
  if (C1 && tid < 80) s_c1w[tid] = tid < 72 ? a.c1_w[g * 72 + tid] : a.c1_b[g * 8 + tid - 72];
  // tiles: every XCD walks its own contiguous eighth of the tile space (the workgroups of a launch go round-robin over
  // the XCDs), the workgroups of an XCD side by side in it: neighbours' halos meet in that XCD's L2
  const int per_xcd = (td.total + 7) >> 3;
  auto tile_of = [&](int t) { return (t & 7) * per_xcd + (t >> 3); };
  int n, oy0, ox0;
  auto decode = [&](int tile) {
    int qd = div_magic(tile, td.m_tx);
    ox0 = (tile - qd * td.tiles_x) * W_TW;
    tile = qd;
    qd = div_magic(tile, td.m_ty);
    oy0 = (tile - qd * td.tiles_y) * W_TH;
    n = qd;
  };
  constexpr int PIECES = CING / 4;           // 16-byte pieces (4 channels) of a patch pixel
  constexpr int NITEM = B_NPX * PIECES;
  constexpr int NP = (NITEM + CT - 1) / CT;  // 7 (the seventh for threads 0..127); C8: 2 (the second for threads 0..287)
  constexpr int PSH = C8 ? 1 : 3, PPI = CT >> PSH;  // item tid + 512 i = pixel (tid >> PSH) + PPI i, piece tid & (PIECES - 1)
  u32x4 pre_p[NP];
  // global -> registers: item i of the 20 x 20 patch of tile (n, oy0, ox0), clamped
  auto issue_item = [&](const int i) __attribute__((always_inline)) {
    int t8 = tid >> PSH;
    asm volatile("" : "+v"(t8));
    int q8 = tid & (PIECES - 1);
    asm volatile("" : "+v"(q8));
    const float* in_n = a.in + (size_t)n * a.H * a.W * CI;
    const unsigned coff = (unsigned)(g * CING + 4 * q8);
    const int px = min(t8 + PPI * i, B_NPX - 1);
    const int py = (int)(__umul24((unsigned)px, 3277u) >> 16), pxx = __mul24(py, -B_PW) + px;  // px / 20 for px < 400
    const int cy = min(max(oy0 - 2 + py, 0), a.H - 1), cx = min(max(ox0 - 2 + pxx, 0), a.W - 1);
    pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, a.W, CI) + coff) << 2));
  };
  auto issue_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_item(i);
  };
  // (C1) one thread per patch pixel: the nine raw input values of conv1's window around it (channel g), clamped addresses
  float c1_raw[C1 ? 9 : 1];
  auto c1_pixel = [&](int& py, int& pxx) __attribute__((always_inline)) {
    int t8 = tid;
    asm volatile("" : "+v"(t8));
    const int px = min(t8, B_NPX - 1);
    py = (int)(__umul24((unsigned)px, 3277u) >> 16);
    pxx = __mul24(py, -B_PW) + px;
  };
  auto issue_c1 = [&]() __attribute__((always_inline)) {
    if constexpr (C1) {
      int py, pxx;
      c1_pixel(py, pxx);
      const float* in_n = a.c1_in + (size_t)n * a.H * a.W * a.groups + g;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int cy = min(max(oy0 - 3 + py + ky, 0), a.H - 1), cx = min(max(ox0 - 3 + pxx + kx, 0), a.W - 1);
          c1_raw[ky * 3 + kx] = *at_off(in_n, (unsigned)((cy * a.W + cx) * a.groups) << 2);
        }
    }
  };
  int t = blockIdx.x;
  // (td.total >= 8 gridDim.x is not required: a workgroup whose first tile does not exist has none)
  while (t < 8 * per_xcd && tile_of(t) >= td.total) t += gridDim.x;
  if (t >= 8 * per_xcd) return;
  decode(tile_of(t));
  if constexpr (C1) issue_c1();
  else issue_patch();
  // what a thread needs of the channel parameters is the same for every tile: registers (one workgroup per CU: 256 to spend)
  const int ch_l = g * 32 + 4 * q;
  f32x4 os_a[2], ob_a[2], rs_b[2], os_b[2], ob_b[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int ch = ch_l + 16 * ct;
    os_a[ct] = *reinterpret_cast<const f32x4*>(a.w_unscale + ch) * (a.act_unscale * b.act_scale);  // (powers of two: exact)
    if (a.out_scale) os_a[ct] *= *reinterpret_cast<const f32x4*>(a.out_scale + ch);
    ob_a[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (a.out_shift) ob_a[ct] = *reinterpret_cast<const f32x4*>(a.out_shift + ch) * b.act_scale;
    rs_b[ct] = *reinterpret_cast<const f32x4*>(b.w_scale + ch) * b.act_scale;  // (the residual's scale; C8: the shortcut weights')
    os_b[ct] = *reinterpret_cast<const f32x4*>(b.w_unscale + ch) * b.act_unscale;
    ob_b[ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (b.out_shift) ob_b[ct] = *reinterpret_cast<const f32x4*>(b.out_shift + ch);
    if (C8 && b.sc_bias) ob_b[ct] += *reinterpret_cast<const f32x4*>(b.sc_bias + ch);
  }
  __syncthreads();  // weights and BatchNorm parameters are in LDS
  const f32x4 psc = s_bn[tid & (PIECES - 1)], psh = s_bn[PIECES + (tid & (PIECES - 1))];
  const f32x4 psc1 = s_bn[C1 ? 1 : 0], psh1 = s_bn[PIECES + (C1 ? 1 : 0)], psc0 = s_bn[0], psh0 = s_bn[PIECES];  // (C1: a thread converts both pieces)
  // out of fp16's range = a high plane that came out infinite.  Every staged value is >= 0 (ReLU), so fp16 bit patterns order
  // as unsigned halves: the running maximum of the high planes, two packed halves per instruction
  unsigned hmax = 0u;
  // registers -> the patch's fp16 planes, in place ([hi 0..1, hi 2..3, lo 0..1, lo 2..3]): BatchNorm + ReLU prologue,
  // padding zeroed.  (py0, px0): the patch's origin in the image
  auto convert_item = [&](const int i, const int py0, const int px0, const bool interior) __attribute__((always_inline)) {
    float v[4];
    {
      const f32x4 y = __builtin_elementwise_fma(__builtin_bit_cast(f32x4, pre_p[i]), psc, psh);  // (two v_pk_fma_f32)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(y[j], 0.0f);
    }
    if (!interior) {
      asm volatile("");
      const int px = (tid >> PSH) + PPI * i;
      const int py = (int)(__umul24((unsigned)px, 3277u) >> 16), pxx = __mul24(py, -B_PW) + px;
      const bool inside = (unsigned)(py0 + py) < (unsigned)a.H && (unsigned)(px0 + pxx) < (unsigned)a.W;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
    }
    unsigned h0, h1, l0, l1;
    split_pair2_h(v[0], v[1], h0, l0);
    split_pair2_h(v[2], v[3], h1, l1);
    hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    pre_p[i] = u32x4{h0, h1, l0, l1};
  };
  auto commit_patch = [&]() __attribute__((always_inline)) {  // converted registers -> LDS
    int t8 = tid >> PSH;
    asm volatile("" : "+v"(t8));
    int q8 = tid & (PIECES - 1);
    asm volatile("" : "+v"(q8));
    if constexpr (C1) {  // a thread holds its pixel's whole entries
      int tpx = tid;
      asm volatile("" : "+v"(tpx));
      if (tpx < B_NPX) {
        reinterpret_cast<u32x4*>(s_px)[tpx] = pre_p[0];
        reinterpret_cast<u32x4*>(s_px)[B8_NPXP + tpx] = pre_p[1];
      }
      return;
    }
    if (C8) {  // [plane][pixel] of 16-byte entries: piece 0 / 1 = the entry's low / high 8 bytes
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if (i * CT + CT <= NITEM || tid < NITEM - i * CT) {
          uint2* sp2 = reinterpret_cast<uint2*>(s_px) + (t8 + PPI * i) * 2 + q8;
          sp2[0] = make_uint2(pre_p[i][0], pre_p[i][1]);
          sp2[B8_NPXP * 2] = make_uint2(pre_p[i][2], pre_p[i][3]);
        }
      }
      return;
    }
    const int st_e2 = (int)(__umul24((unsigned)(q8 >> 2), (unsigned)B_NPXP) + (unsigned)t8) * 4 + (q8 & 3);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (i * CT + CT <= NITEM || tid < NITEM - i * CT) {
        uint2* sp2 = reinterpret_cast<uint2*>(s_r0) + st_e2 + i * (64 * 4);
        sp2[(0 * 2 * B_NPXP) * 4] = make_uint2(pre_p[i][0], pre_p[i][1]);
        sp2[(1 * 2 * B_NPXP) * 4] = make_uint2(pre_p[i][2], pre_p[i][3]);
      }
    }
  };
  // (C1) the pixel's 8 channels of the block's input from the nine raw values (conv1_kernel's multiply-adds, in its order, plus the
  // bias), their float32 copy for the shortcut if the pixel is one of the tile's outputs, then the prologue and the split of both
  // pieces: pre_p[0] / pre_p[1] = the pixel's entry of the high / low plane.  (py0, px0): the patch's origin in the image
  // In three parts (a kernel row each; the last one finishes the pixel): under the second convolution's products a part rides on one
  // tap -- the whole pixel in one lump (~150 vector instructions) stuck out from under a tap's twelve products.
  float acc8[C1 ? 8 : 1];
  auto convert_c1 = [&](const int part, const int py0, const int px0, const bool interior, const int buf) __attribute__((always_inline)) {
    if constexpr (C1) {
      int py, pxx;
      c1_pixel(py, pxx);
      if (part == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c) acc8[c] = 0.0f;
      }
#pragma unroll
      for (int tp = 3 * part; tp < 3 * part + 3; ++tp) {
        float v = c1_raw[tp];
        if (!interior) {  // conv1's own zero padding
          const int ky = tp / 3, kx = tp - 3 * ky;
          v = ((unsigned)(py0 - 1 + py + ky) < (unsigned)a.H && (unsigned)(px0 - 1 + pxx + kx) < (unsigned)a.W) ? v : 0.0f;
        }
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_c1w + tp * 8), w1 = *reinterpret_cast<const f32x4*>(s_c1w + tp * 8 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc8[c] = __fmaf_rn(v, w0[c], acc8[c]);
          acc8[4 + c] = __fmaf_rn(v, w1[c], acc8[4 + c]);
        }
      }
      if (part < 2) return;
      const f32x4 bs0 = *reinterpret_cast<const f32x4*>(s_c1w + 72), bs1 = *reinterpret_cast<const f32x4*>(s_c1w + 76);
      f32x4 x0, x1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        x0[c] = acc8[c] + bs0[c];
        x1[c] = acc8[4 + c] + bs1[c];
      }
      if ((unsigned)(py - 2) < 16u && (unsigned)(pxx - 2) < 16u && tid < B_NPX) {
        f32x4* cc = reinterpret_cast<f32x4*>(s_c1c + buf * B_C1C + ((py - 2) * 16 + (pxx - 2)) * 8);
        cc[0] = x0;
        cc[1] = x1;
      }
      f32x4 y0 = __builtin_elementwise_fma(x0, psc0, psh0), y1 = __builtin_elementwise_fma(x1, psc1, psh1);
      const bool inside = interior || ((unsigned)(py0 + py) < (unsigned)a.H && (unsigned)(px0 + pxx) < (unsigned)a.W);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        y0[j] = inside ? fmaxf(y0[j], 0.0f) : 0.0f;
        y1[j] = inside ? fmaxf(y1[j], 0.0f) : 0.0f;
      }
      unsigned h0, h1, h2, h3, l0, l1, l2, l3;
      split_pair2_h(y0[0], y0[1], h0, l0);
      split_pair2_h(y0[2], y0[3], h1, l1);
      split_pair2_h(y1[0], y1[1], h2, l2);
      split_pair2_h(y1[2], y1[3], h3, l3);
      hmax = pk_max_u16(pk_max_u16(pk_max_u16(pk_max_u16(hmax, h0), h1), h2), h3);
      pre_p[0] = u32x4{h0, h1, h2, h3};
      pre_p[1] = u32x4{l0, l1, l2, l3};
    }
  };
  // (C1: conv1's window reaches one pixel further than the patch)
  auto tile_interior = [&]() {
    constexpr int M = C1 ? 3 : 2;
    return oy0 - M >= 0 && oy0 + 16 + M <= a.H && ox0 - M >= 0 && ox0 + 16 + M <= a.W;
  };
  int cbuf = 0;  // (C1) which centre buffer the tile in hand reads
  {
    const bool interior = tile_interior();
    if constexpr (C1) {
#pragma unroll
      for (int part = 0; part < 3; ++part) convert_c1(part, oy0 - 2, ox0 - 2, interior, 0);
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) convert_item(i, oy0 - 2, ox0 - 2, interior);
    }
  }
#ifdef CPX_B32_STAMPS  // (experiment, scratch/build_conv_variant.sh: where a tile's time goes -- cycle stamps of waves 0 and 7 of one workgroup)
  long long st_last = clock64(), st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int st_tiles = 0;
  const bool st_on = blockIdx.x == 8 && blockIdx.y == 0 && (tid == 0 || tid == 448);
#define B32_STAMP(i_) if (st_on) { const long long now_ = clock64(); st_acc[i_] += now_ - st_last; st_last = now_; }
#else
#define B32_STAMP(i_)
#endif
  for (;;) {
    commit_patch();
    B32_STAMP(0)
    const int n_cur = n, oy_cur = oy0, ox_cur = ox0;
    const bool interior_cur = tile_interior();
    // ---- what goes in flight under the first convolution, in pieces BETWEEN its K steps (address arithmetic and load issue
    //      run beside the matrix pipe instead of in front of the barrier: cycle stamps put them at 1,200-2,200 cycles of a
    //      tile's 18,000 there): the second convolution's accumulators = the residual (the block's input at the tile), the
    //      next tile's coordinates, its patch item by item ----
    f32x4 acc[2][2];
    unsigned opix[2];
    bool ovalid[2];
    unsigned spix[2];  // (C8) element offsets of the lane's two pixels in the block's input, channel g * 8 + q: the shortcut's operand
    bool more = false, interior_next = false;
    auto piece_residual = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int oy = oy_cur + 2 * wave + pt, ox = ox_cur + i16;
        ovalid[pt] = oy < a.H && ox < a.W;
        opix[pt] = pix_off(min(oy, a.H - 1), min(ox, a.W - 1), a.W, C) + (unsigned)ch_l;
        spix[pt] = pix_off(min(oy, a.H - 1), min(ox, a.W - 1), a.W, CI) + (unsigned)(g * CING + q);
      }
      if (C8) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
      } else {
        const float* res_n = b.residual + (size_t)n_cur * a.H * a.W * C;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = *reinterpret_cast<const f32x4*>(at_off(res_n, (opix[pt] + 16u * ct) << 2));
      }
    };
    auto piece_next_tile = [&]() __attribute__((always_inline)) {
      t += gridDim.x;
      more = t < 8 * per_xcd && tile_of(t) < td.total;  // (a workgroup's tiles ascend within its XCD's eighth: the first missing one ends it)
      if (more) decode(tile_of(t));
      interior_next = tile_interior();
    };
    B32_STAMP(1)
    __syncthreads();
    B32_STAMP(2)
    // ---- first convolution: 21 groups of 16 mid pixels x two 16-column tiles.  Waves 0-3 take groups w, w + 8, w + 16; waves
    //      4-7 groups w, w + 8; the 21st group (4 valid pixels) is split by column tile between waves 4 and 5 -- waves w and
    //      w + 4 share a SIMD, so the SIMDs carry 11, 11, 10, 10 (group, tile) units (with three whole groups on wave 4
    //      SIMD 0 carried 12 and every tile waited ~2,000 cycles for it at the barrier) ----
    f32x4 acc1[3][2];
    int abase1[3];
    const bool third_full = wave < 4;                // (uniform)
    const bool third_half = wave == 4 || wave == 5;  // (uniform) one column tile of group 20: tile wave - 4
    const int g3 = third_full ? wave + 16 : 20;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int p = min(16 * (j < 2 ? wave + 8 * j : g3) + i16, W_NPX - 1);
      const int my = (int)(__umul24((unsigned)p, 3641u) >> 16), mx = p - my * W_PW;  // p / 18 for p < 324
      abase1[j] = C8 ? my * B_PW + mx : ((q >> 1) * B_NPXP + my * B_PW + mx) * 2 + (q & 1);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc1[j][ct] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    const int b_base = q * 32 + i16;
    if constexpr (C8) {
#pragma unroll
      for (int t4 = 0; t4 < 3; ++t4) {
        // this lane quarter's tap of the step: 4 t4 + q, at patch offset ky * 20 + kx; beyond the ninth tap: the zero entry
        const int tap = 4 * t4 + q;
        const int ky = (tap * 11) >> 5;  // tap / 3 for tap < 12
        const int toff = ky * B_PW + (tap - 3 * ky);
        u32x4 wv[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) wv[ct][p] = __builtin_bit_cast(u32x4, s_wa[(t4 * 2 + p) * 128 + b_base + 16 * ct]);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if (j < 2 || third_full || third_half) {
            const int e = tap < 9 ? abase1[j] + toff : B_NPX;
            u32x4 xv[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) xv[p] = __builtin_bit_cast(u32x4, s_px[p * B8_NPXP + e]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              if (j < 2 || third_full || ct == wave - 4) {
                acc1[j][ct] = mfma16<true>(wv[ct][1], xv[0], acc1[j][ct]);
                acc1[j][ct] = mfma16<true>(wv[ct][0], xv[1], acc1[j][ct]);
                acc1[j][ct] = mfma16<true>(wv[ct][0], xv[0], acc1[j][ct]);
              }
            }
          }
        }
        if (t4 == 0) {
          piece_residual();
          piece_next_tile();
        } else if (more) {
          if constexpr (C1) {
            if (t4 == 1) issue_c1();
          } else {
            issue_item(t4 - 1);
          }
        }
      }
      static_assert(!C8 || NP == 2, "the patch items of the 8-channel form ride on the second and third K step");
    } else {
    // (as in the second convolution: tap k + 1's fragments are requested before tap k's products)
    u32x4 wv[2][2][2], xv[2][3][2];  // [buffer][column tile | group][plane]
    const bool third_any = third_full || third_half;
    auto frags = [&](const int tp, const int bf) __attribute__((always_inline)) {
      const int r = tp / 3, kx = tp - 3 * r;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wv[bf][ct][p] = __builtin_bit_cast(u32x4, s_wa[r * W_WROW + (p * 3 + kx) * 128 + b_base + 16 * ct]);
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (j < 2 || third_any) xv[bf][j][p] = __builtin_bit_cast(u32x4, s_r0[p * (4 * B_NPXP) + abase1[j] + (r * B_PW + kx) * 2]);
      }
    };
    frags(0, 0);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int bf = tp & 1;
      if (tp + 1 < 9) frags(tp + 1, bf ^ 1);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < 2 || third_any) {
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            if (j < 2 || third_full || ct == wave - 4) {
              acc1[j][ct] = mfma16<true>(wv[bf][ct][1], xv[bf][j][0], acc1[j][ct]);
              acc1[j][ct] = mfma16<true>(wv[bf][ct][0], xv[bf][j][1], acc1[j][ct]);
              acc1[j][ct] = mfma16<true>(wv[bf][ct][0], xv[bf][j][0], acc1[j][ct]);
            }
          }
        }
      }
      if (tp == 0) piece_residual();
      else if (tp == 1) piece_next_tile();
      else if (more) issue_item(tp - 2);
    }
    static_assert(C8 || NP == 7, "the patch items ride on K steps 2..8 of the first convolution");
    }
    B32_STAMP(3)
    // every wave has read its patch fragments: the region becomes mid (C8: the patch has a region of its own -- a wave
    // writes its share of mid as soon as its own products are done)
    if (!C8) __syncthreads();
    B32_STAMP(4)
    // ---- mid = relu(acc * a_scale + a_shift), times the second convolution's range scale, as its fp16 planes in LDS;
    //      pixels outside the image are that convolution's zero padding ----
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const int c4 = 4 * ct + q;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int p = 16 * (j < 2 ? wave + 8 * j : g3) + i16;
        if (p < W_NPX && (j < 2 || third_full || (third_half && ct == wave - 4))) {
          f32x4 v = acc1[j][ct] * os_a[ct] + ob_a[ct];
          v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          u32x4 pv = planes_of(v);
          hmax = pk_max_u16(pk_max_u16(hmax, pv[0]), pv[1]);
          if (!interior_cur) {  // (uniform)
            asm volatile("");
            const int my = (int)(__umul24((unsigned)p, 3641u) >> 16), mx = p - my * W_PW;
            const bool inside = (unsigned)(oy_cur - 1 + my) < (unsigned)a.H && (unsigned)(ox_cur - 1 + mx) < (unsigned)a.W;
#pragma unroll
            for (int k = 0; k < 4; ++k) pv[k] = inside ? pv[k] : 0u;
          }
          uint2* sp2 = reinterpret_cast<uint2*>(s_r0) + ((c4 >> 2) * W_NPXP + p) * 4 + (c4 & 3);
          sp2[(0 * 2 * W_NPXP) * 4] = make_uint2(pv[0], pv[1]);
          sp2[(1 * 2 * W_NPXP) * 4] = make_uint2(pv[2], pv[3]);
        }
      }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        if (!C8) acc[ct][pt] *= rs_b[ct];  // the accumulators hold act_scale * w_scale[channel] times the sum: so must the residual
    B32_STAMP(5)
    __syncthreads();
    B32_STAMP(6)
    // ---- second convolution: conv_bf3w_kernel's loop on the mid image; between its taps the next tile's patch (landed
    //      during the first convolution) takes its prologue and split, one item per tap: vector work under the products ----
    {
      const int a_base = ((q >> 1) * W_NPXP + (2 * wave) * W_PW + i16) * 2 + (q & 1);
      // (the fragments of tap k + 1 are requested before the products of tap k: their LDS round trip runs under those
      //  products and the prologue work between the taps, not in front of the next tap's first product)
      u32x4 xv[2][2][2], wv[2][2][2];  // [buffer][pixel row | column tile][plane]
      auto frags = [&](const int tp, const int bf) __attribute__((always_inline)) {
        const int r = tp / 3, kx = tp - 3 * r;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) xv[bf][pt][p] = __builtin_bit_cast(u32x4, s_r0[p * (4 * W_NPXP) + a_base + ((pt + r) * W_PW + kx) * 2]);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) wv[bf][ct][p] = __builtin_bit_cast(u32x4, s_wb[r * W_WROW + (p * 3 + kx) * 128 + b_base + 16 * ct]);
        }
      };
      frags(0, 0);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int bf = tp & 1;
        if (tp + 1 < 9) frags(tp + 1, bf ^ 1);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) {
            acc[ct][pt] = mfma16<true>(wv[bf][ct][1], xv[bf][pt][0], acc[ct][pt]);
            acc[ct][pt] = mfma16<true>(wv[bf][ct][0], xv[bf][pt][1], acc[ct][pt]);
            acc[ct][pt] = mfma16<true>(wv[bf][ct][0], xv[bf][pt][0], acc[ct][pt]);
          }
#ifndef CPX_B32_NO_CONVERT
        if constexpr (C1) {
          if (tp < 3 && more) convert_c1(tp, oy0 - 2, ox0 - 2, interior_next, cbuf ^ 1);  // (in one lump on the first tap: 24,641 against 24,691 samples/s)
        } else {
          if (tp < NP && more) convert_item(tp, oy0 - 2, ox0 - 2, interior_next);
        }
#endif
      }
    }
    if constexpr (C8) {
      // ---- the block's 1x1 shortcut (conv_bf3w_kernel's): A = weights [column][k] times their column's scale, B = pixels ----
      const float* sc_n = b.sc_in + (size_t)n_cur * a.H * a.W * CI;
      const float* wsc = b.sc_w + ((size_t)g * CING + q) * 32 + i16;
      float ss[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) ss[ct] = b.w_scale[g * 32 + 16 * ct + i16] * b.act_scale;
#pragma unroll
      for (int k4 = 0; k4 < CING; k4 += 4) {
        float xs[2], ws[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          xs[pt] = C1 ? s_c1c[cbuf * B_C1C + ((2 * wave + pt) * 16 + i16) * 8 + q + k4] : *at_off(sc_n, (spix[pt] + (unsigned)k4) << 2);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) ws[ct] = wsc[(size_t)k4 * 32 + 16 * ct] * ss[ct];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int pt = 0; pt < 2; ++pt) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[ct], xs[pt], acc[ct][pt], 0, 0, 0);
      }
    }
    B32_STAMP(7)
    // ---- epilogue: unscale, bias, ReLU, one 16-byte store per accumulator tile ----
    {
      float* out_n = b.out + (size_t)n_cur * a.H * a.W * C;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
          f32x4 v = acc[ct][pt] * os_b[ct];
          v += ob_b[ct];
          v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
          if (ovalid[pt]) *reinterpret_cast<f32x4*>(at_off(out_n, (opix[pt] + 16u * ct) << 2)) = v;
        }
      }
    }
    B32_STAMP(8)
#ifdef CPX_B32_STAMPS
    ++st_tiles;
#endif
    if (!more) break;
    cbuf ^= 1;
    // every wave has read its mid fragments: the region takes the next patch (C8: the patch goes to its own region, and the
    // barrier behind the commit is passed only by waves that are through with this tile's mid)
    if (!C8) __syncthreads();
    B32_STAMP(9)
  }
#ifdef CPX_B32_STAMPS
  if (st_on)
    printf("b32 stamps C8 %d wave %d tiles %d: commit %lld issue %lld bar1 %lld phaseA %lld bar2 %lld midepi %lld bar3 %lld phaseB %lld outepi %lld bar4 %lld\n",
           (int)C8, wave, st_tiles, st_acc[0], st_acc[1], st_acc[2], st_acc[3], st_acc[4], st_acc[5], st_acc[6], st_acc[7], st_acc[8], st_acc[9]);
#endif
#undef B32_STAMP
  if ((hmax & 0xFFFFu) >= 0x7C00u || (hmax >> 16) >= 0x7C00u) atomicOr(a.ovf, 1);  // (infinity or NaN)
}

constexpr long long RERUN_GRID = 1024;  // workgroups along x of a guarded rerun (four per CU and group row)
template <int NTN, int NPXC, int PL, bool H = false, int CT = 256>
int launch_bf3flat_t(const ConvArgs& a, const uint4* wimg, hipStream_t s) {
  constexpr int BAND = CT / 2;
  const size_t lds = std::max(((size_t)2 * PL * NPXC + (size_t)18 * PL * 32 * NTN) * 16, (size_t)(CT / 64) * 32 * 32 * sizeof(float));
  static bool lds_ready[64], lds_ready_p[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3flat_kernel<NTN, NPXC, PL, H, false, CT>), lds_ready, 160 * 1024 - 1024)) return -1;
  TileDiv td;
  td.tiles_x = (a.Ho * a.Wo + BAND - 1) / BAND;
  td.tiles_y = 1;
  td.nsplit = (a.Cout / a.groups) / (32 * NTN);
  const long long blocks = (long long)td.tiles_x * a.N * td.nsplit;
  if (blocks >= (1 << 22) || td.tiles_x >= 4096) return -3;
  td.m_nsplit = (1ull << 42) / td.nsplit + 1;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) + 1;
  td.total = (int)blocks;
  if constexpr (PL == 3 && !H) if (a.guard != nullptr) {  // the guarded rerun of a fp16x2 layer: a small grid that walks the tiles (PERSIST)
    if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3flat_kernel<NTN, NPXC, PL, false, true, CT>), lds_ready_p, 160 * 1024 - 1024)) return -1;
    hipLaunchKernelGGL((conv_bf3flat_kernel<NTN, NPXC, PL, false, true, CT>), dim3((unsigned)std::min<long long>(blocks, RERUN_GRID), a.groups), dim3(CT), lds, s, a, wimg, td);
    return 0;
  }
  hipLaunchKernelGGL((conv_bf3flat_kernel<NTN, NPXC, PL, H, false, CT>), dim3((unsigned)blocks, a.groups), dim3(CT), lds, s, a, wimg, td);
  return 0;
}
// the flattened tiling applies to stride-1 SAME 3 x 3 layers whose staged rows fit the LDS cap and pays when the
// rectangular bands waste more than 8 % over it
static bool flat_pays(const ConvArgs& a, int TW, int TH, int npx_cap, int band = 128) {
  if (a.stride != 1 || a.ksize != 3 || a.pad_top != 1 || a.pad_left != 1 || a.H != a.Ho || a.W != a.Wo) return false;
  const int M = a.Ho * a.Wo;
  const int rows = (band - 1 + a.Wo - 1) / a.Wo + 1;
  if ((rows + 2) * (a.W + 2) > npx_cap) return false;
  const double rect = (double)M / ((double)((a.Wo + TW - 1) / TW) * TW * ((a.Ho + TH - 1) / TH) * TH);
  const double flat = (double)M / ((double)((M + band - 1) / band) * band);
  return flat > rect + 0.08;
}
#ifndef CPX_BF3FLAT_WIDE
#define CPX_BF3FLAT_WIDE 0  // fp16x2: 256-position bands (512 threads) where the staged rows fit 384 pixels -- measured SLOWER on stage 4 (36.5 vs 31.4 ms per 15 launches of 1,536 samples, profiles/r06_conv_rw_experiments.md): not shipped
#endif

// packed float32 weights [g][tap][cin_g][cout_g] -> bf16 plane image [g][chunk][3][9][2][cout_g] of 16-byte entries
// cin_g == 8: [g][3][5][2][cout_g], the entry of (step s, k half h) = the 8 channels of tap 2 s + h (zeros for tap 9)
__global__ __launch_bounds__(256) void split_weights8_kernel(const float* __restrict__ w, uint4* __restrict__ out, int groups,
                                                             int cout_g) {
  const size_t total = (size_t)groups * 5 * 2 * cout_g;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int col = (int)(r % cout_g);
    r /= cout_g;
    const int h = (int)(r & 1);
    r >>= 1;
    const int st = (int)(r % 5);
    const int g = (int)(r / 5);
    const int tap = 2 * st + h;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tap < 9 ? w[(((size_t)g * 9 + tap) * 8 + j) * cout_g + col] : 0.0f;
    uint4 p0, p1, p2;
    split_pair(v[0], v[1], p0.x, p1.x, p2.x);
    split_pair(v[2], v[3], p0.y, p1.y, p2.y);
    split_pair(v[4], v[5], p0.z, p1.z, p2.z);
    split_pair(v[6], v[7], p0.w, p1.w, p2.w);
    const size_t base = (size_t)g * 30 * cout_g;
    out[base + ((size_t)(0 * 5 + st) * 2 + h) * cout_g + col] = p0;
    out[base + ((size_t)(1 * 5 + st) * 2 + h) * cout_g + col] = p1;
    out[base + ((size_t)(2 * 5 + st) * 2 + h) * cout_g + col] = p2;
  }
}

// CPX_CNN_MATH_FP16X2: the power of two that brings an output channel's largest weight magnitude into [8, 16) -- the
// hi plane then is a normal fp16 with room to spare and the lo plane of every weight down to 2^-7 of the largest keeps
// its 11 bits -- and its inverse.  packed float32 weights [g][9][cin_g][cout_g]; one thread per output channel
__global__ __launch_bounds__(256) void weight_scales_kernel(const float* __restrict__ w, float* __restrict__ wscale,
                                                            float* __restrict__ wunscale, int groups, int rows, int cout_g) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= groups * cout_g) return;
  const int g = idx / cout_g, col = idx - g * cout_g;
  float m = 0.0f;
  for (int r = 0; r < rows; ++r) m = fmaxf(m, fabsf(w[((size_t)g * rows + r) * cout_g + col]));
  int e = 0;
  if (m > 0.0f && m < 3.0e38f) (void)frexpf(m, &e);  // m = f 2^e, f in [0.5, 1)
  else e = 4;                                          // all-zero (or non-finite) column: scale 1
  const int kw = min(max(4 - e, -100), 100);
  wscale[idx] = ldexpf(1.0f, kw);
  wunscale[idx] = ldexpf(1.0f, -kw);
}

// the 8-channel layer's fp16 image for conv_block32_kernel<true>: packed float32 weights [g][9][8][32] ->
// [g][step t][plane][quarter q][column] of 16-byte entries = the 8 channels of tap 4 t + q (zero beyond the ninth tap),
// times wscale[output channel]
__global__ __launch_bounds__(256) void split_weights8h_kernel(const float* __restrict__ w, uint4* __restrict__ out, int groups,
                                                              const float* __restrict__ wscale) {
  const size_t total = (size_t)groups * 3 * 4 * 32;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(idx & 31), qq = (int)((idx >> 5) & 3), t = (int)((idx >> 7) % 3), g = (int)(idx / 384);
    const int tap = 4 * t + qq;
    const float ws = wscale[g * 32 + col];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tap < 9 ? w[(((size_t)g * 9 + tap) * 8 + j) * 32 + col] * ws : 0.0f;
    uint4 p0, p1;
    split_pair2_h(v[0], v[1], p0.x, p1.x);
    split_pair2_h(v[2], v[3], p0.y, p1.y);
    split_pair2_h(v[4], v[5], p0.z, p1.z);
    split_pair2_h(v[6], v[7], p0.w, p1.w);
    const size_t base = (size_t)g * B8_WIMG;
    out[base + (size_t)(t * 2 + 0) * 128 + qq * 32 + col] = p0;
    out[base + (size_t)(t * 2 + 1) * 128 + qq * 32 + col] = p1;
  }
}

// planes = 2 (CPX_CNN_MATH_BF16X2): [g][chunk][2][9][2][cout_g], hi / lo rounded to nearest
// wscale != nullptr (CPX_CNN_MATH_FP16X2, planes == 2): fp16 planes of w * wscale[output channel] (a power of two)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, uint4* __restrict__ out, int groups,
                                                            int cin_g, int cout_g, int planes, const float* __restrict__ wscale) {
  const int nchunks = cin_g / KC;
  const size_t total = (size_t)groups * nchunks * 9 * 2 * cout_g;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int col = (int)(r % cout_g);
    r /= cout_g;
    const int h = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % 9);
    r /= 9;
    const int chunk = (int)(r % nchunks);
    const int g = (int)(r / nchunks);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = w[(((size_t)g * 9 + tap) * cin_g + chunk * KC + 8 * h + j) * cout_g + col];
    uint4 p0, p1, p2;
    if (wscale) {
      const float ws = wscale[g * cout_g + col];
      split_pair2_h(v[0] * ws, v[1] * ws, p0.x, p1.x);
      split_pair2_h(v[2] * ws, v[3] * ws, p0.y, p1.y);
      split_pair2_h(v[4] * ws, v[5] * ws, p0.z, p1.z);
      split_pair2_h(v[6] * ws, v[7] * ws, p0.w, p1.w);
    } else if (planes == 3) {
      split_pair(v[0], v[1], p0.x, p1.x, p2.x);
      split_pair(v[2], v[3], p0.y, p1.y, p2.y);
      split_pair(v[4], v[5], p0.z, p1.z, p2.z);
      split_pair(v[6], v[7], p0.w, p1.w, p2.w);
    } else {
      split_pair2(v[0], v[1], p0.x, p1.x);
      split_pair2(v[2], v[3], p0.y, p1.y);
      split_pair2(v[4], v[5], p0.z, p1.z);
      split_pair2(v[6], v[7], p0.w, p1.w);
    }
    const size_t base = ((size_t)g * nchunks + chunk) * (18 * planes) * cout_g;
    out[base + ((size_t)(0 * 9 + tap) * 2 + h) * cout_g + col] = p0;
    out[base + ((size_t)(1 * 9 + tap) * 2 + h) * cout_g + col] = p1;
    if (planes == 3) out[base + ((size_t)(2 * 9 + tap) * 2 + h) * cout_g + col] = p2;
  }
}

// packed float32 weights [g][tap][cin_g][cout_g] -> [g][chunk of 32][ky][plane][kx][quarter][cout_g] of 16-byte entries
// (conv_bf3w_kernel: a (chunk, ky) sub-chunk is 36 rows of cout_g entries)
// planes = 2: [g][chunk][ky][2][kx][quarter][cout_g], hi / lo rounded to nearest (a sub-chunk is 24 rows)
// wscale != nullptr (CPX_CNN_MATH_FP16X2, planes == 2): fp16 planes of w * wscale[output channel]
__global__ __launch_bounds__(256) void split_weights32_kernel(const float* __restrict__ w, uint4* __restrict__ out, int groups,
                                                              int cin_g, int cout_g, int planes, const float* __restrict__ wscale) {
  const int nch = cin_g / KW;
  const size_t total = (size_t)groups * nch * 9 * 4 * cout_g;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t r = idx;
    const int col = (int)(r % cout_g);
    r /= cout_g;
    const int qq = (int)(r & 3);
    r >>= 2;
    const int kx = (int)(r % 3);
    r /= 3;
    const int ky = (int)(r % 3);
    r /= 3;
    const int chunk = (int)(r % nch);
    const int g = (int)(r / nch);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = w[(((size_t)g * 9 + ky * 3 + kx) * cin_g + chunk * KW + 8 * qq + j) * cout_g + col];
    uint4 p0, p1, p2;
    if (wscale) {
      const float ws = wscale[g * cout_g + col];
      split_pair2_h(v[0] * ws, v[1] * ws, p0.x, p1.x);
      split_pair2_h(v[2] * ws, v[3] * ws, p0.y, p1.y);
      split_pair2_h(v[4] * ws, v[5] * ws, p0.z, p1.z);
      split_pair2_h(v[6] * ws, v[7] * ws, p0.w, p1.w);
    } else if (planes == 3) {
      split_pair(v[0], v[1], p0.x, p1.x, p2.x);
      split_pair(v[2], v[3], p0.y, p1.y, p2.y);
      split_pair(v[4], v[5], p0.z, p1.z, p2.z);
      split_pair(v[6], v[7], p0.w, p1.w, p2.w);
    } else {
      split_pair2(v[0], v[1], p0.x, p1.x);
      split_pair2(v[2], v[3], p0.y, p1.y);
      split_pair2(v[4], v[5], p0.z, p1.z);
      split_pair2(v[6], v[7], p0.w, p1.w);
    }
    const size_t base = (((size_t)g * nch + chunk) * 3 + ky) * (12 * planes) * cout_g;
    out[base + ((size_t)(0 * 3 + kx) * 4 + qq) * cout_g + col] = p0;
    out[base + ((size_t)(1 * 3 + kx) * 4 + qq) * cout_g + col] = p1;
    if (planes == 3) out[base + ((size_t)(2 * 3 + kx) * 4 + qq) * cout_g + col] = p2;
  }
}

template <int NH, int NG, int PL, bool H = false>
static int launch_bf3w_t(const ConvArgs& a, const uint4* wimg, hipStream_t s) {
  const size_t lds = (size_t)(w_patch_entries(PL) + w_wsub_entries(PL)) * 16 + (size_t)(a.Cin / a.groups) * 8;  // + BatchNorm scale / shift
  static bool lds_ready[64], lds_ready_p[64];
  // (NH = 2 carries 16 more accumulator registers: the BatchNorm parameters go to LDS there)
  constexpr bool WALK = CPX_BF3W_RUN > 1, LDSBN = CPX_BF3W_LDSBN != 0 || NH > 1 || PL == 2;
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3w_kernel<WALK, LDSBN, NH, NG, PL, H>), lds_ready, 160 * 1024 - 1024)) return -1;
  TileDiv td;
  const int tx = (a.Wo + W_TW - 1) / W_TW;
  // tiles per workgroup: the largest of CPX_BF3W_RUN .. 2 that divides the tiles of a row, else the whole row if it is short
  int run = 1;
  for (int r = CPX_BF3W_RUN; r >= 2; --r)  // (never entered at CPX_BF3W_RUN = 1)
    if (tx % r == 0) { run = r; break; }
  if (run == 1 && tx <= CPX_BF3W_RUN) run = tx;
  td.run = run;
  td.tiles_x = (tx + run - 1) / run;
  td.tiles_y = (a.Ho + W_TH - 1) / W_TH;
  td.nsplit = (a.Cout / a.groups) / (32 * NH);
  const long long blocks = (long long)td.tiles_x * td.tiles_y * a.N * td.nsplit;
  if (blocks >= (1 << 22) || td.tiles_x >= 4096 || td.tiles_y >= 4096) return -3;
  td.m_nsplit = (1ull << 42) / td.nsplit + 1;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) / td.tiles_y + 1;
  td.total = (int)blocks;
  if constexpr (PL == 3 && !H) if (a.guard != nullptr) {  // the guarded rerun of a fp16x2 layer: a small grid that walks the tiles (PERSIST)
    if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3w_kernel<WALK, LDSBN, NH, NG, PL, false, true>), lds_ready_p, 160 * 1024 - 1024)) return -1;
    hipLaunchKernelGGL((conv_bf3w_kernel<WALK, LDSBN, NH, NG, PL, false, true>), dim3((unsigned)std::min<long long>(blocks, RERUN_GRID), a.groups / NG), dim3(512), lds, s, a, wimg, td);
    return 0;
  }
  hipLaunchKernelGGL((conv_bf3w_kernel<WALK, LDSBN, NH, NG, PL, H>), dim3((unsigned)blocks, a.groups / NG), dim3(512), lds, s, a, wimg, td);
  return 0;
}
#ifndef CPX_BF3W_NH
#define CPX_BF3W_NH 2  // 32-column slices per workgroup where the group has 64 columns (1: one slice, two workgroups per tile)
#endif
#ifndef CPX_BF3W_NG
#define CPX_BF3W_NG 1  // groups a workgroup walks on its tile in the one-slice form (2: measured 6.5 % slower on stage 2, scratch/patches/README.md)
#endif
// the three-plane image of a layer, then (these layers only) the two-plane one
static size_t bf3w_image3_bytes(const ConvArgs& a) { return (size_t)a.groups * (a.Cin / a.groups / KW) * 3 * 36 * (a.Cout / a.groups) * 16; }
static size_t bf3w_image2_bytes(const ConvArgs& a) { return (size_t)a.groups * (a.Cin / a.groups / KW) * 3 * 24 * (a.Cout / a.groups) * 16; }
static int launch_bf3w(const ConvArgs& a, const uint4* wimg, hipStream_t s) {
  if (a.planes == 2 && a.half) {  // the fp16 image lies behind the two bf16 ones
    const uint4* wh = wimg + (bf3w_image3_bytes(a) + bf3w_image2_bytes(a)) / 16;
    // 64 -> 64 channels per group: weights in registers (cpx_cnn_rw.hip) -- but for the launch that carries a stage's 1x1
    // shortcut: that side product (strided float32 operands) stays with this kernel, measured faster than the shortcut as a
    // launch of its own + conv_rw64_kernel with its output as residual (profiles/r06_conv_rw_experiments.md)
    if (conv_rw_layer(a) && !a.in_planes && !a.out_planes && !a.sc_in) {
      const int rc = launch_conv_rw(a, wh, s);
      if (rc != -2) return rc;
    }
    if (CPX_BF3W_NH == 2 && a.Cout / a.groups == 64) return launch_bf3w_t<2, 1, 2, true>(a, wh, s);
    return launch_bf3w_t<1, 1, 2, true>(a, wh, s);
  }
  if (a.planes == 2) {
    const uint4* w2 = wimg + bf3w_image3_bytes(a) / 16;
    if (CPX_BF3W_NH == 2 && a.Cout / a.groups == 64) return launch_bf3w_t<2, 1, 2>(a, w2, s);
    return launch_bf3w_t<1, 1, 2>(a, w2, s);
  }
  if (CPX_BF3W_NH == 2 && a.Cout / a.groups == 64) return launch_bf3w_t<2, 1, 3>(a, wimg, s);
  constexpr int NG = (CPX_BF3W_NG == 2 && CPX_BF3W_RUN <= 1 && CPX_BF3W_LDSBN == 0) ? 2 : 1;
  if (NG == 2 && (a.groups & 1) == 0) return launch_bf3w_t<1, NG, 3>(a, wimg, s);
  return launch_bf3w_t<1, 1, 3>(a, wimg, s);
}

template <int NTN, int S, int NB, int TW, int CT, bool C8 = false, int PL = 3, bool H = false>
int launch_bf3_t(const ConvArgs& a, const uint4* wimg, hipStream_t s) {
  constexpr int TB = 128 / TW, TH = TB * NB;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
  size_t lds = ((size_t)(C8 ? 3 : 2 * PL) * PH * PW + (size_t)(C8 ? 30 : 18 * PL) * 32 * NTN) * 16;
  if (lds < (CT / 64) * 32 * 32 * sizeof(float)) lds = (CT / 64) * 32 * 32 * sizeof(float);
  static bool lds_ready[64], lds_ready_p[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3_kernel<NTN, S, NB, TW, CT, C8, PL, H>), lds_ready, 160 * 1024 - 1024)) return -1;
  TileDiv td;
  td.tiles_x = (a.Wo + TW - 1) / TW;
  td.tiles_y = (a.Ho + TH - 1) / TH;
  td.nsplit = (a.Cout / a.groups) / (32 * NTN);
  const long long blocks = (long long)td.tiles_x * td.tiles_y * a.N * td.nsplit;
  if (blocks >= (1 << 22) || td.tiles_x >= 4096 || td.tiles_y >= 4096) return -3;  // div_magic's range
  td.m_nsplit = (1ull << 42) / td.nsplit + 1;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) / td.tiles_y + 1;
  td.total = (int)blocks;
  if constexpr (PL == 3 && !H) if (a.guard != nullptr) {  // the guarded rerun of a fp16x2 layer: a small grid that walks the tiles (PERSIST)
    if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_bf3_kernel<NTN, S, NB, TW, CT, C8, PL, false, true>), lds_ready_p, 160 * 1024 - 1024)) return -1;
    hipLaunchKernelGGL((conv_bf3_kernel<NTN, S, NB, TW, CT, C8, PL, false, true>), dim3((unsigned)std::min<long long>(blocks, RERUN_GRID), a.groups), dim3(CT), lds, s, a, wimg, td);
    return 0;
  }
  hipLaunchKernelGGL((conv_bf3_kernel<NTN, S, NB, TW, CT, C8, PL, H>), dim3((unsigned)blocks, a.groups), dim3(CT), lds, s, a, wimg, td);
  return 0;
}

}  // namespace

// band width: with 32-pixel bands a wave's 32 pixels are one row, i.e. 32 consecutive 16-byte LDS entries -- the
// only A-fragment layout that is conflict-free in ds_read_b128's lane groups ({0-3,12-15,20-27}, ...); with 8 x 16
// bands lanes 12-13 and 26-27 (4-5 and 20-21) share bank slots and every A read takes two LDS passes.  Measured:
// no difference on the 160-wide maps (LDS is not the limit), and the 80-wide maps of stage 3 lose 17 % of a 3 x 32
// tiling to padding, so both stay at 16
#ifndef CPX_BF3_TW_S2
#define CPX_BF3_TW_S2 16
#endif
#ifndef CPX_BF3_TW_S3
#define CPX_BF3_TW_S3 16
#endif
#ifndef CPX_BF3_NTN_S3
#define CPX_BF3_NTN_S3 1
#endif
#ifndef CPX_BF3_NTN_S4
#define CPX_BF3_NTN_S4 2
#endif
#ifndef CPX_BF3_NTN_ST3
#define CPX_BF3_NTN_ST3 1
#endif
#ifndef CPX_BF3_NB_S2
#define CPX_BF3_NB_S2 2
#endif
#ifndef CPX_BF3_NB_S3
#define CPX_BF3_NB_S3 2
#endif
#ifndef CPX_BF3_NB_S4
#define CPX_BF3_NB_S4 1
#endif
#ifndef CPX_BF3_CT_S2
#define CPX_BF3_CT_S2 512
#endif
#ifndef CPX_BF3_CT_S3
#define CPX_BF3_CT_S3 512
#endif
#ifndef CPX_BF3_CT_S4
#define CPX_BF3_CT_S4 256
#endif

// (8 input channels per group: the tap-paired form, built for the 32 output channels per group the network has)
static bool bf3_c8(const ConvArgs& a) { return a.Cin / a.groups == 8 && a.Cout / a.groups == 32; }
// the strided first convolution of a stage (wr_resnet.py:27-30: stride = stage index): the same kernel with a strided
// patch -- the MFMA loop is unchanged, only the A-fragment addresses carry the stride.  S = 2 (res3b0_branch2a): 33 x 33
// staged pixels for 16 x 16 outputs and both 32-column tiles of a group in one workgroup (160 KB, one workgroup per CU):
// 123 TFLOP/s against 95 on the float32 MFMA.  S = 3 (res4b0_branch2a, -DCPX_BF3_STRIDED=3): 24 x 48 staged pixels for
// 8 x 16 outputs, 138 KB, four waves per CU and the patch split once per 32-column slice: 40 TFLOP/s against 78 on the
// float32 MFMA (windows of a stride-3 3 x 3 kernel do not overlap: nothing is reused) -- measured, not shipped.
#ifndef CPX_BF3_STRIDED
#define CPX_BF3_STRIDED 1
#endif
static bool bf3_strided(const ConvArgs& a) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  if (!CPX_BF3_STRIDED || a.ksize != 3 || cin_g < KC || (cin_g % KC) != 0) return false;
  return (a.stride == 2 && cout_g == 64) || (CPX_BF3_STRIDED >= 3 && a.stride == 3 && cout_g == 128);
}
// stride-1 layers with 32 or 64 channels per group (stages 2 and 3): the 16x16x32 form (conv_bf3w_kernel) and its
// weight image; a property of the layer shape alone, so that the image built at cpx_cnn_create is the one every
// launch of the layer reads (with or without a fused shortcut)
#ifndef CPX_BF3W
#define CPX_BF3W 1
#endif

static bool bf3w_layer(const ConvArgs& a) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  return CPX_BF3W && a.ksize == 3 && a.stride == 1 && cin_g >= KW && (cin_g % KW) == 0 && (cout_g == 32 || cout_g == 64) && cin_g <= 64;
}
// the strided first convolutions conv_rw_kernel takes in fp16x2 (cpx_cnn_rw.hip): the stride-2 one has this file's kernels for
// the other modes and the rerun; the stride-3 one has none here -- every launch of it that is not conv_rw_kernel's goes to the
// float32 kernel (launch_conv, which honours ConvArgs::guard), as all of them did before round 6
static bool rw_stride3(const ConvArgs& a) { return conv_rw_kind(a) == 3; }
static bool rw_stride2(const ConvArgs& a) { return conv_rw_kind(a) == 2; }
// [g][chunk of 32][ky][plane 2][kx][quarter][cout_g] of 16-byte entries (split_weights32_kernel)
static size_t rw_image_bytes(const ConvArgs& a) { return (size_t)a.groups * (a.Cin / a.groups / KW) * 3 * 24 * (a.Cout / a.groups) * 16; }
bool conv_bf3_supported(const ConvArgs& a) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  if (a.ksize == 3 && a.stride == 1 && bf3_c8(a)) return true;
  if (bf3_strided(a)) return true;
  if (rw_stride3(a)) return true;
  return a.ksize == 3 && a.stride == 1 && cin_g >= KC && (cin_g % KC) == 0 && (cout_g == 32 || cout_g == 64 || cout_g == 128);
}
// three-plane image of the layers conv_bf3_kernel / conv_bf3flat_kernel take: [g][chunk of 16][3][9][2][cout_g]
static size_t image3_bytes(const ConvArgs& a) { return (size_t)a.groups * (a.Cin / a.groups / KC) * 54 * (a.Cout / a.groups) * 16; }
// stride-1 layers with 128 columns per group (stage 4): the flattened kernel takes them when the map is small
// (flat_pays, a property of the launch), in either math mode
static bool flat_layer(const ConvArgs& a) {
  const int cin_g = a.Cin / a.groups;
  if (bf3_c8(a) || bf3w_layer(a) || a.ksize != 3 || cin_g < KC || (cin_g % KC) != 0) return false;
  // ... and the stride-2 first convolution of stage 3 (conv_bf3_kernel<2,2,2,16,512>)
  return (a.stride == 1 && a.Cout / a.groups == 128) || (a.stride == 2 && bf3_strided(a));
}
// the per-channel weight scales of the fp16 image and their inverses: 2 x Cout floats behind the images (16-byte aligned)
static size_t scales_bytes(const ConvArgs& a) { return ((size_t)2 * a.Cout * sizeof(float) + 15) / 16 * 16; }
// the 8-channel layer: its three-plane tap-paired image, then the fp16 tap-quad image of conv_block32_kernel<true> and its scales
static size_t c8_image3_bytes(const ConvArgs& a) { return (size_t)a.groups * 30 * (a.Cout / a.groups) * 16; }
static size_t c8_half_bytes(const ConvArgs& a) { return (size_t)a.groups * B8_WIMG * 16; }
size_t conv_bf3_weight_bytes(const ConvArgs& a) {
  if (rw_stride3(a)) return rw_image_bytes(a) + scales_bytes(a);
  if (rw_stride2(a) && flat_layer(a)) return image3_bytes(a) + 2 * (image3_bytes(a) / 3 * 2) + scales_bytes(a) + rw_image_bytes(a);
  if (bf3_c8(a)) return c8_image3_bytes(a) + c8_half_bytes(a) + scales_bytes(a);
  // the images of the three math modes one after the other: three bf16 planes, two bf16 planes, two fp16 planes, scales
  if (bf3w_layer(a)) return bf3w_image3_bytes(a) + 2 * bf3w_image2_bytes(a) + scales_bytes(a);
  return image3_bytes(a) + (flat_layer(a) ? 2 * (image3_bytes(a) / 3 * 2) + scales_bytes(a) : 0);
}
bool conv_bf3_two_planes(const ConvArgs& a) { return bf3w_layer(a) || flat_layer(a) || rw_stride3(a); }
// producer-side split (ConvArgs::out_planes / in_planes): the kernels whose epilogue can store the next layer's fp16
// planes -- conv_bf3w_kernel and conv_bf3_kernel, i.e. every split-operand launch but the flattened one -- and the one
// whose staging can take them (conv_bf3w_kernel in fp16x2).  `a` describes the launch (H, W, Ho, Wo filled in).
bool conv_bf3_can_store_planes(const ConvArgs& a) {
  if (!conv_bf3_supported(a) || (a.Cout / a.groups) % 4 != 0) return false;
  if (a.stride != 1) return a.stride == 2 && bf3_strided(a);
  if (a.Cout / a.groups == 128) return !flat_pays(a, 32, 4 * CPX_BF3_NB_S4, 256);
  return true;
}
bool conv_bf3_can_load_planes(const ConvArgs& a) { return bf3w_layer(a) && !conv_rw_layer(a); }  // (conv_rw64_kernel stages float32)
// where the fp16 image and its scales lie inside a two-plane layer's weight images
static size_t half_image_offset(const ConvArgs& a) {
  if (rw_stride3(a)) return 0;
  if (bf3_c8(a)) return c8_image3_bytes(a);
  return bf3w_layer(a) ? bf3w_image3_bytes(a) + bf3w_image2_bytes(a) : image3_bytes(a) + image3_bytes(a) / 3 * 2;
}
// the 32-channel-chunk fp16 image conv_rw_kernel reads of the stride-2 layer: behind this file's images and the scales
static size_t rw2_image_offset(const ConvArgs& a) { return image3_bytes(a) + 2 * (image3_bytes(a) / 3 * 2) + scales_bytes(a); }
static size_t scales_offset(const ConvArgs& a) {
  if (rw_stride3(a)) return rw_image_bytes(a);
  if (bf3_c8(a)) return c8_image3_bytes(a) + c8_half_bytes(a);
  return bf3w_layer(a) ? bf3w_image3_bytes(a) + 2 * bf3w_image2_bytes(a) : image3_bytes(a) + 2 * (image3_bytes(a) / 3 * 2);
}
void launch_split_weights(const ConvArgs& a, void* wimg, hipStream_t s) {
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  if (rw_stride3(a)) {
    const size_t total32 = (size_t)a.groups * (cin_g / KW) * 36 * cout_g;
    float* ws = reinterpret_cast<float*>(reinterpret_cast<char*>(wimg) + scales_offset(a));
    hipLaunchKernelGGL(weight_scales_kernel, dim3((unsigned)((a.Cout + 255) / 256)), dim3(256), 0, s, a.weights, ws, ws + a.Cout,
                       a.groups, 9 * cin_g, cout_g);
    hipLaunchKernelGGL(split_weights32_kernel, dim3((unsigned)((total32 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg), a.groups, cin_g, cout_g, 2, ws);
    return;
  }
  if (bf3_c8(a)) {
    const size_t total8 = (size_t)a.groups * 10 * cout_g;
    hipLaunchKernelGGL(split_weights8_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg), a.groups, cout_g);
    float* ws = reinterpret_cast<float*>(reinterpret_cast<char*>(wimg) + scales_offset(a));
    hipLaunchKernelGGL(weight_scales_kernel, dim3((unsigned)((a.Cout + 255) / 256)), dim3(256), 0, s, a.weights, ws, ws + a.Cout,
                       a.groups, 9 * cin_g, cout_g);
    hipLaunchKernelGGL(split_weights8h_kernel, dim3((unsigned)((a.groups * 384 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg) + half_image_offset(a) / 16, a.groups, ws);
    return;
  }
  if (bf3w_layer(a)) {
    const size_t total32 = (size_t)a.groups * (cin_g / KW) * 36 * cout_g;
    float* ws = reinterpret_cast<float*>(reinterpret_cast<char*>(wimg) + scales_offset(a));
    hipLaunchKernelGGL(weight_scales_kernel, dim3((unsigned)((a.Cout + 255) / 256)), dim3(256), 0, s, a.weights, ws, ws + a.Cout,
                       a.groups, 9 * cin_g, cout_g);
    hipLaunchKernelGGL(split_weights32_kernel, dim3((unsigned)((total32 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg), a.groups, cin_g, cout_g, 3, nullptr);
    hipLaunchKernelGGL(split_weights32_kernel, dim3((unsigned)((total32 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg) + bf3w_image3_bytes(a) / 16, a.groups, cin_g, cout_g, 2, nullptr);
    hipLaunchKernelGGL(split_weights32_kernel, dim3((unsigned)((total32 + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg) + half_image_offset(a) / 16, a.groups, cin_g, cout_g, 2, ws);
    return;
  }
  const size_t total = (size_t)a.groups * (cin_g / KC) * 18 * cout_g;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.weights,
                     reinterpret_cast<uint4*>(wimg), a.groups, cin_g, cout_g, 3, nullptr);
  if (flat_layer(a)) {  // the two-plane images of the layers the flattened kernel may take, behind the three-plane one
    float* ws = reinterpret_cast<float*>(reinterpret_cast<char*>(wimg) + scales_offset(a));
    hipLaunchKernelGGL(weight_scales_kernel, dim3((unsigned)((a.Cout + 255) / 256)), dim3(256), 0, s, a.weights, ws, ws + a.Cout,
                       a.groups, 9 * cin_g, cout_g);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg) + image3_bytes(a) / 16, a.groups, cin_g, cout_g, 2, nullptr);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a.weights,
                       reinterpret_cast<uint4*>(wimg) + half_image_offset(a) / 16, a.groups, cin_g, cout_g, 2, ws);
    if (rw_stride2(a)) {
      const size_t total32 = (size_t)a.groups * (cin_g / KW) * 36 * cout_g;
      hipLaunchKernelGGL(split_weights32_kernel, dim3((unsigned)((total32 + 255) / 256)), dim3(256), 0, s, a.weights,
                         reinterpret_cast<uint4*>(wimg) + rw2_image_offset(a) / 16, a.groups, cin_g, cout_g, 2, ws);
    }
  }
}
int launch_conv_bf3(const ConvArgs& a_in, const void* wimg, hipStream_t s) {
  ConvArgs a = a_in;
  const int cout_g = a.Cout / a.groups;
  if (a.stride == 3 && rw_stride3(a) && !a.half) return launch_conv(a, s);  // (the float32 kernel: also the guarded rerun)
  if (a.half) {  // CPX_CNN_MATH_FP16X2: the two-plane layers only, with the network's overflow word
    if (a.planes != 2 || !conv_bf3_two_planes(a) || a.ovf == nullptr) return -2;
    if (a.in_planes && !conv_bf3_can_load_planes(a)) return -2;
    a.w_scale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wimg) + scales_offset(a));
    a.w_unscale = a.w_scale + a.Cout;
  }
  // pixel offsets are formed with 24-bit multiplies (pix_off): a sample of 2^24 pixels or more is out of their range
  if ((long long)a.H * a.W >= (1 << 24) || (long long)a.Ho * a.Wo >= (1 << 24) || a.Cin >= (1 << 24) || a.Cout >= (1 << 24)) return -3;
  if (a.in_planes && !a.half) return -2;
  if (a.out_planes && (a.ovf == nullptr || !conv_bf3_can_store_planes(a) || a.residual != nullptr)) return -2;
  const uint4* w = reinterpret_cast<const uint4*>(wimg);
  if (bf3_c8(a)) return launch_bf3_t<1, 1, CPX_BF3_NB_S2, CPX_BF3_TW_S2, CPX_BF3_CT_S2, true>(a, w, s);
  if (a.stride == 2) {
    if (!bf3_strided(a)) return -2;
    if (a.planes == 2 && a.half && rw_stride2(a) && flat_layer(a) && !a.out_planes && !a.in_planes) {
      const int rc = launch_conv_rw(a, w + rw2_image_offset(a) / 16, s);
      if (rc != -2) return rc;
    }
    if (a.planes == 2 && a.half) return launch_bf3_t<2, 2, 2, 16, 512, false, 2, true>(a, w + half_image_offset(a) / 16, s);
    if (a.planes == 2 && flat_layer(a)) return launch_bf3_t<2, 2, 2, 16, 512, false, 2>(a, w + image3_bytes(a) / 16, s);
    return launch_bf3_t<2, 2, 2, 16, 512>(a, w, s);
  }
  if (a.stride == 3 && a.half && rw_stride3(a)) {
    const int rc = launch_conv_rw(a, w + half_image_offset(a) / 16, s);
    // (a form conv_rw_kernel does not take -- a residual on a strided layer: the float32 kernel computes it here, and its
    // guarded twin behind finds the overflow word clear)
    return rc == -2 ? launch_conv(a, s) : rc;
  }
  if (a.stride == 3) return bf3_strided(a) ? launch_bf3_t<CPX_BF3_NTN_ST3, 3, 1, 16, 256>(a, w, s) : -2;
  if (a.stride != 1) return -2;
  if (bf3w_layer(a)) {
    if (a.sc_in && ((a.sc_cin / a.groups) & 3)) return -2;  // the fused shortcut walks K in fours
    return launch_bf3w(a, w, s);
  }
  if (cout_g == 32) return launch_bf3_t<1, 1, CPX_BF3_NB_S2, CPX_BF3_TW_S2, CPX_BF3_CT_S2>(a, w, s);
  if (cout_g == 64) return launch_bf3_t<CPX_BF3_NTN_S3, 1, CPX_BF3_NB_S3, CPX_BF3_TW_S3, CPX_BF3_CT_S3>(a, w, s);
  // 256 staged pixels + two N tiles = 80 KB: two workgroups per CU.  (Wider maps -- 54 x 54 at frame size 64 -- would
  // need 384 staged pixels and then fit only one N tile per workgroup: measured slower than the rectangular bands,
  // 399 vs 371 ms, the patch being activated and split by four column slices instead of two.)
  if (cout_g == 128 && flat_pays(a, 32, 4 * CPX_BF3_NB_S4, 256)) {
    if (a.planes == 2 && a.half && CPX_BF3FLAT_WIDE && flat_pays(a, 32, 4 * CPX_BF3_NB_S4, 384, 256))
      return launch_bf3flat_t<2, 384, 2, true, 512>(a, w + half_image_offset(a) / 16, s);
    if (a.planes == 2 && a.half) return launch_bf3flat_t<2, 256, 2, true>(a, w + half_image_offset(a) / 16, s);
    if (a.planes == 2 && flat_layer(a)) return launch_bf3flat_t<2, 256, 2>(a, w + image3_bytes(a) / 16, s);
    return launch_bf3flat_t<2, 256, 3>(a, w, s);
  }
  if (cout_g == 128) {
    if (a.planes == 2 && a.half)
      return launch_bf3_t<CPX_BF3_NTN_S4, 1, CPX_BF3_NB_S4, 32, CPX_BF3_CT_S4, false, 2, true>(a, w + half_image_offset(a) / 16, s);
    if (a.planes == 2 && flat_layer(a))
      return launch_bf3_t<CPX_BF3_NTN_S4, 1, CPX_BF3_NB_S4, 32, CPX_BF3_CT_S4, false, 2>(a, w + image3_bytes(a) / 16, s);
    return launch_bf3_t<CPX_BF3_NTN_S4, 1, CPX_BF3_NB_S4, 32, CPX_BF3_CT_S4>(a, w, s);
  }
  return -2;
}

// a stage-2 residual block in one launch (conv_block32_kernel): `a` the first convolution (BatchNorm prologue, folded
// BatchNorm + ReLU epilogue), `b` the second (bias, residual = the block's input, ReLU); fp16x2 with both act_scale set
bool conv_block32_supported(const ConvArgs& a, const ConvArgs& b) {
  const auto plain = [](const ConvArgs& c) {
    return c.ksize == 3 && c.stride == 1 && c.Cout / c.groups == 32 && c.relu && c.pad_top == 1 && c.pad_left == 1 && !c.out_planes &&
           !c.in_planes;
  };
  if (!(plain(a) && plain(b) && a.in_scale && a.in_shift && !a.residual && !a.sc_in && !b.in_scale && !b.out_scale &&
        a.Cout == b.Cin && b.Cin == b.Cout && a.groups == b.groups && a.H == b.H && a.W == b.W && a.N == b.N && a.ovf != nullptr))
    return false;
  if (a.Cin / a.groups == 8)  // the stage's first block: the second convolution adds the 1x1 shortcut of the block's input
    return b.residual == nullptr && b.sc_in == a.in && b.sc_w != nullptr && b.sc_cin == a.Cin && b.sc_stride == 1 && b.sc_H == a.H &&
           b.sc_W == a.W;
  return a.Cin == a.Cout && b.residual == a.in && b.sc_in == nullptr;
}
int launch_conv_block32(const ConvArgs& a_in, const ConvArgs& b_in, const void* wimg_a, const void* wimg_b, hipStream_t s) {
  ConvArgs a = a_in, b = b_in;
  if (!conv_block32_supported(a, b)) return -2;
  if ((long long)a.H * a.W >= (1 << 24)) return -3;
  a.w_scale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wimg_a) + scales_offset(a));
  a.w_unscale = a.w_scale + a.Cout;
  b.w_scale = reinterpret_cast<const float*>(reinterpret_cast<const char*>(wimg_b) + scales_offset(b));
  b.w_unscale = b.w_scale + b.Cout;
  const uint4* wa = reinterpret_cast<const uint4*>(wimg_a) + half_image_offset(a) / 16;
  const uint4* wb = reinterpret_cast<const uint4*>(wimg_b) + half_image_offset(b) / 16;
  TileDiv td{};
  td.tiles_x = (a.W + W_TW - 1) / W_TW;
  td.tiles_y = (a.H + W_TH - 1) / W_TH;
  td.nsplit = 1;
  const long long tiles = (long long)td.tiles_x * td.tiles_y * a.N;
  if (tiles >= (1 << 22) - 8 || td.tiles_x >= 4096 || td.tiles_y >= 4096) return -3;
  td.m_nsplit = (1ull << 42) + 1;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) / td.tiles_y + 1;
  td.total = (int)tiles;
  const bool c8 = a.Cin / a.groups == 8;
  // blocks past the stage's first: the two convolutions on different waves (cpx_cnn_blk.hip) unless CPX_BLOCK32_SPLIT=0
  static const bool split_roles = [] {
    const char* e = std::getenv("CPX_BLOCK32_SPLIT");
    return e == nullptr || std::atoi(e) != 0;
  }();
  if (!c8 && split_roles) return launch_conv_block32s(a, b, wa, wb, s);
  // conv1 computed inside (ConvArgs::c1_in): the network's first convolution must be the shape the kernel restates
  const bool c1 = c8 && a.c1_in != nullptr;
  if (c1 && !(a.groups == 2 && a.c1_w != nullptr && a.c1_b != nullptr)) return -2;
  const size_t lds = c8 ? (size_t)(B_R0 + B8_WIMG + B_WIMG) * 16 + 256 + (size_t)2 * B8_NPXP * 16 + (c1 ? (size_t)(B_C1W + 2 * B_C1C) * 4 : 0)
                        : (size_t)(B_R0 + 2 * B_WIMG) * 16 + 256;
  static bool lds_ready[64], lds_ready8[64], lds_ready1[64];
  if (c1 ? !cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_block32_kernel<true, true>), lds_ready1, 160 * 1024 - 1024)
      : c8 ? !cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_block32_kernel<true>), lds_ready8, 160 * 1024 - 1024)
           : !cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_block32_kernel<false>), lds_ready, 160 * 1024 - 1024))
    return -1;
  // one workgroup per CU (125 KB of LDS), shared among the groups; a multiple of eight per group so that blockIdx.x & 7 is the XCD
  static int grid_x[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (grid_x[dev] == 0) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    grid_x[dev] = cus;
  }
  int gx = std::max(8, grid_x[dev] / a.groups / 8 * 8);
  if (const char* e = std::getenv("CPX_BLOCK32_GRID")) gx = std::max(8, std::atoi(e) / 8 * 8);
  gx = (int)std::min<long long>(gx, (tiles + 7) / 8 * 8);
  if (c1) hipLaunchKernelGGL((conv_block32_kernel<true, true>), dim3((unsigned)gx, a.groups), dim3(512), lds, s, a, b, wa, wb, td);
  else if (c8) hipLaunchKernelGGL(conv_block32_kernel<true>, dim3((unsigned)gx, a.groups), dim3(512), lds, s, a, b, wa, wb, td);
  else hipLaunchKernelGGL(conv_block32_kernel<false>, dim3((unsigned)gx, a.groups), dim3(512), lds, s, a, b, wa, wb, td);
  return 0;
}

}  // namespace cpx
