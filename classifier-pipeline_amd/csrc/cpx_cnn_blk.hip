// cpx_cnn_blk.hip -- a stage-2 residual block in one launch, its two convolutions on DIFFERENT WAVES (round 6).
//
// conv_block32_kernel (cpx_cnn_bf3.hip) keeps `mid` in LDS but runs a tile's phases one after the other on all eight waves:
// patch commit, first convolution, mid epilogue, second convolution, output epilogue, four barriers -- the matrix pipe is busy
// 49 % of the cycles (profiles/r06_conv_sq_counters.json), 31 % of a tile's time is spent in the phases that issue no product
// (profiles/r05_block32_experiments.md, cycle stamps), and a second workgroup that would fill them does not fit: patch / mid
// 51 KB + both weight images 74 KB = 125 KB of LDS.
//
// Here the workgroup is a two-stage pipeline over tiles.  Waves 0-3 ("A") compute the FIRST convolution of tile k + 1 while
// waves 4-7 ("B") compute the SECOND convolution of tile k; wave w and wave w + 4 share a SIMD, so every SIMD always has two
// independent product streams, and a step has two barriers:
//   phase 1   A: products of conv a (patch P -> accumulators)        B: products of conv b (mid M[k & 1] -> accumulators)
//             both: the global loads of tile k + 2's patch are issued between the products; B: tile k's residual rows
//   phase 2   A: mid epilogue (folded BatchNorm + ReLU, fp16 split) -> M[(k + 1) & 1]
//             B: output epilogue (unscale, bias, residual, ReLU, 16-byte stores)
//             both: tile k + 2's patch takes its prologue + split and lands in P (free: A has read it)
// Each wave holds ONE convolution's weights for ITS 16 output channels in registers (9 taps x 2 planes x 16 bytes = 72
// VGPRs, the conv_rw_kernel idea), so the weight images leave LDS: P (51.5 KB) + two mid buffers (2 x 41.7 KB) = 135 KB.
// Fragments: a wave owns whole pixel rows, so the row read for tap row 0 of output row o is tap row 1 of o - 1 and tap row 2
// of o - 2 (cpx_cnn_rw.hip): A reads 11 patch rows per kx for 9 mid rows (+ the two extra mid columns as three 16-pixel edge
// groups without reuse), B 10 mid rows per kx for 8 output rows.
// Arithmetic: conv_block32_kernel's (fp16x2, the same planes and scales); the taps are summed kx-major, another float32
// order of the same terms -- logits within 1e-5 of the two-launch form (tests/test_cnn_gpu.py).
// Reference semantics: /root/reference/src/ml_tools/resnet/wr_resnet.py:49-98 (wr_block).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "cpx_kernels.h"

namespace cpx {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
__device__ __forceinline__ const T* at_off(const T* base, unsigned bytes) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}
template <typename T>
__device__ __forceinline__ T* at_off(T* base, unsigned bytes) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + bytes);
}
__device__ __forceinline__ unsigned pix_off(int y, int x, int W, int C) {
  return __umul24(__umul24((unsigned)y, (unsigned)W) + (unsigned)x, (unsigned)C);
}
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
__device__ __forceinline__ void split_h(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 v = {(_Float16)a, (_Float16)b};
  hi = __builtin_bit_cast(unsigned, v);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(hi));
}
__device__ __forceinline__ f32x4 mfma_h(u32x4 w, u32x4 x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pk_max_u16(unsigned x, unsigned y) {
  unsigned r;
  asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ int med3(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
  return r;
}
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
struct BlkTiles {
  unsigned long long m_tx, m_ty;  // floor(2^42 / d) + 1
  int tiles_x, tiles_y, total;
};
__device__ __forceinline__ int div_magic(int n, unsigned long long m) { return (int)(((unsigned long long)n * m) >> 42); }

constexpr int K_T = 16;                          // output tile: 16 x 16
constexpr int K_PP = 20, K_PNPX = 400;           // the staged input patch: 20 x 20
constexpr int K_PNPXP = 402;                     // pixels per (plane, quarter pair) region, x 32 B = 64 mod 128 (cpx_cnn_bf3.hip: B_NPXP)
constexpr int K_MP = 18;                         // mid: 18 x 18 (324 pixels)
constexpr int K_MNPXP = 326;
constexpr int K_PBUF = 2 * 2 * K_PNPXP * 2;      // 16-byte entries: [plane][quarter pair][pixel][2]
constexpr int K_MBUF = 2 * 2 * K_MNPXP * 2;
constexpr int K_CT = 512;
constexpr int K_NP = 7;                          // staging items per thread: 400 pixels x 8 pieces = 3,200 = 6 x 512 + 128
constexpr int K_PPI = K_CT / 8;                  // patch pixels per staging round
constexpr int K_WROW = 2 * 3 * 4 * 32;           // entries of one kernel row of a group's fp16 image: [plane][kx][quarter][32]
constexpr int K_WIMG = 3 * K_WROW;
// staged items: convert (BatchNorm, padding, split) between the products of phase 1 and only store in phase 2 (1), or do both in
// phase 2 (0).  Measured (scratch/cnn_probe.py 1536, six launches): 57.7 ms with 1, 50.9 ms with 0 -- the conversion's 245 vector
// instructions per wave compete with the two product streams for the SIMD's issue slots, and a second tile in the registers pushes
// the residual rows behind the products
// 2: only the B waves convert early (the shorter product stream has the slack), the A waves in phase 2 -- phase 2 then has one
// wave per SIMD issuing vector instructions instead of two
#ifndef CPX_BLK_CONVERT_EARLY
#define CPX_BLK_CONVERT_EARLY 0
#endif
// the residual rows of B's tile: 1 = requested into 32 registers ahead of the products and added in the epilogue (fits only
// without early conversion), 0 = requested into the accumulators at the end of the previous step
#ifndef CPX_BLK_RES_REGS
#define CPX_BLK_RES_REGS 0
#endif
constexpr bool RES_REGS = CPX_BLK_RES_REGS != 0;
constexpr bool EARLY_A = CPX_BLK_CONVERT_EARLY == 1, EARLY_B = CPX_BLK_CONVERT_EARLY >= 1;
constexpr bool K_EARLY = EARLY_B;  // (some role keeps a second tile in its registers: tile k + 3 is looked up)
constexpr size_t K_LDS = (size_t)(K_PBUF + 2 * K_MBUF) * 16;

__global__ __launch_bounds__(K_CT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_block32s_kernel(ConvArgs a, ConvArgs b, const uint4* __restrict__ wa, const uint4* __restrict__ wb, BlkTiles td) {
  if (*a.ovf != 0) return;  // (the block's guarded three-plane launches follow)
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  uint4* s_P = lds4;
  uint4* s_M = lds4 + K_PBUF;  // two buffers of K_MBUF entries
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const bool is_a = wave < 4;                       // (wave-uniform) first or second convolution
  const int ct = wave & 1, hh = (wave >> 1) & 1;    // the wave's 16 of the group's 32 output channels; upper / lower half of the rows
  const int g = blockIdx.y;
  const int C = b.Cout, H = a.H, W = a.W;

  // ---- the wave's weights: ONE convolution's fp16 image [ky][plane][kx][quarter][32] of its group, its 16 columns ----
  u32x4 Wr[9][2];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(is_a ? wa : wb) + (size_t)g * K_WIMG + q * 32 + ct * 16 + i16;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < 2; ++p) Wr[ky * 3 + kx][p] = wg[ky * K_WROW + (p * 3 + kx) * 128];
  }
  // per-channel epilogue parameters of the wave's role (channels g 32 + ct 16 + 4 q ..)
  const int ch_l = g * 32 + ct * 16 + 4 * q;
  f32x4 os, ob, rs = {1.0f, 1.0f, 1.0f, 1.0f};
  if (is_a) {  // mid = relu(acc os + ob): the folded BatchNorm, times the second convolution's range scale
    os = *reinterpret_cast<const f32x4*>(a.w_unscale + ch_l) * (a.act_unscale * b.act_scale);  // (powers of two: exact)
    if (a.out_scale) os *= *reinterpret_cast<const f32x4*>(a.out_scale + ch_l);
    ob = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (a.out_shift) ob = *reinterpret_cast<const f32x4*>(a.out_shift + ch_l) * b.act_scale;
  } else {
    rs = *reinterpret_cast<const f32x4*>(b.w_scale + ch_l) * b.act_scale;  // (the residual's scale: what the sums carry)
    os = *reinterpret_cast<const f32x4*>(b.w_unscale + ch_l) * b.act_unscale;
    ob = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (b.out_shift) ob = *reinterpret_cast<const f32x4*>(b.out_shift + ch_l);
  }

  // ---- tiles: every XCD walks its own contiguous eighth of the tile space (conv_block32_kernel) ----
  const int per_xcd = (td.total + 7) >> 3;
  auto tile_of = [&](int t) { return (t & 7) * per_xcd + (t >> 3); };
  // tile j of this workgroup -> (n, oy, ox); false: it has fewer (its tiles ascend within its XCD's eighth)
  auto get_tile = [&](int j, int& n_, int& oy_, int& ox_) {
    const int t = (int)blockIdx.x + j * (int)gridDim.x;
    if (t >= 8 * per_xcd) return false;
    int tile = tile_of(t);
    if (tile >= td.total) return false;
    int qd = div_magic(tile, td.m_tx);
    ox_ = (tile - qd * td.tiles_x) * K_T;
    tile = qd;
    qd = div_magic(tile, td.m_ty);
    oy_ = (tile - qd * td.tiles_y) * K_T;
    n_ = qd;
    return true;
  };

  // ---- staging: item i of a thread = one 16-byte piece (4 channels) of patch pixel (tid >> 3) + 64 i, piece tid & 7 ----
  u32x4 pre_p[K_NP];
  const int q8 = tid & 7;
  int ipos[K_NP];  // patch row << 8 | column
#pragma unroll
  for (int i = 0; i < K_NP; ++i) {
    const int px = min((tid >> 3) + K_PPI * i, K_PNPX - 1);
    const int py = px / K_PP;
    ipos[i] = (py << 8) | (px - py * K_PP);
  }
  const unsigned st_base = (unsigned)((((q8 >> 2) * K_PNPXP + (tid >> 3)) * 4 + (q8 & 3)) * 8);
  f32x4 psc, psh;  // relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k), exactly
  psc = *reinterpret_cast<const f32x4*>(a.in_scale + g * 32 + 4 * q8) * a.act_scale;
  psh = *reinterpret_cast<const f32x4*>(a.in_shift + g * 32 + 4 * q8) * a.act_scale;
  const unsigned coff0 = (unsigned)(g * 32 + 4 * q8);
  const int Hm1 = H - 1, Wm1 = W - 1;
  auto issue_item = [&](const int i, const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const float* in_n = a.in + (size_t)n_ * H * W * a.Cin;  // (uniform)
    const int iy = oy_ - 2 + (ipos[i] >> 8), ix = ox_ - 2 + (ipos[i] & 0xFF);
    const int cy = med3(iy, 0, Hm1), cx = med3(ix, 0, Wm1);  // a clamped address is always loaded; padding is zeroed at commit
    pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, W, a.Cin) + coff0) << 2));
  };
  unsigned hmax = 0u;  // out of fp16's range = a high plane that came out infinite (every staged value is >= 0: ReLU)
  // a staged item's way into P has a register half and an LDS half.  convert: BatchNorm + ReLU prologue, zero padding, fp16
  // split -- registers only, so it rides between the products of phase 1 (of the step AFTER the one that requested the
  // item: two tiles are in the registers, one raw, one as planes); store: the two 8-byte LDS stores, the only part that has
  // to wait for P (phase 2)
  u32x4 pl[K_EARLY ? K_NP : 1];
  auto convert_item = [&](const int i, const int oy_, const int ox_, const bool to_pl) __attribute__((always_inline)) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(__uint_as_float(pre_p[i][j]), psc[j], psh[j]), 0.0f);
    const int iy = oy_ - 2 + (ipos[i] >> 8), ix = ox_ - 2 + (ipos[i] & 0xFF);
    const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;  // zero padding, as TensorFlow pads the activated tensor
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
    unsigned h0, h1, l0, l1;
    split_h(v[0], v[1], h0, l0);
    split_h(v[2], v[3], h1, l1);
    hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    if (to_pl) pl[K_EARLY ? i : 0] = u32x4{h0, h1, l0, l1};
    else pre_p[i] = u32x4{h0, h1, l0, l1};  // (in place: the store follows at once)
  };
  auto store_item = [&](const int i, const bool from_pl) __attribute__((always_inline)) {
    if (i == K_NP - 1 && tid >= K_PNPX * 8 - (K_NP - 1) * K_CT) return;
    u32x4 v;  // (two loads under a branch, not one load of a selected address: the arrays must stay registers)
    if (from_pl) v = pl[K_EARLY ? i : 0];
    else v = pre_p[i];
    unsigned char* sp = reinterpret_cast<unsigned char*>(s_P) + st_base;
    *reinterpret_cast<uint2*>(sp + i * (K_PPI * 32)) = make_uint2(v[0], v[1]);
    *reinterpret_cast<uint2*>(sp + 2 * K_PNPXP * 32 + i * (K_PPI * 32)) = make_uint2(v[2], v[3]);
  };

  // ---- accumulators (both roles use the same registers): A 9 mid rows + up to 2 edge groups, B 8 output rows ----
  f32x4 acc[11];
  constexpr int AHEAD = 3, RING = 4;

  // A: first convolution of the tile whose patch is in P.  Rows: mid rows 9 hh .. 9 hh + 8 (16 columns each) from patch rows
  // 9 hh .. 9 hh + 10; edge groups: mid columns 16, 17 of all 18 rows = 36 pixels in groups of 16 (e -> row e >> 1, column
  // 16 + (e & 1)); the half hh = 1 takes groups 0 and 1, hh = 0 the third (4 pixels).  `between(s)`: staging hooks, step s of 33 + 9
  auto conv_a = [&](auto&& between) __attribute__((always_inline)) {
    const uint4* pb = s_P + ((q >> 1) * K_PNPXP + (9 * hh) * K_PP + i16) * 2 + (q & 1);
#pragma unroll
    for (int o = 0; o < 11; ++o) acc[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 xh[RING], xl[RING];
    auto frag = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 11, i = s - 11 * kx, bf = s % RING;
      xh[bf] = __builtin_bit_cast(u32x4, pb[(i * K_PP + kx) * 2]);
      xl[bf] = __builtin_bit_cast(u32x4, pb[4 * K_PNPXP + (i * K_PP + kx) * 2]);
    };
    static_for<0, AHEAD>(frag);
    static_for<0, 33>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 11, i = s - 11 * kx, bf = s % RING;
      if constexpr (s + AHEAD < 33) frag(std::integral_constant<int, s + AHEAD>{});
      static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value, o = i - r;
          if constexpr (o >= 0 && o < 9) {
            if constexpr (pr == 0) acc[o] = mfma_h(Wr[r * 3 + kx][1], xh[bf], acc[o]);
            if constexpr (pr == 1) acc[o] = mfma_h(Wr[r * 3 + kx][0], xl[bf], acc[o]);
            if constexpr (pr == 2) acc[o] = mfma_h(Wr[r * 3 + kx][0], xh[bf], acc[o]);
          }
        });
      });
      between(sc);
      __builtin_amdgcn_sched_barrier(0);  // (pins the fragment reads three steps ahead of their products: cpx_cnn_rw.hip)
    });
    // the edge groups: per-lane pixel, no reuse across taps.  (hh == 0: one group, 32 + i16; hh == 1: two)
    int eb[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = min((hh ? 16 * k : 32) + i16, 35);
      eb[k] = ((q >> 1) * K_PNPXP + (e >> 1) * K_PP + 16 + (e & 1)) * 2 + (q & 1);
    }
    static_for<0, 9>([&](auto tc) __attribute__((always_inline)) {
      constexpr int tp = decltype(tc)::value, kx = tp / 3, ky = tp - 3 * kx;  // (kx-major, as the rows)
      u32x4 eh[2], el[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 0 || hh) {  // (uniform)
          eh[k] = __builtin_bit_cast(u32x4, s_P[eb[k] + (ky * K_PP + kx) * 2]);
          el[k] = __builtin_bit_cast(u32x4, s_P[4 * K_PNPXP + eb[k] + (ky * K_PP + kx) * 2]);
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (k == 0 || hh) {
          acc[9 + k] = mfma_h(Wr[ky * 3 + kx][1], eh[k], acc[9 + k]);
          acc[9 + k] = mfma_h(Wr[ky * 3 + kx][0], el[k], acc[9 + k]);
          acc[9 + k] = mfma_h(Wr[ky * 3 + kx][0], eh[k], acc[9 + k]);
        }
      }
      between(std::integral_constant<int, 33 + tp>{});
    });
  };
  // A, phase 2: mid = relu(acc os + ob) as the second convolution's fp16 planes in M[par]; pixels outside the image are
  // that convolution's zero padding.  (oy, ox): the tile's output origin; mid pixel (r, c) is image pixel (oy - 1 + r, ox - 1 + c)
  auto mid_put = [&](const int par, const int oy_, const int ox_, const f32x4 accv, const int mrow, const int mcol, const bool valid) __attribute__((always_inline)) {
    unsigned char* mb = reinterpret_cast<unsigned char*>(s_M + par * K_MBUF);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(accv[j], os[j], ob[j]), 0.0f);
    const bool inside = (unsigned)(oy_ - 1 + mrow) < (unsigned)H && (unsigned)(ox_ - 1 + mcol) < (unsigned)W;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
    unsigned h0, h1, l0, l1;
    split_h(v[0], v[1], h0, l0);
    split_h(v[2], v[3], h1, l1);
    hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    // channels ct 16 + 4 q .. of the 32: piece 4 ct + q -> quarter pair ct, 8-byte slot q of the pixel's 32 bytes
    const unsigned off = (unsigned)(((ct * K_MNPXP + mrow * K_MP + mcol) * 4 + q) * 8);
    if (valid) {
      *reinterpret_cast<uint2*>(mb + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(mb + off + 2 * K_MNPXP * 32) = make_uint2(l0, l1);
    }
  };
  // (a mid row is complete once its last tap column has passed: row o after step 24 + o of the 33 -- its epilogue rides on the
  // steps behind that one, under the remaining products, instead of behind all of them)
  auto mid_row = [&](const int par, const int oy_, const int ox_, const int o) __attribute__((always_inline)) {
    mid_put(par, oy_, ox_, acc[o], 9 * hh + o, i16, true);
  };
  auto mid_edges = [&](const int par, const int oy_, const int ox_) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 0 || hh) {
        const int e = (hh ? 16 * k : 32) + i16;
        mid_put(par, oy_, ox_, acc[9 + k], min(e, 35) >> 1, 16 + (e & 1), e < 36);
      }
    }
  };
  // B: second convolution of the tile whose mid is in M[par]: output rows 8 hh .. 8 hh + 7 from mid rows 8 hh .. 8 hh + 9
  auto conv_b = [&](const int par, auto&& between) __attribute__((always_inline)) {
    const uint4* mbp = s_M + par * K_MBUF + ((q >> 1) * K_MNPXP + (8 * hh) * K_MP + i16) * 2 + (q & 1);
    // the accumulators hold the tile's residual rows (requested at the end of the previous step: preload_res): they take the
    // scale of the sums, act_scale * w_scale[channel], and the products go on top -- no registers beside the accumulators
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = RES_REGS ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[o] * rs;
    u32x4 xh[RING], xl[RING];
    auto frag = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 10, i = s - 10 * kx, bf = s % RING;
      xh[bf] = __builtin_bit_cast(u32x4, mbp[(i * K_MP + kx) * 2]);
      xl[bf] = __builtin_bit_cast(u32x4, mbp[4 * K_MNPXP + (i * K_MP + kx) * 2]);
    };
    static_for<0, AHEAD>(frag);
    static_for<0, 30>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 10, i = s - 10 * kx, bf = s % RING;
      if constexpr (s + AHEAD < 30) frag(std::integral_constant<int, s + AHEAD>{});
      static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value, o = i - r;
          if constexpr (o >= 0 && o < 8) {
            if constexpr (pr == 0) acc[o] = mfma_h(Wr[r * 3 + kx][1], xh[bf], acc[o]);
            if constexpr (pr == 1) acc[o] = mfma_h(Wr[r * 3 + kx][0], xl[bf], acc[o]);
            if constexpr (pr == 2) acc[o] = mfma_h(Wr[r * 3 + kx][0], xh[bf], acc[o]);
          }
        });
      });
      between(sc);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  f32x4 rres[RES_REGS ? 8 : 1];
  auto preload_res = [&](const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const float* res_n = b.residual + (size_t)n_ * H * W * C;  // (uniform)
    const int ox = min(ox_ + i16, Wm1);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int oy = min(oy_ + 8 * hh + o, Hm1);
      const f32x4 v = *reinterpret_cast<const f32x4*>(at_off(res_n, (pix_off(oy, ox, W, C) + (unsigned)ch_l) << 2));
      if constexpr (RES_REGS) rres[o] = v;
      else acc[o] = v;
    }
  };
  // B: out = relu(acc os + ob (+ residual)), one 16-byte store per row; a row is complete after step 22 + o of the 30
  auto out_row = [&](const int n_, const int oy_, const int ox_, const int o) __attribute__((always_inline)) {
    float* out_n = b.out + (size_t)n_ * H * W * C;
    const int ox = min(ox_ + i16, Wm1);
    const int oy = oy_ + 8 * hh + o;
    const unsigned off = (pix_off(min(oy, Hm1), ox, W, C) + (unsigned)ch_l) << 2;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __fmaf_rn(acc[o][j], os[j], ob[j]);
    if constexpr (RES_REGS) v += rres[o];
    v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
    if (ox_ + i16 < W && oy < H) *reinterpret_cast<f32x4*>(at_off(out_n, off)) = v;
  };

  // ---- the pipeline.  Step k: A on tile k + 1, B on tile k; the staging converts tile k + 2 and requests tile k + 3 in phase
  //      1, stores tile k + 2 in phase 2; k = -1 fills it ----
  int n0 = 0, oy0 = 0, ox0 = 0, n1 = 0, oy1 = 0, ox1 = 0, n2 = 0, oy2 = 0, ox2 = 0, n3 = 0, oy3 = 0, ox3 = 0;
  bool v0 = false;                               // tile k     (B)
  bool v1 = get_tile(0, n1, oy1, ox1);           // tile k + 1 (A)
  if (!v1) return;                               // (the whole workgroup: no tile at all)
  bool v2 = get_tile(1, n2, oy2, ox2);           // tile k + 2 (converted, stored)
  bool v3 = K_EARLY && v2 && get_tile(2, n3, oy3, ox3);  // tile k + 3 (K_EARLY: requested a step ahead of its conversion)
  // the first tile's patch goes through the registers with nothing beside it; the second one's requests follow
#pragma unroll
  for (int i = 0; i < K_NP; ++i) issue_item(i, n1, oy1, ox1);
#pragma unroll
  for (int i = 0; i < K_NP; ++i) {
    convert_item(i, oy1, ox1, false);
    store_item(i, false);
  }
  // a role that converts a step ahead has the second tile's requests in flight from the start.  (Role-dependent staging is
  // written as two constant-argument copies under `if (is_a)`: with a run-time flag the optimiser merges the accesses of
  // pre_p / pl into one load of a selected address and both arrays go to scratch)
  {
    const int nq = v2 ? n2 : n1, oyq = v2 ? oy2 : oy1, oxq = v2 ? ox2 : ox1;
    if (is_a) {
      if constexpr (EARLY_A) {
#pragma unroll
        for (int i = 0; i < K_NP; ++i) issue_item(i, nq, oyq, oxq);
      }
    } else {
      if constexpr (EARLY_B) {
#pragma unroll
        for (int i = 0; i < K_NP; ++i) issue_item(i, nq, oyq, oxq);
      }
    }
  }
  __syncthreads();
  for (int k = -1;; ++k) {
    // ---- phase 1: products, and each role's own epilogue right behind them (A writes M[(k + 1) & 1] while B still reads
    //      M[k & 1]; B's stores touch no LDS): vector work under the other role's products ----
    // a tile that does not exist is replaced by the last one that does (its requests and planes are never stored)
    const int oyc = v2 ? oy2 : oy1, oxc = v2 ? ox2 : ox1;
    const int nr = v3 ? n3 : (v2 ? n2 : n1), oyr = v3 ? oy3 : (v2 ? oy2 : oy1), oxr = v3 ? ox3 : (v2 ? ox2 : ox1);
    auto stage = [&](const int i, const bool early_) __attribute__((always_inline)) {
      if (early_) {
        convert_item(i, oyc, oxc, true);
        issue_item(i, nr, oyr, oxr);
      } else {  // (the request of tile k + 2; its conversion and store follow in phase 2)
        issue_item(i, v2 ? n2 : n1, oyc, oxc);
      }
    };
    if (is_a) {
      if (v1) {
        conv_a([&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          if constexpr (s % 5 == 2 && s / 5 < K_NP) stage(s / 5, EARLY_A);
          if constexpr (s >= 25 && s <= 32) mid_row((k + 1) & 1, oy1, ox1, s - 25);  // (one step behind the row's last product)
        });
        mid_row((k + 1) & 1, oy1, ox1, 8);
        mid_edges((k + 1) & 1, oy1, ox1);
      } else {
#pragma unroll
        for (int i = 0; i < K_NP; ++i) stage(i, EARLY_A);
      }
    } else {
      if (v0) {
        if constexpr (RES_REGS) preload_res(n0, oy0, ox0);  // (ahead of the products, added behind them)
        conv_b(k & 1, [&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          if constexpr (s % 4 == 1 && s / 4 < K_NP) stage(s / 4, EARLY_B);
          if constexpr (s >= 23 && s <= 29) out_row(n0, oy0, ox0, s - 23);
        });
        out_row(n0, oy0, ox0, 7);
      } else {
#pragma unroll
        for (int i = 0; i < K_NP; ++i) stage(i, EARLY_B);
      }
      // the NEXT tile's residual rows (= the block's input at its output pixels) into the accumulators, now free: in flight
      // across the barriers and phase 2
      if constexpr (!RES_REGS) {
        if (v1) preload_res(n1, oy1, ox1);
      }
    }
    __syncthreads();
    // ---- phase 2: tile k + 2's planes -> P (free: A has read tile k + 1's) ----
    if (v2) {  // (uniform)
#pragma unroll
      for (int i = 0; i < K_NP; ++i) {
        if (is_a) {
          if constexpr (!EARLY_A) convert_item(i, oy2, ox2, false);
          store_item(i, EARLY_A);
        } else {
          if constexpr (!EARLY_B) convert_item(i, oy2, ox2, false);
          store_item(i, EARLY_B);
        }
      }
    }
    if (!v1) break;  // (uniform) tile k was the workgroup's last
    __syncthreads();
    n0 = n1; oy0 = oy1; ox0 = ox1; v0 = v1;
    n1 = n2; oy1 = oy2; ox1 = ox2; v1 = v2;
    if constexpr (K_EARLY) {
      n2 = n3; oy2 = oy3; ox2 = ox3; v2 = v3;
      v3 = v2 && get_tile(k + 4, n3, oy3, ox3);  // (the next step is k + 1: it requests tile k + 4)
    } else {
      v2 = v1 && get_tile(k + 3, n2, oy2, ox2);  // (the next step is k + 1: its staging tile is k + 3)
    }
  }
  if ((hmax & 0xFFFFu) >= 0x7C00u || (hmax >> 16) >= 0x7C00u) atomicOr(a.ovf, 1);  // (infinity or NaN: out of fp16's range)
}


// ---------------------------------------------------------------------------------------------------------------------
// conv_block32p_kernel: the same two-role pipeline with ONE barrier per step.  conv_block32s_kernel's second phase (the next
// patch's prologue + stores into the single patch buffer, behind a barrier of its own) leaves the matrix pipe idle for ~2.5 k
// of a tile's 14.4 k cycles.  With 16 x 8 output tiles both the patch (12 x 20 pixels: 33 KB) and mid (10 x 18: 23 KB) fit
// twice (113 KB), so the staging of tile k + 2 writes P[k & 1] WHILE A reads P[(k + 1) & 1] -- and it is the B waves alone
// that stage (108 products per step against A's 162: the shorter stream has the slack): load requests a step ahead, prologue +
// split + store between B's products.  A: mid rows 5 hh .. 5 hh + 4 and one edge group (mid columns 16, 17 of the 10 rows = 20
// pixels: 16 + 4); B: output rows 4 hh .. 4 hh + 3.  Halo: 1.875 input pixels staged per output pixel instead of 1.56.
constexpr int Q_R = 8;                           // output rows of a tile
constexpr int Q_PP = 20, Q_PNPX = 12 * 20;       // patch 12 x 20
constexpr int Q_NP = 8;                          // staging items per B thread: 240 pixels x 8 pieces = 1,920 = 7.5 x 256
constexpr int Q_PNPXP = Q_NP * 32 + 2;           // 258: the patch + the idle threads' slots; x 32 B = 64 mod 128
constexpr int Q_MP = 18;                         // mid 10 x 18 = 180 pixels
constexpr int Q_MNPXP = 182;
constexpr int Q_PBUF = 2 * 2 * Q_PNPXP * 2, Q_MBUF = 2 * 2 * Q_MNPXP * 2;
constexpr size_t Q_LDS = (size_t)(2 * Q_PBUF + 2 * Q_MBUF) * 16;

__global__ __launch_bounds__(K_CT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_block32p_kernel(ConvArgs a, ConvArgs b, const uint4* __restrict__ wa, const uint4* __restrict__ wb, BlkTiles td) {
  if (*a.ovf != 0) return;  // (the block's guarded three-plane launches follow)
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  uint4* s_P = lds4;                 // two buffers of Q_PBUF entries
  uint4* s_M = lds4 + 2 * Q_PBUF;    // two of Q_MBUF
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const bool is_a = wave < 4;
  const int ct = wave & 1, hh = (wave >> 1) & 1;
  const int g = blockIdx.y;
  const int C = b.Cout, H = a.H, W = a.W;
  u32x4 Wr[9][2];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(is_a ? wa : wb) + (size_t)g * K_WIMG + q * 32 + ct * 16 + i16;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < 2; ++p) Wr[ky * 3 + kx][p] = wg[ky * K_WROW + (p * 3 + kx) * 128];
  }
  const int ch_l = g * 32 + ct * 16 + 4 * q;
  f32x4 os, ob, rs = {1.0f, 1.0f, 1.0f, 1.0f};
  if (is_a) {
    os = *reinterpret_cast<const f32x4*>(a.w_unscale + ch_l) * (a.act_unscale * b.act_scale);  // (powers of two: exact)
    if (a.out_scale) os *= *reinterpret_cast<const f32x4*>(a.out_scale + ch_l);
    ob = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (a.out_shift) ob = *reinterpret_cast<const f32x4*>(a.out_shift + ch_l) * b.act_scale;
  } else {
    rs = *reinterpret_cast<const f32x4*>(b.w_scale + ch_l) * b.act_scale;
    os = *reinterpret_cast<const f32x4*>(b.w_unscale + ch_l) * b.act_unscale;
    ob = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (b.out_shift) ob = *reinterpret_cast<const f32x4*>(b.out_shift + ch_l);
  }
  const int per_xcd = (td.total + 7) >> 3;
  auto tile_of = [&](int t) { return (t & 7) * per_xcd + (t >> 3); };
  auto get_tile = [&](int j, int& n_, int& oy_, int& ox_) {
    const int t = (int)blockIdx.x + j * (int)gridDim.x;
    if (t >= 8 * per_xcd) return false;
    int tile = tile_of(t);
    if (tile >= td.total) return false;
    int qd = div_magic(tile, td.m_tx);
    ox_ = (tile - qd * td.tiles_x) * K_T;
    tile = qd;
    qd = div_magic(tile, td.m_ty);
    oy_ = (tile - qd * td.tiles_y) * Q_R;
    n_ = qd;
    return true;
  };
  // ---- staging (B threads): item i = one 16-byte piece of patch pixel (tb >> 3) + 32 i, piece tb & 7 ----
  const int tb = tid & 255;
  u32x4 pre_p[Q_NP];
  const int q8 = tb & 7;
  int ipos[Q_NP];
#pragma unroll
  for (int i = 0; i < Q_NP; ++i) {
    const int px = min((tb >> 3) + 32 * i, Q_PNPX - 1);
    const int py = px / Q_PP;
    ipos[i] = (py << 8) | (px - py * Q_PP);
  }
  const unsigned st_base = (unsigned)((((q8 >> 2) * Q_PNPXP + (tb >> 3)) * 4 + (q8 & 3)) * 8);
  const f32x4 psc = *reinterpret_cast<const f32x4*>(a.in_scale + g * 32 + 4 * q8) * a.act_scale;
  const f32x4 psh = *reinterpret_cast<const f32x4*>(a.in_shift + g * 32 + 4 * q8) * a.act_scale;
  const unsigned coff0 = (unsigned)(g * 32 + 4 * q8);
  const int Hm1 = H - 1, Wm1 = W - 1;
  auto issue_item = [&](const int i, const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const float* in_n = a.in + (size_t)n_ * H * W * a.Cin;
    const int iy = oy_ - 2 + (ipos[i] >> 8), ix = ox_ - 2 + (ipos[i] & 0xFF);
    const int cy = med3(iy, 0, Hm1), cx = med3(ix, 0, Wm1);
    pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, W, a.Cin) + coff0) << 2));
  };
  unsigned hmax = 0u;
  auto activate_item = [&](const int i, const int oy_, const int ox_) __attribute__((always_inline)) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(__uint_as_float(pre_p[i][j]), psc[j], psh[j]), 0.0f);
    const int iy = oy_ - 2 + (ipos[i] >> 8), ix = ox_ - 2 + (ipos[i] & 0xFF);
    const bool inside = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
#pragma unroll
    for (int j = 0; j < 4; ++j) pre_p[i][j] = __float_as_uint(inside ? v[j] : 0.0f);
  };
  auto store_item = [&](const int i, const int par) __attribute__((always_inline)) {  // (idle slots past the patch are written too: no predicate)
    unsigned h0, h1, l0, l1;
    split_h(__uint_as_float(pre_p[i][0]), __uint_as_float(pre_p[i][1]), h0, l0);
    split_h(__uint_as_float(pre_p[i][2]), __uint_as_float(pre_p[i][3]), h1, l1);
    hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    unsigned char* sp = reinterpret_cast<unsigned char*>(s_P + par * Q_PBUF) + st_base;
    *reinterpret_cast<uint2*>(sp + i * (32 * 32)) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(sp + 2 * Q_PNPXP * 32 + i * (32 * 32)) = make_uint2(l0, l1);
  };
  f32x4 acc[6];
  constexpr int AHEAD = 3, RING = 4;
  auto mid_put = [&](const int par, const int oy_, const int ox_, const f32x4 accv, const int mrow, const int mcol, const bool valid) __attribute__((always_inline)) {
    unsigned char* mb = reinterpret_cast<unsigned char*>(s_M + par * Q_MBUF);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fmaxf(__fmaf_rn(accv[j], os[j], ob[j]), 0.0f);
    const bool inside = (unsigned)(oy_ - 1 + mrow) < (unsigned)H && (unsigned)(ox_ - 1 + mcol) < (unsigned)W;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = inside ? v[j] : 0.0f;
    unsigned h0, h1, l0, l1;
    split_h(v[0], v[1], h0, l0);
    split_h(v[2], v[3], h1, l1);
    hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    const unsigned off = (unsigned)(((ct * Q_MNPXP + mrow * Q_MP + mcol) * 4 + q) * 8);
    if (valid) {
      *reinterpret_cast<uint2*>(mb + off) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(mb + off + 2 * Q_MNPXP * 32) = make_uint2(l0, l1);
    }
  };
  // A: conv a of the tile whose patch is in P[ppar] -> mid planes in M[mpar] (rows as they complete, the edge group last)
  auto role_a = [&](const int ppar, const int mpar, const int oy_, const int ox_) __attribute__((always_inline)) {
    const uint4* pb = s_P + ppar * Q_PBUF + ((q >> 1) * Q_PNPXP + (5 * hh) * Q_PP + i16) * 2 + (q & 1);
#pragma unroll
    for (int o = 0; o < 6; ++o) acc[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    u32x4 xh[RING], xl[RING];
    auto frag = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 7, i = s - 7 * kx, bf = s % RING;
      xh[bf] = __builtin_bit_cast(u32x4, pb[(i * Q_PP + kx) * 2]);
      xl[bf] = __builtin_bit_cast(u32x4, pb[4 * Q_PNPXP + (i * Q_PP + kx) * 2]);
    };
    static_for<0, AHEAD>(frag);
    static_for<0, 21>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 7, i = s - 7 * kx, bf = s % RING;
      if constexpr (s + AHEAD < 21) frag(std::integral_constant<int, s + AHEAD>{});
      static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value, o = i - r;
          if constexpr (o >= 0 && o < 5) {
            if constexpr (pr == 0) acc[o] = mfma_h(Wr[r * 3 + kx][1], xh[bf], acc[o]);
            if constexpr (pr == 1) acc[o] = mfma_h(Wr[r * 3 + kx][0], xl[bf], acc[o]);
            if constexpr (pr == 2) acc[o] = mfma_h(Wr[r * 3 + kx][0], xh[bf], acc[o]);
          }
        });
      });
      if constexpr (s >= 17 && s <= 20) mid_put(mpar, oy_, ox_, acc[s - 17], 5 * hh + (s - 17), i16, true);  // (row o is complete after step 16 + o)
      __builtin_amdgcn_sched_barrier(0);
    });
    // the edge group: mid columns 16, 17 of the 10 rows = 20 pixels, e -> row e >> 1, column 16 + (e & 1); hh = 0: 0..15, hh = 1: 16..19
    const int e = 16 * hh + i16;
    const int ec = min(e, 19);
    const int eb = ((q >> 1) * Q_PNPXP + (ec >> 1) * Q_PP + 16 + (ec & 1)) * 2 + (q & 1);
    const uint4* pe = s_P + ppar * Q_PBUF + eb;
    static_for<0, 9>([&](auto tc) __attribute__((always_inline)) {
      constexpr int tp = decltype(tc)::value, kx = tp / 3, ky = tp - 3 * kx;
      const u32x4 eh = __builtin_bit_cast(u32x4, pe[(ky * Q_PP + kx) * 2]);
      const u32x4 el = __builtin_bit_cast(u32x4, pe[4 * Q_PNPXP + (ky * Q_PP + kx) * 2]);
      acc[5] = mfma_h(Wr[ky * 3 + kx][1], eh, acc[5]);
      acc[5] = mfma_h(Wr[ky * 3 + kx][0], el, acc[5]);
      acc[5] = mfma_h(Wr[ky * 3 + kx][0], eh, acc[5]);
    });
    mid_put(mpar, oy_, ox_, acc[4], 5 * hh + 4, i16, true);
    mid_put(mpar, oy_, ox_, acc[5], ec >> 1, 16 + (ec & 1), e < 20);
  };
  auto out_row = [&](const int n_, const int oy_, const int ox_, const int o) __attribute__((always_inline)) {
    float* out_n = b.out + (size_t)n_ * H * W * C;
    const int ox = min(ox_ + i16, Wm1);
    const int oy = oy_ + 4 * hh + o;
    const unsigned off = (pix_off(min(oy, Hm1), ox, W, C) + (unsigned)ch_l) << 2;
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __fmaf_rn(acc[o][j], os[j], ob[j]);
    v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
    if (ox_ + i16 < W && oy < H) *reinterpret_cast<f32x4*>(at_off(out_n, off)) = v;
  };
  // B: conv b of the tile whose mid is in M[mpar] (accumulators preloaded with its residual rows), stores; between the products
  // the staging: `between(s)`
  auto role_b = [&](const int mpar, const int n_, const int oy_, const int ox_, auto&& between) __attribute__((always_inline)) {
    const uint4* mbp = s_M + mpar * Q_MBUF + ((q >> 1) * Q_MNPXP + (4 * hh) * Q_MP + i16) * 2 + (q & 1);
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] *= rs;
    u32x4 xh[RING], xl[RING];
    auto frag = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 6, i = s - 6 * kx, bf = s % RING;
      xh[bf] = __builtin_bit_cast(u32x4, mbp[(i * Q_MP + kx) * 2]);
      xl[bf] = __builtin_bit_cast(u32x4, mbp[4 * Q_MNPXP + (i * Q_MP + kx) * 2]);
    };
    static_for<0, AHEAD>(frag);
    static_for<0, 18>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, kx = s / 6, i = s - 6 * kx, bf = s % RING;
      if constexpr (s + AHEAD < 18) frag(std::integral_constant<int, s + AHEAD>{});
      static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value, o = i - r;
          if constexpr (o >= 0 && o < 4) {
            if constexpr (pr == 0) acc[o] = mfma_h(Wr[r * 3 + kx][1], xh[bf], acc[o]);
            if constexpr (pr == 1) acc[o] = mfma_h(Wr[r * 3 + kx][0], xl[bf], acc[o]);
            if constexpr (pr == 2) acc[o] = mfma_h(Wr[r * 3 + kx][0], xh[bf], acc[o]);
          }
        });
      });
      if constexpr (s >= 15 && s <= 17) out_row(n_, oy_, ox_, s - 15);  // (row o is complete after step 14 + o)
      between(sc);
      __builtin_amdgcn_sched_barrier(0);
    });
    out_row(n_, oy_, ox_, 3);
  };
  auto preload_res = [&](const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const float* res_n = b.residual + (size_t)n_ * H * W * C;
    const int ox = min(ox_ + i16, Wm1);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int oy = min(oy_ + 4 * hh + o, Hm1);
      acc[o] = *reinterpret_cast<const f32x4*>(at_off(res_n, (pix_off(oy, ox, W, C) + (unsigned)ch_l) << 2));
    }
  };

  // ---- the pipeline.  Step k: A on tile k + 1 (P[(k + 1) & 1] -> M[(k + 1) & 1]), B on tile k (M[k & 1]) and staging tile k + 2
  //      (registers -> P[k & 1]) while requesting tile k + 3; one barrier per step; k = -1 fills it ----
  int n0 = 0, oy0 = 0, ox0 = 0, n1 = 0, oy1 = 0, ox1 = 0, n2 = 0, oy2 = 0, ox2 = 0, n3 = 0, oy3 = 0, ox3 = 0;
  bool v0 = false;
  bool v1 = get_tile(0, n1, oy1, ox1);
  if (!v1) return;
  bool v2 = get_tile(1, n2, oy2, ox2);
  bool v3 = v2 && get_tile(2, n3, oy3, ox3);
  if (!is_a) {  // the first tile's patch, then the second one's requests
#pragma unroll
    for (int i = 0; i < Q_NP; ++i) issue_item(i, n1, oy1, ox1);
#pragma unroll
    for (int i = 0; i < Q_NP; ++i) {
      activate_item(i, oy1, ox1);
      store_item(i, 0);
    }
    const int nq = v2 ? n2 : n1, oyq = v2 ? oy2 : oy1, oxq = v2 ? ox2 : ox1;
#pragma unroll
    for (int i = 0; i < Q_NP; ++i) issue_item(i, nq, oyq, oxq);
  }
  __syncthreads();
  for (int k = -1;; ++k) {
    if (is_a) {
      if (v1) role_a((k + 1) & 1, (k + 1) & 1, oy1, ox1);
    } else {
      // a tile that does not exist is replaced by the last one that does (its planes land in a buffer nobody reads)
      const int oyc = v2 ? oy2 : oy1, oxc = v2 ? ox2 : ox1;
      const int nr = v3 ? n3 : (v2 ? n2 : n1), oyr = v3 ? oy3 : (v2 ? oy2 : oy1), oxr = v3 ? ox3 : (v2 ? ox2 : ox1);
      const int par = k & 1;
      // item i: activate, split + store, next request on consecutive slots; 24 slots dealt over the 18 steps
      auto stage_slot = [&](const int j) __attribute__((always_inline)) {
        const int i = j / 3, st = j - 3 * i;
        if (st == 0) activate_item(i, oyc, oxc);
        if (st == 1) store_item(i, par);
        if (st == 2) issue_item(i, nr, oyr, oxr);
      };
      if (v0) {
        role_b(k & 1, n0, oy0, ox0, [&](auto sc) __attribute__((always_inline)) {
          constexpr int s = decltype(sc)::value;
          constexpr int jlo = (s * 24 + 17) / 18, jhi = ((s + 1) * 24 + 17) / 18;
          static_for<jlo, (jhi < 24 ? jhi : 24)>([&](auto jc) __attribute__((always_inline)) { stage_slot(decltype(jc)::value); });
        });
      } else {
        static_for<0, 24>([&](auto jc) __attribute__((always_inline)) { stage_slot(decltype(jc)::value); });
      }
      if (v1) preload_res(n1, oy1, ox1);  // the next tile's residual rows into the accumulators, now free
    }
    if (!v1) break;  // (uniform) tile k was the workgroup's last
    __syncthreads();
    n0 = n1; oy0 = oy1; ox0 = ox1; v0 = v1;
    n1 = n2; oy1 = oy2; ox1 = ox2; v1 = v2;
    n2 = n3; oy2 = oy3; ox2 = ox3; v2 = v3;
    v3 = v2 && get_tile(k + 4, n3, oy3, ox3);
  }
  if ((hmax & 0xFFFFu) >= 0x7C00u || (hmax >> 16) >= 0x7C00u) atomicOr(a.ovf, 1);
}

}  // namespace

// `a` / `b`: the block's two convolutions as launch_conv_block32 has prepared them (w_scale / w_unscale set); wa / wb: their
// fp16 plane images ([g][ky][plane][kx][quarter][32])
int launch_conv_block32s(const ConvArgs& a, const ConvArgs& b, const void* wa, const void* wb, hipStream_t s) {
  // CPX_BLOCK32_SPLIT: 1 (default) = conv_block32s_kernel, 2 = conv_block32p_kernel (16 x 8 tiles, one barrier per step:
  // measured slower, 59.6 against 54.6 ms per six launches of 1,536 samples on the same box -- profiles/r06_conv_rw_experiments.md)
  static const int form = [] {
    const char* e = std::getenv("CPX_BLOCK32_SPLIT");
    return e == nullptr ? 1 : std::atoi(e);
  }();
  const bool pform = form >= 2;
  BlkTiles td{};
  td.tiles_x = (a.W + K_T - 1) / K_T;
  td.tiles_y = (a.H + (pform ? Q_R : K_T) - 1) / (pform ? Q_R : K_T);
  const long long tiles = (long long)td.tiles_x * td.tiles_y * a.N;
  if (tiles >= (1 << 22) - 8 || td.tiles_x >= 4096 || td.tiles_y >= 4096) return -3;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) / td.tiles_y + 1;
  td.total = (int)tiles;
  static bool lds_ready[64], lds_ready_p[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_block32s_kernel), lds_ready, 160 * 1024 - 1024)) return -1;
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_block32p_kernel), lds_ready_p, 160 * 1024 - 1024)) return -1;
  static int cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (cus_of[dev] == 0) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus_of[dev] = cus;
  }
  int gx = std::max(8, cus_of[dev] / a.groups / 8 * 8);
  if (const char* e = std::getenv("CPX_BLOCK32_GRID")) gx = std::max(8, std::atoi(e) / 8 * 8);
  gx = (int)std::min<long long>(gx, (tiles + 7) / 8 * 8);
  if (pform)
    hipLaunchKernelGGL(conv_block32p_kernel, dim3((unsigned)gx, a.groups), dim3(K_CT), Q_LDS, s, a, b, reinterpret_cast<const uint4*>(wa),
                       reinterpret_cast<const uint4*>(wb), td);
  else
    hipLaunchKernelGGL(conv_block32s_kernel, dim3((unsigned)gx, a.groups), dim3(K_CT), K_LDS, s, a, b, reinterpret_cast<const uint4*>(wa),
                       reinterpret_cast<const uint4*>(wb), td);
  return 0;
}

}  // namespace cpx
