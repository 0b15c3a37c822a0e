// cpx_cnn_rw.hip -- fp16x2 3x3 convolutions with the WEIGHTS RESIDENT IN REGISTERS (round 6).
//
// Why.  conv_bf3w_kernel (cpx_cnn_bf3.hip) feeds every 16x16x32 MFMA from LDS: per tap a wave reads four pixel fragments
// and four weight fragments for twelve products, 0.67 ds_read_b128 per MFMA, and the LDS array serves one such read per
// MFMA slot of a CU -- with its 26 % of bank-conflict cycles the LDS is as busy as the matrix pipe, the weights go through
// LDS once per (chunk, column slice) behind two barriers each, and the kernel sits at 0.40 of the pipe's peak
// (profiles/r05_bench_e2e.json).  For the layers whose weights are small -- stage 3 of WR-ResNet-22-4: 64 -> 64 channels
// per group, 9 x 64 x 64 -- the matrix pipe can be fed differently:
//   * A operand = WEIGHTS, held in registers for the lifetime of a persistent workgroup.  A wave owns 16 output channels
//     of its group: 2 chunks x 9 taps x 2 planes fragments of 16 bytes = 144 VGPRs, loaded once per launch.
//   * B operand = PIXELS: one row of 16 pixels x 32 channels per fragment.  A wave owns all 16 output rows of a 16 x 16 tile
//     (for its 16 channels) and walks the 18 patch rows they need ONCE per (chunk, kx): the row read for output row o (ky = 0)
//     is the row of o - 1 (ky = 1) and o - 2 (ky = 2), so each fragment pair feeds up to nine products -- 108 reads per 432
//     products (0.25 per MFMA), no weight traffic in LDS at all.
//   * Four waves per workgroup, ONE per SIMD, each with the whole 512-register file of its SIMD: 144 weights + 64
//     accumulators + 44 staging + fragments in flight.  What a second wave per SIMD would hide is hidden by distance instead:
//     fragments are requested several steps ahead of their products, global loads a whole chunk ahead.
//   * The patch goes through LDS in 32-channel chunks, double-buffered: while the products of chunk u run from one
//     buffer, chunk u + 1 (loaded while u - 1 ran) takes its BatchNorm + ReLU prologue and fp16 split BETWEEN the products
//     and lands in the other buffer, and chunk u + 2's global loads are issued into the registers that just emptied.
//     ONE barrier per chunk; 2 x 41.7 KB of LDS; one workgroup per CU.
// Arithmetic: conv_bf3w_kernel's fp16x2 (two fp16 planes per operand, three products w_lo x_hi + w_hi x_lo + w_hi x_hi, float32
// accumulate, powers-of-two scales, overflow word + guarded bf16x3 rerun: include/cpx.h CPX_CNN_MATH_FP16X2); the taps are
// summed kx-major (kx, then ky) instead of ky-major -- another float32 order of the same terms.
// The same kernel at stride 2 and 3 (template parameters S, NCH, ROWS; RwGeo below): conv_rw_kernel<2, 1, 8> is the first
// convolution of stage 3 (32 -> 64 channels per group: one 32-channel chunk per tile, 16 x 8 output tiles from 33 x 17 patches,
// the two buffers alternate per tile; HBM-bound at 5.05 TB/s), conv_rw_kernel<3, 2, 4> the first convolution of stage 4
// (64 -> 128: 16 x 4 output tiles from 48 x 12 patches, a workgroup computes 64 of the 128 output channels of its group).
// Measurements of every step on the way, the forms that lost and what was considered and not built:
// profiles/r06_conv_rw_experiments.md; counters of the shipped forms: profiles/r06_conv_sq_counters.json, r06_conv_traffic.json.
// Reference semantics: /root/reference/src/ml_tools/resnet/wr_resnet.py:49-98 (wr_block: BN -> ReLU -> conv3x3 -> BN -> ReLU ->
// conv3x3 -> add), kerasmodel.py:441-454.
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
// element offset of pixel (y, x) in an NHWC sample (24-bit multiplies: see cpx_cnn_bf3.hip)
__device__ __forceinline__ unsigned pix_off(int y, int x, int W, int C) {
  return __umul24(__umul24((unsigned)y, (unsigned)W) + (unsigned)x, (unsigned)C);
}
__device__ __forceinline__ float relu_bits(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// two float32 -> two fp16 planes, each rounded to nearest (cpx_cnn_bf3.hip: split_pair2_h)
__device__ __forceinline__ void split_h(float a, float b, unsigned& hi, unsigned& lo) {
  f16x2 v = {(_Float16)a, (_Float16)b};
  hi = __builtin_bit_cast(unsigned, v);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(a), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(b), "v"(hi));
}
__device__ __forceinline__ f32x4 mfma_h(u32x4 w, u32x4 x, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
}

struct RwTiles {
  unsigned long long m_tx, m_ty;  // floor(2^42 / d) + 1 (cpx_cnn_bf3.hip: TileDiv)
  int tiles_x, tiles_y, total;
};
// a loop whose index is a compile-time constant in every iteration (register arrays indexed by it stay registers)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ int div_magic(int n, unsigned long long m) { return (int)(((unsigned long long)n * m) >> 42); }

constexpr int RW_TW = 16;                      // output columns of a tile = the pixels of one B fragment
constexpr int RW_CT = 256;                     // threads: four waves, one per SIMD
constexpr int RW_PPI = RW_CT / 8;              // patch pixels per staging round (eight 16-byte pieces per pixel and chunk)
constexpr int RW_WC = 64;                      // output channels of a workgroup: 16 per wave

// Geometry of an instantiation.  S: stride; NCH: 32-channel chunks of the group's input channels; ROWS: output rows of a tile
// (every wave computes all of them for its 16 channels).
template <int S, int NCH, int ROWS>
struct RwGeo {
  static constexpr int PW = 15 * S + 3, PH = (ROWS - 1) * S + 3, NPX = PH * PW;   // the staged patch of one tile
  static constexpr int NP = (NPX * 8 + RW_CT - 1) / RW_CT;                           // staging items per thread and chunk
  // pixels per (plane, quarter pair) region: the patch + the slots the last staging round's idle threads write (no
  // predicate in the stream); NPXP * 32 B = 64 mod 128 (cpx_cnn_bf3.hip: W_NPXP)
  static constexpr int NPXP = (NP * RW_PPI + 3) / 4 * 4 + 2;
  static constexpr int BUF = 2 * 2 * NPXP * 2;                                       // 16-byte entries of one chunk buffer
  static constexpr int STEPS = 3 * PH;                                               // (kx, patch row) steps of a chunk
  static constexpr int CING = 32 * NCH;
  static constexpr size_t LDS = (size_t)2 * BUF * 16 + (size_t)(2 * CING + 3 * RW_WC) * sizeof(float);
};

// LDS: [chunk buffer 0][chunk buffer 1][BatchNorm scale, shift of the group's input channels][per-channel epilogue parameters: 3 x 64]
// BN: the layer has a BatchNorm + ReLU prologue (a template parameter, like every other condition inside the product stream:
// a branch there ends the scheduling region, and with one wave per SIMD nothing else fills the matrix pipe meanwhile)
// RES: the layer adds a residual tensor (requested into registers beside the tile's last products, added in the epilogue)
template <int S, int NCH, int ROWS, bool BN, bool RES>
__global__ __launch_bounds__(RW_CT) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_rw_kernel(ConvArgs a, const uint4* __restrict__ wimg, RwTiles td) {
  using G = RwGeo<S, NCH, ROWS>;
  constexpr int PW = G::PW, NPX = G::NPX, NP = G::NP, NPXP = G::NPXP, BUF = G::BUF, STEPS = G::STEPS, CING = G::CING, PH = G::PH;
  if (*a.ovf != 0) return;  // (the guarded rerun follows)
  extern __shared__ __attribute__((aligned(16))) uint4 lds4[];
  uint4* s_buf = lds4;
  float* s_bn = reinterpret_cast<float*>(lds4 + 2 * BUF);
  float* s_par = s_bn + 2 * CING;  // [os 64][ob 64][rs 64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
  const int ct = wave;  // the wave's 16-column tile of the workgroup's 64 output channels
  const int cout_g = a.Cout / a.groups, nhalf = cout_g / RW_WC;
  const int g = blockIdx.y / nhalf, half = blockIdx.y - g * nhalf;  // group; which 64 of its output channels
  constexpr bool has_bn = BN;

  // ---- the wave's weights: fp16 image [g][chunk][ky][plane][kx][quarter][cout_g] of 16-byte entries (split_weights32_kernel) ----
  u32x4 Wr[NCH][9][2];
  {
    const u32x4* wg = reinterpret_cast<const u32x4*>(wimg) + (size_t)g * NCH * 3 * 24 * cout_g + (q * cout_g + half * RW_WC + ct * 16 + i16);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int p = 0; p < 2; ++p) Wr[c][ky * 3 + kx][p] = wg[(size_t)((c * 3 + ky) * 24 + (p * 3 + kx) * 4) * cout_g];
  }
  if (tid < CING) {  // relu(x s + b) 2^k = relu(x (s 2^k) + b 2^k), exactly
    const int ci = g * CING + tid;
    s_bn[tid] = has_bn ? a.in_scale[ci] * a.act_scale : a.act_scale;
    s_bn[CING + tid] = has_bn ? a.in_shift[ci] * a.act_scale : 0.0f;
  }
  if (tid < RW_WC) {
    const int ch = g * cout_g + half * RW_WC + tid;
    float os = a.w_unscale[ch] * a.act_unscale;  // (powers of two: exact)
    if (a.out_scale) os *= a.out_scale[ch];
    float ob = a.out_shift ? a.out_shift[ch] : 0.0f;
    if (a.sc_in && a.sc_bias) ob += a.sc_bias[ch];
    s_par[tid] = os;
    s_par[RW_WC + tid] = ob;
    s_par[2 * RW_WC + tid] = a.w_scale[ch] * a.act_scale;  // what the accumulators are scaled by: the fused shortcut's factor
  }

  // ---- tiles: every XCD walks its own contiguous eighth of the tile space (conv_block32_kernel) ----
  const int per_xcd = (td.total + 7) >> 3;
  auto tile_of = [&](int t) { return (t & 7) * per_xcd + (t >> 3); };
  auto decode = [&](int tile, int& n_, int& oy_, int& ox_) {
    int qd = div_magic(tile, td.m_tx);
    ox_ = (tile - qd * td.tiles_x) * RW_TW;
    tile = qd;
    qd = div_magic(tile, td.m_ty);
    oy_ = (tile - qd * td.tiles_y) * ROWS;
    n_ = qd;
  };
  int t = blockIdx.x;
  while (t < 8 * per_xcd && tile_of(t) >= td.total) t += gridDim.x;
  if (t >= 8 * per_xcd) return;
  // the next tile of this workgroup, or (none left) the given one again: the last tiles stage themselves once more -- into
  // registers and a buffer nobody reads -- rather than branch in the stream
  auto advance = [&](int& n_, int& oy_, int& ox_) {
    t += gridDim.x;
    const bool more = t < 8 * per_xcd && tile_of(t) < td.total;  // (a workgroup's tiles ascend within its XCD's eighth)
    if (more) decode(tile_of(t), n_, oy_, ox_);
    return more;
  };
  int n0, oy0, ox0, n1, oy1, ox1, n2, oy2, ox2;
  decode(tile_of(t), n0, oy0, ox0);

  // ---- staging: item i of a thread = one 16-byte piece (4 channels) of patch pixel (tid >> 3) + 32 i, piece tid & 7.
  //      What depends on the thread alone is computed once per launch: each item's patch row / column, the byte offset of its
  //      LDS slot (the items of a thread lie 1 KB apart: immediate offsets of one base per buffer), the piece's BatchNorm
  //      parameters.  Per item and tile that leaves: two clamps (v_med3_i32), three multiply-adds for the address, two
  //      compares for the padding ----
  u32x4 pre_p[NP];
  const int q8 = tid & 7;
  int ipos[NP];  // patch row << 8 | column of item i (the slots past the patch repeat its last pixel)
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int px = min((tid >> 3) + RW_PPI * i, NPX - 1);
    const int py = px / PW;
    ipos[i] = (py << 8) | (px - py * PW);
  }
  // channels 4 q8 .. 4 q8 + 3 of a chunk: quarter pair q8 >> 2, 8-byte slot q8 & 3 of the pixel's 32 bytes
  const unsigned st_base = (unsigned)((((q8 >> 2) * NPXP + (tid >> 3)) * 4 + (q8 & 3)) * 8);
  f32x4 psc[NCH], psh[NCH];  // filled behind the barrier that publishes s_bn
  const unsigned coff0 = (unsigned)(g * CING + 4 * q8);
  auto med3 = [](int x, int lo, int hi) __attribute__((always_inline)) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "s"(hi));
    return r;
  };
  const int Hm1 = a.H - 1, Wm1 = a.W - 1;
  auto issue_item = [&](const int i, const int c, const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const float* in_n = a.in + (size_t)n_ * a.H * a.W * a.Cin;  // (uniform)
    const int iy = oy_ * S - a.pad_top + (ipos[i] >> 8), ix = ox_ * S - a.pad_left + (ipos[i] & 0xFF);
    const int cy = med3(iy, 0, Hm1), cx = med3(ix, 0, Wm1);  // a clamped address is always loaded; padding is zeroed at commit
    pre_p[i] = *reinterpret_cast<const u32x4*>(at_off(in_n, (pix_off(cy, cx, a.W, a.Cin) + coff0 + (unsigned)(c * 32)) << 2));
  };
  // out of fp16's range = a high plane that came out infinite: the running maximum of the high planes' magnitudes, two
  // packed halves per instruction (conv_block32_kernel's test; the sign bits are masked off where no ReLU precedes)
  unsigned hmax = 0u;
  auto pk_max_u16 = [](unsigned x, unsigned y) __attribute__((always_inline)) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
  };
  // registers -> the chunk's fp16 planes in LDS buffer `bsel`, in two stages that ride on different steps of the product
  // stream (a step of nine products hides ~18 vector instructions; an item's 35 in one lump left the matrix pipe idle):
  // activate = BatchNorm + ReLU prologue and zero padding, in place; store = fp16 split + the two 8-byte LDS stores
  // (no predicate, no branch: the last round's idle threads convert a clamped load into slots past the patch)
  auto activate_item = [&](const int i, const int c, const int oy_, const int ox_) __attribute__((always_inline)) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __fmaf_rn(__uint_as_float(pre_p[i][j]), psc[c][j], psh[c][j]);
    if (has_bn) {  // (without a prologue the values are only scaled)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
    }
    // zero padding, as TensorFlow pads the activated tensor
    const int iy = oy_ * S - a.pad_top + (ipos[i] >> 8), ix = ox_ * S - a.pad_left + (ipos[i] & 0xFF);
    const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
    for (int j = 0; j < 4; ++j) pre_p[i][j] = __float_as_uint(inside ? v[j] : 0.0f);
  };
  auto store_item = [&](const int i, const int bsel) __attribute__((always_inline)) {
    unsigned h0, h1, l0, l1;
    split_h(__uint_as_float(pre_p[i][0]), __uint_as_float(pre_p[i][1]), h0, l0);
    split_h(__uint_as_float(pre_p[i][2]), __uint_as_float(pre_p[i][3]), h1, l1);
    if (has_bn) hmax = pk_max_u16(pk_max_u16(hmax, h0), h1);
    else hmax = pk_max_u16(pk_max_u16(hmax, h0 & 0x7FFF7FFFu), h1 & 0x7FFF7FFFu);
    unsigned char* sp = reinterpret_cast<unsigned char*>(s_buf) + st_base;
    *reinterpret_cast<uint2*>(sp + (bsel * BUF * 16 + i * (RW_PPI * 32))) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(sp + (bsel * BUF * 16 + 2 * NPXP * 32 + i * (RW_PPI * 32))) = make_uint2(l0, l1);
  };

  // ---- accumulators: ROWS output rows x (16 pixels x 16 channels); a lane holds channels ct 16 + 4 q .. + 3 of pixel column i16 ----
  f32x4 acc[ROWS];
  const int ch_l = g * cout_g + half * RW_WC + ct * 16 + 4 * q;
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int o = 0; o < ROWS; ++o) acc[o] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  };
  // the tile's residual rows: one wave per SIMD leaves registers to spare, so they are requested beside the products of the
  // tile's last chunk and have landed when the epilogue adds them (in the accumulators before the first product, as
  // conv_bf3w_kernel has them, the first products of every tile waited out a trip to HBM: SQ_WAIT_ANY 46 % of the cycles)
  f32x4 rres[RES ? ROWS : 1];
  auto issue_res = [&](const int o, const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    if constexpr (RES) {
      const float* res_n = a.residual + (size_t)n_ * a.Ho * a.Wo * a.Cout;  // (uniform)
      const int ox = min(ox_ + i16, a.Wo - 1), oy = min(oy_ + o, a.Ho - 1);
      rres[o] = *reinterpret_cast<const f32x4*>(at_off(res_n, (pix_off(oy, ox, a.Wo, a.Cout) + (unsigned)ch_l) << 2));
    }
  };

  // ---- the products of chunk C from buffer B: 3 kx x PH patch rows, each row's fragment pair feeding the output rows it is
  //      tap row 0, 1, 2 of (stride 1: up to three; stride 3: exactly one); `between(s)` is called after step s with the
  //      staging work that rides along.  Fragments are requested RW_AHEAD steps before their products (one wave per SIMD:
  //      distance instead of a partner) ----
  const int a_base = ((q >> 1) * NPXP + S * i16) * 2 + (q & 1);
#ifndef CPX_RW_AHEAD
#define CPX_RW_AHEAD 3
#endif
  constexpr int RW_AHEAD = CPX_RW_AHEAD, RW_RING = CPX_RW_AHEAD + 1;
  auto compute = [&](auto cc, auto bc, auto&& between) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value, B = decltype(bc)::value;
    const uint4* sb = s_buf + B * BUF + a_base;
    u32x4 xh[RW_RING], xl[RW_RING];
    auto frag = [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      constexpr int kx = s / PH, i = s - PH * kx, bf = s % RW_RING;
      xh[bf] = __builtin_bit_cast(u32x4, sb[(i * PW + kx) * 2]);
      xl[bf] = __builtin_bit_cast(u32x4, sb[4 * NPXP + (i * PW + kx) * 2]);
    };
    static_for<0, RW_AHEAD>(frag);
    static_for<0, STEPS>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      constexpr int kx = s / PH, i = s - PH * kx, bf = s % RW_RING;
      if constexpr (s + RW_AHEAD < STEPS) frag(std::integral_constant<int, s + RW_AHEAD>{});
      // the three plane products of a tap in their order, the output rows of the step taking turns
      static_for<0, 3>([&](auto pc) __attribute__((always_inline)) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, 3>([&](auto rc) __attribute__((always_inline)) {
          constexpr int r = decltype(rc)::value, d = i - r, o = d / S;
          if constexpr (d >= 0 && d % S == 0 && o < ROWS) {
            if constexpr (pr == 0) acc[o] = mfma_h(Wr[C][r * 3 + kx][1], xh[bf], acc[o]);
            if constexpr (pr == 1) acc[o] = mfma_h(Wr[C][r * 3 + kx][0], xl[bf], acc[o]);
            if constexpr (pr == 2) acc[o] = mfma_h(Wr[C][r * 3 + kx][0], xh[bf], acc[o]);
          }
        });
      });
      between(sc);
      // pin the step order: left alone, the scheduler sinks the fragment reads to just in front of their products and every
      // step then waits out an LDS round trip with nothing else on the SIMD (SQ_WAIT_ANY 30 % of the wave's cycles, measured)
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  // one unit of the stream: the products of (chunk C, buffer B) and, spread over its steps, the staging of what follows:
  // the NP items in registers (chunk CC of the tile at (oyc, oxc)) take their prologue and go to the other buffer, three
  // slots per item (activate, split + store, next request) dealt evenly over the steps; the request that refills an item's
  // registers is chunk CI of the tile (ni, oyi, oxi); LAST: the tile's residual rows are requested too
  auto unit = [&](auto cc, auto bc, auto ccc, auto cic, auto lastc, const int oyc, const int oxc, const int ni, const int oyi,
                  const int oxi) __attribute__((always_inline)) {
    constexpr int B = decltype(bc)::value, CC = decltype(ccc)::value, CI = decltype(cic)::value;
    constexpr bool LAST = decltype(lastc)::value;
    compute(cc, bc, [&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value;
      constexpr int jlo = (s * 3 * NP + STEPS - 1) / STEPS, jhi = ((s + 1) * 3 * NP + STEPS - 1) / STEPS;
      static_for<jlo, (jhi < 3 * NP ? jhi : 3 * NP)>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value, i = j / 3, k = j % 3;
        if constexpr (k == 0) activate_item(i, CC, oyc, oxc);
        if constexpr (k == 1) store_item(i, B ^ 1);
        if constexpr (k == 2) issue_item(i, CI, ni, oyi, oxi);
      });
      if constexpr (LAST && RES) {
        constexpr int olo = (s * ROWS + STEPS - 1) / STEPS, ohi = ((s + 1) * ROWS + STEPS - 1) / STEPS;
        static_for<olo, (ohi < ROWS ? ohi : ROWS)>([&](auto oc) __attribute__((always_inline)) { issue_res(decltype(oc)::value, n0, oy0, ox0); });
      }
    });
  };

  // ---- epilogue of a tile: fused 1x1 shortcut (float32 MFMA, as conv_bf3w_kernel), affine, residual, ReLU, 16-byte stores ----
  auto finish = [&](const int n_, const int oy_, const int ox_) __attribute__((always_inline)) {
    const int ox = min(ox_ + i16, a.Wo - 1);
    if (a.sc_in) {
      const int sc_cg = a.sc_cin / a.groups;
      const float* wsc = a.sc_w + ((size_t)g * sc_cg + q) * cout_g + half * RW_WC + ct * 16 + i16;
      const float ss = s_par[2 * RW_WC + ct * 16 + i16];
      const float* sc_n = a.sc_in + (size_t)n_ * a.sc_H * a.sc_W * a.sc_cin + g * sc_cg + q;
      for (int k4 = 0; k4 < sc_cg; k4 += 4) {
        const float ws = wsc[(size_t)k4 * cout_g] * ss;
#pragma unroll
        for (int o = 0; o < ROWS; ++o) {
          const int oy = min(oy_ + o, a.Ho - 1);
          const float xs = sc_n[((size_t)(oy * a.sc_stride) * a.sc_W + ox * a.sc_stride) * a.sc_cin + k4];
          acc[o] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws, xs, acc[o], 0, 0, 0);
        }
      }
    }
    float* out_n = a.out + (size_t)n_ * a.Ho * a.Wo * a.Cout;
    const f32x4 os = *reinterpret_cast<const f32x4*>(s_par + ct * 16 + 4 * q);
    const f32x4 ob = *reinterpret_cast<const f32x4*>(s_par + RW_WC + ct * 16 + 4 * q);
    const bool xok = ox_ + i16 < a.Wo;
    // nothing runs beside the epilogue on this SIMD: every instruction here is matrix-pipe idle time.  One fused multiply-add
    // per value (the affine's one rounding instead of two: inside the layer's tolerance, another float32 order like the tap
    // order), one per-lane offset for the tile + a uniform row stride, whole tiles without per-row tests
    const unsigned off0 = (pix_off(min(oy_, a.Ho - 1), ox, a.Wo, a.Cout) + (unsigned)ch_l) << 2;
    const unsigned row_bytes = (unsigned)(a.Wo * a.Cout) << 2;  // (uniform)
    const bool whole = oy_ + ROWS <= a.Ho;                      // (uniform)
    auto row = [&](const int o, const bool ok, const unsigned off) __attribute__((always_inline)) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __fmaf_rn(acc[o][j], os[j], ob[j]);
      if constexpr (RES) v += rres[o];
      if (a.relu) {
        asm volatile("");
        v.x = relu_bits(v.x); v.y = relu_bits(v.y); v.z = relu_bits(v.z); v.w = relu_bits(v.w);
      }
      if (ok) *reinterpret_cast<f32x4*>(at_off(out_n, off)) = v;
    };
    if (whole) {
#pragma unroll
      for (int o = 0; o < ROWS; ++o) row(o, xok, off0 + (unsigned)o * row_bytes);
    } else {
#pragma unroll
      for (int o = 0; o < ROWS; ++o) row(o, xok && oy_ + o < a.Ho, off0 + (unsigned)min(o, a.Ho - 1 - oy_) * row_bytes);
    }
  };

  // ---- prologue: the first tile's first chunk goes through the registers with nothing beside it ----
#pragma unroll
  for (int i = 0; i < NP; ++i) issue_item(i, 0, n0, oy0, ox0);
  __syncthreads();  // BatchNorm and epilogue parameters are in LDS
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    psc[c] = *reinterpret_cast<const f32x4*>(s_bn + c * 32 + 4 * q8);
    psh[c] = *reinterpret_cast<const f32x4*>(s_bn + CING + c * 32 + 4 * q8);
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    activate_item(i, 0, oy0, ox0);
    store_item(i, 0);
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  if constexpr (NCH == 2) {
    // units (tile, chunk 0) from buffer 0, (tile, chunk 1) from buffer 1
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_item(i, 1, n0, oy0, ox0);
    init_acc();
    for (;;) {
      n1 = n0; oy1 = oy0; ox1 = ox0;
      const bool more = advance(n1, oy1, ox1);
      // beside chunk 0's products: chunk 1 -> buffer 1, the next tile's chunk 0 -> registers
      __syncthreads();
      unit(I0{}, I0{}, I1{}, I0{}, No{}, oy0, ox0, n1, oy1, ox1);
      // beside chunk 1's: the next tile's chunk 0 -> buffer 0, its chunk 1 -> registers, this tile's residual -> registers
      __syncthreads();
      unit(I1{}, I1{}, I0{}, I1{}, Yes{}, oy1, ox1, n1, oy1, ox1);
      finish(n0, oy0, ox0);
      if (!more) break;
      n0 = n1; oy0 = oy1; ox0 = ox1;
      init_acc();
    }
  } else {
    // one chunk per tile: the tiles alternate between the buffers; beside tile k's products tile k + 1 (in registers) goes
    // to the other buffer and tile k + 2 is requested
    n1 = n0; oy1 = oy0; ox1 = ox0;
    bool more1 = advance(n1, oy1, ox1);
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_item(i, 0, n1, oy1, ox1);
    init_acc();
    for (;;) {
      n2 = n1; oy2 = oy1; ox2 = ox1;
      bool more2 = more1 && advance(n2, oy2, ox2);
      __syncthreads();
      unit(I0{}, I0{}, I0{}, I0{}, Yes{}, oy1, ox1, n2, oy2, ox2);
      finish(n0, oy0, ox0);
      if (!more1) break;
      n0 = n1; oy0 = oy1; ox0 = ox1; n1 = n2; oy1 = oy2; ox1 = ox2; more1 = more2;
      init_acc();
      n2 = n1; oy2 = oy1; ox2 = ox1;
      more2 = more1 && advance(n2, oy2, ox2);
      __syncthreads();
      unit(I0{}, I1{}, I0{}, I0{}, Yes{}, oy1, ox1, n2, oy2, ox2);
      finish(n0, oy0, ox0);
      if (!more1) break;
      n0 = n1; oy0 = oy1; ox0 = ox1; n1 = n2; oy1 = oy2; ox1 = ox2; more1 = more2;
      init_acc();
    }
  }
  if ((hmax & 0x7FFFu) >= 0x7C00u || ((hmax >> 16) & 0x7FFFu) >= 0x7C00u) atomicOr(a.ovf, 1);  // (infinity or NaN: out of fp16's range)
}

}  // namespace

// the layers conv_rw_kernel takes (fp16x2, 3x3, SAME padding; channels per group in -> out):
//   1  stride 1, 64 -> 64   (stage 3 of WR-ResNet-22-4: the convolutions of its blocks but the strided first one)
//   2  stride 2, 32 -> 64   (that strided first one)
//   3  stride 3, 64 -> 128  (stage 4's)
// CPX_CNN_RW: bit k - 1 enables kind k (default: all; 0 = none)
int conv_rw_kind(const ConvArgs& a) {
  static const int enabled = [] {
    const char* e = std::getenv("CPX_CNN_RW");
    return e == nullptr ? 7 : std::atoi(e);
  }();
  if (a.ksize != 3 || a.groups < 1 || a.Cin % a.groups || a.Cout % a.groups) return 0;
  const int cin_g = a.Cin / a.groups, cout_g = a.Cout / a.groups;
  int kind = 0;
  if (a.stride == 1 && cin_g == 64 && cout_g == 64) kind = 1;
  else if (a.stride == 2 && cin_g == 32 && cout_g == 64) kind = 2;
  else if (a.stride == 3 && cin_g == 64 && cout_g == 128) kind = 3;
  return (kind && ((enabled >> (kind - 1)) & 1)) ? kind : 0;
}
bool conv_rw_layer(const ConvArgs& a) { return conv_rw_kind(a) == 1; }

namespace {
template <int S, int NCH, int ROWS, bool BN, bool RES>
int launch_rw_t(const ConvArgs& a, const uint4* wimg, hipStream_t s) {
  using G = RwGeo<S, NCH, ROWS>;
  static_assert(G::LDS <= 160 * 1024 - 1024, "two chunk buffers must fit the CU's LDS");
  RwTiles td{};
  td.tiles_x = (a.Wo + RW_TW - 1) / RW_TW;
  td.tiles_y = (a.Ho + ROWS - 1) / ROWS;
  const long long tiles = (long long)td.tiles_x * td.tiles_y * a.N;
  if (tiles >= (1 << 22) - 8 || td.tiles_x >= 4096 || td.tiles_y >= 4096) return -3;
  td.m_tx = (1ull << 42) / td.tiles_x + 1;
  td.m_ty = (1ull << 42) / td.tiles_y + 1;
  td.total = (int)tiles;
  static bool lds_ready[64];
  if (!cpx_dyn_lds_ready(reinterpret_cast<const void*>(conv_rw_kernel<S, NCH, ROWS, BN, RES>), lds_ready, 160 * 1024 - 1024)) return -1;
  // one workgroup per CU (four waves, each with a SIMD's whole register file), shared among the (group, 64-channel half)
  // pairs; a multiple of eight per pair so that blockIdx.x & 7 names the XCD -- the halves of a group then walk the same tiles
  // on the same XCD at the same time and share the patch in its L2
  static int cus_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
  if (cus_of[dev] == 0) {
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    cus_of[dev] = cus;
  }
  const int ny = a.groups * ((a.Cout / a.groups) / RW_WC);
  int gx = std::max(8, cus_of[dev] / ny / 8 * 8);
  if (const char* e = std::getenv("CPX_RW_GRID")) gx = std::max(8, std::atoi(e) / 8 * 8);
  gx = (int)std::min<long long>(gx, (tiles + 7) / 8 * 8);
  hipLaunchKernelGGL((conv_rw_kernel<S, NCH, ROWS, BN, RES>), dim3((unsigned)gx, (unsigned)ny), dim3(RW_CT), G::LDS, s, a, wimg, td);
  return 0;
}
}  // namespace

// `wimg`: the layer's fp16 plane image in 32-channel chunks (split_weights32_kernel with scales); a.w_scale / a.w_unscale /
// a.ovf / act_scale set.  -2: not a launch this kernel takes (the caller's other kernels do)
int launch_conv_rw(const ConvArgs& a, const void* wimg, hipStream_t s) {
  const int kind = conv_rw_kind(a);
  if (!kind || !a.half || a.planes != 2 || a.ovf == nullptr || a.in_planes || a.out_planes) return -2;
  if ((long long)a.H * a.W >= (1 << 24) || (long long)a.Ho * a.Wo >= (1 << 24)) return -3;
  if (a.sc_in && ((a.sc_cin / a.groups) & 3)) return -2;  // the fused shortcut walks K in fours
  const uint4* wi = reinterpret_cast<const uint4*>(wimg);
  const bool bn = a.in_scale != nullptr, res = a.residual != nullptr;
  if (kind == 1) {
    if (a.pad_top != 1 || a.pad_left != 1 || a.H != a.Ho || a.W != a.Wo) return -2;
    if (bn && res) return launch_rw_t<1, 2, 16, true, true>(a, wi, s);
    if (bn) return launch_rw_t<1, 2, 16, true, false>(a, wi, s);
    if (res) return launch_rw_t<1, 2, 16, false, true>(a, wi, s);
    return launch_rw_t<1, 2, 16, false, false>(a, wi, s);
  }
  if (res || a.sc_in || a.pad_top < 0 || a.pad_top > 1 || a.pad_left < 0 || a.pad_left > 1) return -2;  // (a stage's first convolution)
  if (kind == 2) return bn ? launch_rw_t<2, 1, 8, true, false>(a, wi, s) : launch_rw_t<2, 1, 8, false, false>(a, wi, s);
  return bn ? launch_rw_t<3, 2, 4, true, false>(a, wi, s) : launch_rw_t<3, 2, 4, false, false>(a, wi, s);
}

}  // namespace cpx
