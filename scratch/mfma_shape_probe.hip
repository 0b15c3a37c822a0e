// MFMA shape probe for the split-operand convolution ("16x16x32 measured, not argued").
// Two kernels do the SAME arithmetic per wave -- a 32-pixel x 32-column output tile over K = 288 (9 taps x 32 channels),
// three bf16 planes per operand, six plane products per K step, every operand fragment re-read from LDS by
// ds_read_b128 exactly as conv_bf3_kernel's inner loop does -- once with v_mfma_f32_32x32x16_bf16 (one accumulator
// tile, 18 steps of 6 MFMAs) and once with v_mfma_f32_16x16x32_bf16 (2 x 2 accumulator tiles, 9 steps of 24 MFMAs).
// Operands are random float32 values split into planes (planes 1 and 2 are remainders: dense random bits), 16 waves per CU (one 1024-thread workgroup: the 115 KB image of both forms), no global traffic inside the loop.  Interleaved rounds in one process; prints
// TFLOP/s of float32-equivalent work (2 * M * N * K per tile, i.e. the figure bench.py's roofline uses) and the
// in-kernel clock (s_memtime / s_memrealtime) of each form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NPX = 324, PW = 18;   // 18 x 18 staged pixels of a 16 x 16 tile
constexpr int LDS_A = 3 * 4 * NPX;  // entries: 3 planes x 32 channels (4 x 8) x pixels
constexpr int LDS_B = 3 * 9 * 4 * 32;

__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k32(const uint4* src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < LDS_A + LDS_B; i += 1024) lds[i] = src[i];
  __syncthreads();
  const uint4* sa = lds;
  const uint4* sb = lds + LDS_A;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 7;
  const int pi = lane & 31, kh = lane >> 5;
  const int a_base = (2 * wave + pi / 16) * PW + (pi % 16);
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");  // the image is loop-invariant: keep the fragment reads inside the loop
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int koff = (tap / 3) * PW + tap % 3;
        bf16x8 av[3], bv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          av[p] = __builtin_bit_cast(bf16x8, sa[(p * 4 + 2 * c + kh) * NPX + a_base + koff]);
          bv[p] = __builtin_bit_cast(bf16x8, sb[((p * 9 + tap) * 4 + 2 * c + kh) * 32 + pi]);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[2], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[1], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[0], bv[0], acc, 0, 0, 0);
      }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

// LDS image of the 16x16x32 form: A [plane][quarter pair][pixel][2 quarters] (32 bytes per pixel: conflict-free in
// ds_read_b128's lane groups without padding), B [plane][tap][quarter][column]
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void k16(const uint4* src, float* out, unsigned long long* clk, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < LDS_A + LDS_B; i += 1024) lds[i] = src[i];
  __syncthreads();
  const uint4* sa = lds;
  const uint4* sb = lds + LDS_A;
  const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 7;
  const int i16 = lane & 15, q = lane >> 4;
  const int a_base = ((q >> 1) * NPX + (2 * wave) * PW + i16) * 2 + (q & 1);
  const int b_base = q * 32 + i16;
  f32x4 acc[2][2];
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 4; ++r) acc[m][n][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");  // the image is loop-invariant: keep the fragment reads inside the loop
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int koff = (tap / 3) * PW + tap % 3;
      bf16x8 av[2][3], bv[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int m = 0; m < 2; ++m) av[m][p] = __builtin_bit_cast(bf16x8, sa[p * 4 * NPX + a_base + (m * PW + koff) * 2]);
#pragma unroll
        for (int n = 0; n < 2; ++n) bv[n][p] = __builtin_bit_cast(bf16x8, sb[(p * 9 + tap) * 128 + b_base + 16 * n]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][1], bv[n][1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][0], bv[n][2], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][2], bv[n][0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][0], bv[n][1], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][1], bv[n][0], acc[m][n], 0, 0, 0);
          acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m][0], bv[n][0], acc[m][n], 0, 0, 0);
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int r = 0; r < 4; ++r) s += acc[m][n][r];
  out[blockIdx.x * 1024 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static unsigned short bf16_rn(float x) {
  unsigned u; memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const bool zero = argc > 1 && !strcmp(argv[1], "zero");
  const int n_entries = LDS_A + LDS_B;
  std::vector<unsigned short> h((size_t)n_entries * 8);
  srand(7);
  // every 16-byte entry = 8 values of one plane; planes are filled from independent random float32 values split
  // exactly as split_pair does (value, remainder, remainder of the remainder)
  auto fill = [&](int base, int per_plane, float scale, bool relu) {
    for (int e = 0; e < per_plane; ++e)
      for (int j = 0; j < 8; ++j) {
        float x = scale * ((rand() % 20001) / 10000.0f - 1.0f);
        if (relu && x < 0) x = 0;
        if (zero) x = 0;
        const unsigned short p0 = bf16_rn(x);
        const float r1 = x - bf16_f(p0);
        const unsigned short p1 = bf16_rn(r1);
        const float r2 = r1 - bf16_f(p1);
        const unsigned short p2 = bf16_rn(r2);
        h[((size_t)(base + 0 * per_plane + e)) * 8 + j] = p0;
        h[((size_t)(base + 1 * per_plane + e)) * 8 + j] = p1;
        h[((size_t)(base + 2 * per_plane + e)) * 8 + j] = p2;
      }
  };
  fill(0, 4 * NPX, 4.0f, true);
  fill(LDS_A, 9 * 4 * 32, 0.08f, false);
  uint4* dsrc; float* dout; unsigned long long* dclk;
  const int blocks = 512;
  hipMalloc(&dsrc, (size_t)n_entries * 16); hipMalloc(&dout, (size_t)blocks * 1024 * 4); hipMalloc(&dclk, (size_t)blocks * 16);
  hipMemcpy(dsrc, h.data(), (size_t)n_entries * 16, hipMemcpyHostToDevice);
  const size_t lds = (size_t)n_entries * 16;
  hipFuncSetAttribute((const void*)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 4000;
  const double flops = (double)blocks * 16 * iters * 2.0 * 32 * 32 * 288;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<unsigned long long> hc((size_t)blocks * 2);
  for (int round = 0; round < 6; ++round)
    for (int which = 0; which < 2; ++which) {
      hipEventRecord(e0);
      for (int rep = 0; rep < 3; ++rep) {
        if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(1024), lds, 0, dsrc, dout, dclk, iters);
        else hipLaunchKernelGGL(k16, dim3(blocks), dim3(1024), lds, 0, dsrc, dout, dclk, iters);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost);
      double cyc = 0, ghz = 0;
      for (int b = 0; b < blocks; ++b) { cyc += hc[2 * b]; ghz += (double)hc[2 * b] / hc[2 * b + 1] * 0.1; }
      printf("round %d %s: %.3f ms/launch  %.1f TFLOP/s (f32-equivalent)  loop cycles %.0f  in-kernel clock %.2f GHz%s\n", round,
             which == 0 ? "32x32x16" : "16x16x32", ms / 3, 3 * flops / (ms * 1e-3) / 1e12, cyc / blocks, ghz / blocks, zero ? " [zero data]" : "");
    }
  return 0;
}
