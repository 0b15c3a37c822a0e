// Bare MFMA issue-rate probe: v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 streams per wave, operands in
// registers, random (non-zero) data, 1 or 2 waves per SIMD on every CU.  Prints TFLOP/s of MFMA work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k_bf16(const uint4* in, float* out, int iters) {
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p) {
    a[p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * p) & 1023]);
    b[p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * p + 333) & 1023]);
  }
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c % 3], b[(c + 1) % 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k_bf16_16(const uint4* in, float* out, int iters) {
  bf16x8 a[3], b[3];
  for (int p = 0; p < 3; ++p) {
    a[p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * p) & 1023]);
    b[p] = __builtin_bit_cast(bf16x8, in[(threadIdx.x + 64 * p + 333) & 1023]);
  }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c % 3], b[(c + 1) % 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void k_f32(const float* in, float* out, int iters) {
  float a = in[threadIdx.x], b = in[threadIdx.x + 256];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}
int main() {
  std::vector<unsigned> h(4096);
  srand(1);
  for (auto& v : h) {  // random bf16 pairs in [0.5, 2)
    unsigned lo = 0x3f00 + (rand() & 0xff), hi = 0x3f00 + (rand() & 0xff);
    v = lo | (hi << 16);
  }
  std::vector<float> hf(1024);
  for (auto& v : hf) v = 0.5f + (rand() % 1000) / 1000.f;
  uint4* din; float* dinf; float* dout;
  hipMalloc(&din, 4096 * 4); hipMalloc(&dinf, 4096); hipMalloc(&dout, 1 << 24);
  hipMemcpy(din, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dinf, hf.data(), 4096, hipMemcpyHostToDevice);
  const int iters = 2000;
  for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
    const int blocks = 256 * wg_per_cu;  // 4 waves per workgroup: 1 or 2 waves per SIMD
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL((k_bf16<2, 6>), dim3(blocks), dim3(256), 0, 0, din, dout, iters); });
    printf("bf16 32x32x16, 2 acc x chain 6, %d waves/SIMD: %.1f TFLOP/s\n", wg_per_cu, blocks * 4.0 * iters * 12 * 32768.0 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL((k_bf16<4, 1>), dim3(blocks), dim3(256), 0, 0, din, dout, iters * 3); });
    printf("bf16 32x32x16, 4 acc x chain 1, %d waves/SIMD: %.1f TFLOP/s\n", wg_per_cu, blocks * 4.0 * iters * 3 * 4 * 32768.0 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL((k_bf16_16<8, 6>), dim3(blocks), dim3(256), 0, 0, din, dout, iters); });
    printf("bf16 16x16x32, 8 acc x chain 6, %d waves/SIMD: %.1f TFLOP/s\n", wg_per_cu, blocks * 4.0 * iters * 48 * 16384.0 / (ms * 1e-3) / 1e12);
    ms = time_ms([&] { hipLaunchKernelGGL((k_f32<2, 8>), dim3(blocks), dim3(256), 0, 0, dinf, dout, iters); });
    printf("f32 32x32x2, 2 acc x chain 8, %d waves/SIMD: %.1f TFLOP/s\n", wg_per_cu, blocks * 4.0 * iters * 16 * 4096.0 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
