// Two questions behind CPX_CNN_MATH_FP16X2, asked of the hardware:
//  (1) does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (the low plane of a small operand is subnormal)?
//  (2) what does a guarded launch cost when the guard says "nothing to do": the full grid of a stage-2 layer
//      (409,600 workgroups of 512 threads, 79 KB LDS, exit after one scalar load) against a persistent grid (512)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_sub(float a_val, float b_val, float* out) {
  h8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
  // A row i16 = lane & 15, k = 8 * (lane >> 4) + j; one non-zero k per row
  if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_guard(const int* flag, float* out, int total) {
  extern __shared__ float lds[];
  if (*flag == 0) return;
  for (int b = blockIdx.x; b < total; b += gridDim.x) {
    lds[threadIdx.x] = (float)b;
    __syncthreads();
    out[threadIdx.x] = lds[(threadIdx.x + 1) & 511];
  }
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  float host[64];
  const float cases[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1.0f}, {3.0e-6f, 1024.0f}, {5.96e-8f, 4096.0f}, {6.0e-5f, 6.0e-5f}, {3.0e-6f, 3.0e-6f}};
  for (auto& cs : cases) {
    hipLaunchKernelGGL(k_sub, dim3(1), dim3(64), 0, 0, cs[0], cs[1], out);
    hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
    const double want = (double)(float)(_Float16)cs[0] * (double)(float)(_Float16)cs[1];
    printf("subnormal probe: a = %.4g (fp16 %.6g) b = %.4g -> mfma %.9g, exact product of the fp16 values %.9g\n", cs[0],
           (double)(float)(_Float16)cs[0], cs[1], host[0], want);
  }
  int* flag;
  hipMalloc(&flag, 4);
  hipMemset(flag, 0, 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_guard), hipFuncAttributeMaxDynamicSharedMemorySize, 79 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grids[] = {409600, 102400, 25600, 4096, 1024, 512, 256};
  for (int g : grids) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_guard, dim3(g), dim3(512), 79 * 1024, 0, flag, out, 409600);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("guarded no-op launch, grid %6d x 512 threads, 79 KB LDS: %.1f us per launch\n", g, ms * 1000 / 20);
    }
  }
  return 0;
}
