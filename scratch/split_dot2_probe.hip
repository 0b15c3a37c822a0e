// Is v_dot2c_f32_bf16 an exact way to subtract a bf16 plane from a float32?  (conv_bf3*: split_pair.)
// r = x - bf(p.lo) as dot2c(acc = x, (-1, 0), p): compares with the shift / subtract form on random, tiny, huge and
// denormal inputs, for the truncating pack (v_perm) and for v_cvt_pk_bf16_f32 (round to nearest even).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_trunc(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sub_lo(float x, unsigned p) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, 0x0000bf80u), __builtin_bit_cast(bf16x2, p), x, false);
}
__device__ __forceinline__ float sub_hi(float x, unsigned p) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, 0xbf800000u), __builtin_bit_cast(bf16x2, p), x, false);
}
template <bool RNE>
__global__ void k(const float* x, int n, unsigned* planes_ref, unsigned* planes_dot, unsigned* bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float a = x[2 * i], b = x[2 * i + 1];
  unsigned p[3], d[3];
  {
    p[0] = RNE ? pack_rne(a, b) : pack_trunc(a, b);
    const float ra = a - __uint_as_float(p[0] << 16), rb = b - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = RNE ? pack_rne(ra, rb) : pack_trunc(ra, rb);
    const float sa = ra - __uint_as_float(p[1] << 16), sb = rb - __uint_as_float(p[1] & 0xffff0000u);
    p[2] = RNE ? pack_rne(sa, sb) : pack_trunc(sa, sb);
  }
  {
    d[0] = RNE ? pack_rne(a, b) : pack_trunc(a, b);
    const float ra = sub_lo(a, d[0]), rb = sub_hi(b, d[0]);
    d[1] = RNE ? pack_rne(ra, rb) : pack_trunc(ra, rb);
    const float sa = sub_lo(ra, d[1]), sb = sub_hi(rb, d[1]);
    d[2] = RNE ? pack_rne(sa, sb) : pack_trunc(sa, sb);
  }
  if (!RNE && p[1] != d[1] && atomicAdd(bad + 1, 0u) == 0 && i < 64) {
    const float ra = a - __uint_as_float(p[0] << 16), rb = b - __uint_as_float(p[0] & 0xffff0000u);
    printf("i %d a %08x b %08x p0 %08x ra %08x rb %08x  dot ra %08x rb %08x  p1 %08x d1 %08x\n", i, __float_as_uint(a), __float_as_uint(b), p[0],
           __float_as_uint(ra), __float_as_uint(rb), __float_as_uint(sub_lo(a, d[0])), __float_as_uint(sub_hi(b, d[0])), p[1], d[1]);
  }
  for (int j = 0; j < 3; ++j) {
    planes_ref[3 * i + j] = p[j];
    planes_dot[3 * i + j] = d[j];
    if (p[j] != d[j]) atomicAdd(bad, 1u);
  }
  // the three planes must add up to the input exactly
  const float sum_a = (__uint_as_float(d[2] << 16) + __uint_as_float(d[1] << 16)) + __uint_as_float(d[0] << 16);
  if (sum_a != a && a == a && fabsf(a) < 3e38f && fabsf(a) > 1e-30f) atomicAdd(bad + 1, 1u);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> h(n);
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.0f, 1.0f);
  for (int i = 0; i < n; ++i) {
    const int kind = i & 7;
    if (kind < 4) h[i] = nd(rng) * (kind == 0 ? 1.0f : kind == 1 ? 255.0f : kind == 2 ? 1e-3f : 1e4f);
    else if (kind == 4) h[i] = std::ldexp(nd(rng), -120 - (i >> 3) % 29);  // towards and into the denormals
    else if (kind == 5) h[i] = std::ldexp(nd(rng), 100 + (i >> 3) % 27);
    else if (kind == 6) h[i] = 0.0f;
    else { uint32_t u = rng(); if (((u >> 23) & 255) == 255) u &= 0x807fffffu; std::memcpy(&h[i], &u, 4); }
  }
  float* x; unsigned *pr, *pd, *bad;
  hipMalloc(&x, n * 4); hipMalloc(&pr, n / 2 * 12); hipMalloc(&pd, n / 2 * 12); hipMalloc(&bad, 8);
  hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int rne = 0; rne < 2; ++rne) {
    hipMemset(bad, 0, 8);
    if (rne) hipLaunchKernelGGL(k<true>, dim3(n / 2 / 256), dim3(256), 0, 0, x, n, pr, pd, bad);
    else hipLaunchKernelGGL(k<false>, dim3(n / 2 / 256), dim3(256), 0, 0, x, n, pr, pd, bad);
    unsigned hb[2];
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    printf("%s pack: %u of %d plane words differ between subtract and dot2c; %u of %d normal inputs not the exact sum of their planes\n",
           rne ? "round-to-nearest" : "truncating", hb[0], n / 2 * 3, hb[1], n / 2);
  }
  return 0;
}
