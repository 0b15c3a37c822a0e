// Issue cost (cycles per wave-instruction, one wave per SIMD, independent instructions) of the vector instructions the
// NLM kernel is made of, and of the candidates to replace them (instruction census).
// 8 independent chains x 64 repeats per loop iteration, timed with s_memtime; prints cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define KERNEL(NAME, ASM)                                                                                   \
  __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned long long* cyc, unsigned seed) {  \
    unsigned v0 = seed + threadIdx.x, v1 = v0 * 3, v2 = v0 * 5, v3 = v0 * 7, v4 = v0 * 9, v5 = v0 * 11, v6 = v0 * 13, v7 = v0 * 15; \
    unsigned a = seed * 17 + threadIdx.x, b = seed * 19 + 3;                                                \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                            \
    for (int it = 0; it < 256; ++it) {                                                                      \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                       \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                                \
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)       \
                     : "v"(a), "v"(b));                                                                     \
      }                                                                                                     \
    }                                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                            \
    out[blockIdx.x * 256 + threadIdx.x] = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5 ^ v6 ^ v7;                           \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                        \
  }
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define A_DOT2(i) "v_dot2_u32_u16 %" #i ", %8, %9, %" #i "\n"
#define A_DOT4(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n"
#define A_MADU16(i) "v_mad_u32_u16 %" #i ", %8, %9, %" #i "\n"
#define A_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n"
#define A_PKADDC(i) "v_pk_add_u16 %" #i ", %" #i ", %8 clamp\n"
#define A_PKMUL(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n"
#define A_PKMAD(i) "v_pk_mad_u16 %" #i ", %8, %9, %" #i "\n"
#define A_PKMIN(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"
#define A_PKSHR(i) "v_pk_lshrrev_b16 %" #i ", 6, %" #i "\n"
#define A_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 16\n"
#define A_SDWASUB(i) "v_sub_u16_sdwa %" #i ", %8, %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_1\n"
#define A_SDWAMUL(i) "v_mul_u32_u24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %8, %9, %" #i "\n"
#define A_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 6, 10\n"
#define A_DOT2BF(i) "v_dot2c_f32_bf16 %" #i ", %8, %9\n"
#define A_SUBF(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define A_FMAF(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 16, %" #i "\n"
#define A_CVTPKBF(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %8\n"
#define A_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
KERNEL(dot2c_f32_bf16, A_DOT2BF) KERNEL(sub_f32, A_SUBF) KERNEL(fma_f32, A_FMAF) KERNEL(and_b32, A_AND) KERNEL(lshlrev_b32, A_LSHL)
KERNEL(cvt_pk_bf16_f32, A_CVTPKBF) KERNEL(max_i32, A_MAXI)
KERNEL(add_u32, A_ADD) KERNEL(mad_u32_u24, A_MAD24) KERNEL(mul_u32_u24, A_MUL24) KERNEL(mul_lo_u32, A_MULLO) KERNEL(perm_b32, A_PERM)
KERNEL(dot2_u32_u16, A_DOT2) KERNEL(dot4_u32_u8, A_DOT4) KERNEL(mad_u32_u16, A_MADU16) KERNEL(pk_add_u16, A_PKADD)
KERNEL(pk_add_u16_clamp, A_PKADDC) KERNEL(pk_mul_lo_u16, A_PKMUL) KERNEL(pk_mad_u16, A_PKMAD) KERNEL(pk_min_u16, A_PKMIN)
KERNEL(pk_lshrrev_b16, A_PKSHR) KERNEL(alignbit_b32, A_ALIGNBIT) KERNEL(sub_u16_sdwa_bytes, A_SDWASUB)
KERNEL(mul_u32_u24_sdwa_byte, A_SDWAMUL) KERNEL(add3_u32, A_ADD3) KERNEL(lshl_add_u32, A_LSHLADD) KERNEL(sad_u8, A_SAD) KERNEL(bfe_u32, A_BFE)
int main() {
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8192);
  std::vector<unsigned long long> h(1024);
#define RUN(NAME, WAVES)                                                                         \
  for (int wg = 1; wg <= 4; wg *= 2) {                                                           \
    hipLaunchKernelGGL(k_##NAME, dim3(256 * wg), dim3(256), 0, 0, out, cyc, 12345u);             \
    hipLaunchKernelGGL(k_##NAME, dim3(256 * wg), dim3(256), 0, 0, out, cyc, 12345u);             \
    hipDeviceSynchronize();                                                                      \
    hipMemcpy(h.data(), cyc, 256 * wg * 8, hipMemcpyDeviceToHost);                               \
    double s = 0; for (int i = 0; i < 256 * wg; ++i) s += h[i];                                  \
    printf("%-24s %d wave(s)/SIMD: %.2f cycles per wave-instruction (x waves = SIMD cycles per instruction: %.2f)\n", #NAME, wg, \
           s / (256 * wg) / (256.0 * 8 * 8), s / (256 * wg) / (256.0 * 8 * 8) / wg);             \
  }
  RUN(add_u32, 1) RUN(mad_u32_u24, 1) RUN(mul_u32_u24, 1) RUN(mul_lo_u32, 1) RUN(perm_b32, 1) RUN(dot2_u32_u16, 1) RUN(dot4_u32_u8, 1)
  RUN(mad_u32_u16, 1) RUN(pk_add_u16, 1) RUN(pk_add_u16_clamp, 1) RUN(pk_mul_lo_u16, 1) RUN(pk_mad_u16, 1) RUN(pk_min_u16, 1)
  RUN(pk_lshrrev_b16, 1) RUN(alignbit_b32, 1) RUN(sub_u16_sdwa_bytes, 1) RUN(mul_u32_u24_sdwa_byte, 1) RUN(add3_u32, 1)
  RUN(lshl_add_u32, 1) RUN(sad_u8, 1) RUN(bfe_u32, 1)
  RUN(dot2c_f32_bf16, 1) RUN(sub_f32, 1) RUN(fma_f32, 1) RUN(and_b32, 1) RUN(lshlrev_b32, 1) RUN(cvt_pk_bf16_f32, 1) RUN(max_i32, 1)
  return 0;
}
