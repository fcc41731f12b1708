// Experiment (run by tools/exp_valu_rates.py on the GPU box): issue rate of the VALU / transcendental / MFMA instructions the
// softmax and epilogue models of DESIGN.md are built on.  One block per CU, W waves per SIMD; every wave runs ITER rounds of
// 8 independent dependency chains of ONE instruction (inline asm, so the compiler cannot fuse or drop them) and reports
// cycles per instruction per wave (s_memtime).  With 1 wave per SIMD the number is the instruction's issue interval; with
// 2 or 4 it shows whether two waves' instructions of that kind overlap.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHAINS8(ASM, C0, C1, C2, C3, C4, C5, C6, C7) \
  asm volatile(ASM : "+v"(C0));                      \
  asm volatile(ASM : "+v"(C1));                      \
  asm volatile(ASM : "+v"(C2));                      \
  asm volatile(ASM : "+v"(C3));                      \
  asm volatile(ASM : "+v"(C4));                      \
  asm volatile(ASM : "+v"(C5));                      \
  asm volatile(ASM : "+v"(C6));                      \
  asm volatile(ASM : "+v"(C7))

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* cyc, int iters) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = 0.001f * (threadIdx.x + 1) + i;
  f32x4 acc4[4] = {};
  f32x16 acc16[2] = {};
  f16x8 a8 = {(f16)1, (f16)2, (f16)3, (f16)4, (f16)5, (f16)6, (f16)7, (f16)8};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { CHAINS8("v_exp_f32 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 1) { CHAINS8("v_rcp_f32 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 2) { CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 3) { CHAINS8("v_pk_fma_f16 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 4) { CHAINS8("v_dot2_f32_f16 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 5) { CHAINS8("v_cvt_pkrtz_f16_f32 %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 6) { CHAINS8("v_max3_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 7) { CHAINS8("v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 8) { CHAINS8("v_exp_f16 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 9) { CHAINS8("v_pk_mul_f32 %0, %0, %0", *(double*)&c[0], *(double*)&c[2], *(double*)&c[4], *(double*)&c[6],
                           *(double*)&c[0], *(double*)&c[2], *(double*)&c[4], *(double*)&c[6]); }
    if (OP == 10) {   // 8 independent 16x16x32 MFMAs (4 accumulators, twice)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, acc4[k], 0, 0, 0);
    }
    if (OP == 11) {   // 8 MFMAs 32x32x16 on 2 accumulators
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc16[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k], 0, 0, 0);
    }
    if (OP == 12) {   // the softmax inner mix of the reference attention: 2 fma + 2 exp + 1 cvt_pk per pair of scores
      CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      CHAINS8("v_exp_f32 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(c[0]) : "v"(c[1]));
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(c[2]) : "v"(c[3]));
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(c[4]) : "v"(c[5]));
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(c[6]) : "v"(c[7]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc4[k][0];
  s += acc16[0][0] + acc16[1][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

extern "C" int exp_rate(int op, int blocks, int threads, int iters, float* out, unsigned long long* cyc, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(N) case N: hipLaunchKernelGGL(rate_kernel<N>, dim3(blocks), dim3(threads), 0, s, out, cyc, iters); break;
  switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) default: return -1; }
#undef L
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
