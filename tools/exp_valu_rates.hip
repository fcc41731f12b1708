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
__global__ __launch_bounds__(1024) void rate_kernel(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = 0.001f * (threadIdx.x + 1) + i;
  f32x4 acc4[4] = {};
  f32x16 acc16[2] = {};
  f32x16 accA = {}, accB = {};
  f16x8 a8 = {(f16)1, (f16)2, (f16)3, (f16)4, (f16)5, (f16)6, (f16)7, (f16)8};
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
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
    if (OP == 13) { CHAINS8("v_permlane32_swap_b32 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 14) { CHAINS8("v_permlane16_swap_b32 %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
    if (OP == 15) { CHAINS8("v_pk_fma_f32 %0, %0, %0, %0", *(double*)&c[0], *(double*)&c[2], *(double*)&c[4], *(double*)&c[6],
                            *(double*)&c[0], *(double*)&c[2], *(double*)&c[4], *(double*)&c[6]); }
    // same-wave overlap: independent VALU work written between independent MFMAs (cycles per ITERATION reported)
    if (OP == 16) {   // 2 x 32x32x16 + 8 v_exp (4 behind each)
      acc16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[0], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
      acc16[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[1], 0, 0, 0);
      asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
    }
    if (OP == 17) {   // 2 x 32x32x16 + 16 v_fma (8 behind each)
      acc16[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[0], 0, 0, 0);
      CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      acc16[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[1], 0, 0, 0);
      CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
    }
    if (OP == 18) {   // 4 x 16x16x32 + 8 v_exp (2 behind each)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc4[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, acc4[k], 0, 0, 0);
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1" : "+v"(c[2 * k]), "+v"(c[2 * k + 1]));
      }
    }
    if (OP == 19) {   // one attention unit's instruction multiset, MFMA and VALU interleaved 1 : ~5:
                      // 6 x 32x32x16 + 12 x 16x16x32 with 32 fma + 32 exp + 16 cvt_pk + 16 max3 spread behind them
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k & 1], 0, 0, 0);
        CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        acc4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, acc4[k & 3], 0, 0, 0);
        if (k < 8) {
          asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
          asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
          asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        }
      }
    }
    if (OP == 20) {   // the same VALU multiset alone (no MFMA)
#pragma unroll
      for (int k = 0; k < 6; ++k) { CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]); }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
      }
    }
    if (OP == 22) {   // legacy K = 8 form: 8 x v_mfma_f32_32x32x8_f16 on 2 accumulators
      typedef f16 f16x4_ __attribute__((ext_vector_type(4)));
      const f16x4_ a4 = {(f16)1, (f16)2, (f16)3, (f16)4};
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 2; ++k) acc16[k] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, a4, acc16[k], 0, 0, 0);
    }
    if (OP == 23) {   // legacy 16x16x16: 8 on 4 accumulators
      typedef f16 f16x4_ __attribute__((ext_vector_type(4)));
      const f16x4_ a4 = {(f16)1, (f16)2, (f16)3, (f16)4};
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc4[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, acc4[k], 0, 0, 0);
    }
    if (OP == 24) {   // serial chain: 8 DEPENDENT v_max3 (latency, not issue rate)
      asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %0, %0, %3, %4\n v_max3_f32 %0, %0, %4, %5\n"
                   "v_max3_f32 %0, %0, %5, %6\n v_max3_f32 %0, %0, %6, %7\n v_max3_f32 %0, %0, %7, %1\n v_max3_f32 %0, %0, %1, %3"
                   : "+v"(c[0]) : "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
    }
    // the planned reference-attention unit (32 queries x 64 keys at d = 40, scale and running max folded into the
    // contraction): 14 x 32x32x16 + 32 exp + 16 cvt_pk + 16 max3 + 8 fma, ~5 VALU behind every MFMA
    if (OP == 25 || OP == 26 || OP == 27) {
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        if (OP == 25) acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k & 1], 0, 0, 0);
        if (OP == 26) {   // the 8 P V MFMAs accumulate in AGPRs, the 6 score MFMAs in VGPRs
          if (k < 6) acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k & 1], 0, 0, 0);
          else if (k & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accA) : "v"(a8));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accB) : "v"(a8));
        }
        if (OP == 27) {
          if (k & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accA) : "v"(a8));
          else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accB) : "v"(a8));
        }
        if (k < 8) {
          asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
          asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else if (k < 12) {
          asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %1, %1, %0, %2\n v_max3_f32 %2, %2, %1, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else {
          asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        }
      }
    }
    if (OP == 28) {   // that VALU multiset alone
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        if (k < 8) {
          asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
          asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else if (k < 12) {
          asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %1, %1, %0, %2\n v_max3_f32 %2, %2, %1, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else {
          asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        }
      }
    }
    if (OP == 29) {   // 2 x 32x32x16 (AGPR accumulators) + 16 v_fma: op 17 with the accumulators out of the VGPR file
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accA) : "v"(a8));
      CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, %0" : "+a"(accB) : "v"(a8));
      CHAINS8("v_fma_f32 %0, %0, %0, %0", c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
    }
    if (OP == 30 || OP == 31) {   // PHASED unit (no in-wave interleave): 6 score MFMAs | the VALU block | 8 P V MFMAs (AGPR accumulators)
#pragma unroll
      for (int k = 0; k < 6; ++k) acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k & 1], 0, 0, 0);
      if (OP == 31) { c[4] += acc16[0][3]; c[5] += acc16[1][7]; }   // the softmax waits for the scores
#pragma unroll
      for (int k = 0; k < 14; ++k) {
        if (k < 8) {
          asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
          asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1\n v_cvt_pkrtz_f16_f32 %2, %2, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else if (k < 12) {
          asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %1, %1, %0, %2\n v_max3_f32 %2, %2, %1, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        } else {
          asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
        }
      }
      f16x8 pb = a8;
      if (OP == 31) pb[0] = (f16)c[4];                              // the P V MFMAs wait for the probabilities
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(accA) : "v"(a8), "v"(pb));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(accB) : "v"(a8), "v"(pb));
      }
    }
    if (OP == 21) {   // the same MFMA multiset alone
#pragma unroll
      for (int k = 0; k < 6; ++k) acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, a8, acc16[k & 1], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < 12; ++k) acc4[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, a8, acc4[k & 3], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += c[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc4[k][0];
  s += acc16[0][0] + acc16[1][0] + accA[0] + accB[0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0; rt[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r1 - r0; }
}

extern "C" int exp_rate(int op, int blocks, int threads, int iters, float* out, unsigned long long* cyc, unsigned long long* rt, void* stream) {
  hipStream_t s = (hipStream_t)stream;
#define L(N) case N: hipLaunchKernelGGL(rate_kernel<N>, dim3(blocks), dim3(threads), 0, s, out, cyc, rt, iters); break;
  switch (op) { L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(17) L(18) L(19) L(20) L(21) L(22) L(23) L(24) L(25) L(26) L(27) L(28) L(29) L(30) L(31) default: return -1; }
#undef L
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
