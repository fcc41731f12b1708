// LAB: next-generation GEMM/conv kernel (glds 3-stage ring, 256xBN tile, 8 waves).  Standalone harness.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
union U4H8 { u32x4 u; f16x8 h; f16 e[8]; };

struct GP {
  const f16* A; int64_t lda; const f16* W; int64_t ldw; f16* out; int64_t ldo;
  int M, N, K; const float* bias;
};

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int BM = 256, NT2 = 512;

template <int BN, int BK, int MINW, int ABL, int PRIO, int EPI2 = 0>
__global__ __launch_bounds__(NT2, MINW) void gemm2_kernel(const GP p) {
  constexpr int RB = BK * 2;                        // LDS row bytes (128 or 64)
  constexpr int RPI = 1024 / RB;                    // rows per glds instruction (8 or 16)
  constexpr int NCHK = RB / 16;                     // 16-B chunks per row
  constexpr int NB = BN / 32;                       // 16-col MFMA tiles per wave along N (wave tile 64 x BN/2)
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA_I = BM / RPI / 8;                // A glds instructions per wave per K-tile
  constexpr int NB_TOT = BN / RPI;                  // B glds instructions per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN, nblk = nbm * nbn;
  int swz;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = swz / nbn, bn = swz % nbn, m0 = bm * BM, n0 = bn * BN;

  // buffer resources (wave-uniform): OOB lanes (voffset >= num_records) load zeros into LDS
  const uint32_t a_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, ((int64_t)(p.M - 1) * p.lda + p.K) * 2);
  const uint32_t w_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, ((int64_t)(p.N - 1) * p.ldw + p.K) * 2);
  auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
  constexpr uint32_t OOB = 0xFFFFFFF0u;

  // per-lane global byte offsets of the 16-B chunk this lane feeds for each of its glds instructions
  const int lr = lane / NCHK, ls = lane % NCHK;
  auto swz_of = [](int row) { return (RB == 128) ? ((row >> 1) & 7) : (((row >> 3) & 1) * 2); };
  uint32_t a_off[NA_I];
#pragma unroll
  for (int i = 0; i < NA_I; ++i) {
    const int row = (wave * NA_I + i) * RPI + lr;     // row in the A tile
    const int g = ls ^ swz_of(row);
    const int m = m0 + row;
    a_off[i] = (m < p.M) ? (uint32_t)(((int64_t)m * p.lda + g * 8) * 2) : OOB;
  }
  // B instructions are dealt round-robin: instruction j (0..NB_TOT-1) belongs to wave j % 8
  constexpr int NB_I = (NB_TOT + 7) / 8;
  uint32_t b_off[NB_I];
#pragma unroll
  for (int i = 0; i < NB_I; ++i) {
    const int j = wave + 8 * i;
    const int row = j * RPI + lr;
    const int g = ls ^ swz_of(row);
    const int n = n0 + row;
    b_off[i] = (j < NB_TOT && n < p.N) ? (uint32_t)(((int64_t)n * p.ldw + g * 8) * 2) : OOB;
  }
  const int my_b = (NB_TOT - wave + 7) / 8;           // B instructions this wave really issues (wave-uniform)
  const int gch = ls ^ 0;                             // (k-tail handled by OOB below)

  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const uint32_t koff = (uint32_t)kt * (BK * 2);
    const bool ktail = (kt + 1) * BK > p.K;           // wave-uniform
#pragma unroll
    for (int i = 0; i < NA_I; ++i) {
      uint32_t vo = a_off[i];
      if (ktail) { const int row = (wave * NA_I + i) * RPI + lr; const int g = ls ^ swz_of(row); if (kt * BK + g * 8 >= p.K) vo = OOB; }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sa + (wave * NA_I + i) * 1024), 16, vo, koff, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB_I; ++i) {
      if (i < my_b) {
        uint32_t vo = b_off[i];
        if (ktail) { const int row = (wave + 8 * i) * RPI + lr; const int g = ls ^ swz_of(row); if (kt * BK + g * 8 >= p.K) vo = OOB; }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(sb + (wave + 8 * i) * 1024), 16, vo, koff, 0, 0);
      }
    }
  };
  (void)gch;

  const int fr = lane & 15, fq = lane >> 4;
  const int sw_r = swz_of(fr);
  const int a_row_off = (wm * 64 + fr) * RB;
  const int b_row_off = (wn * (BN / 2) + fr) * RB;

  f32x4 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (ABL == 2) ? 0 : (p.K + BK - 1) / BK;
  if (PRIO == 2 && (((blockIdx.x >> 3) >> 5) & 1)) {
    // de-phase the two co-resident blocks of a CU: the second starts half a tile later, so that one
    // block's store-bound epilogue overlaps the other's MFMA-bound main loop from then on
    for (int i = 0; i < nk * (BK / 32); ++i) __builtin_amdgcn_s_sleep(20);
  }
  if (nk > 0) issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt landed (this wave's part): leave only tile kt+1's loads in flight
    if (kt + 1 < nk) {
      if (my_b == NB_I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_I + NB_I) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_I + NB_I - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) issue(kt + 2, (kt + 2) % 3);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    const char* sa = smem + (kt % 3) * STAGE;
    const char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      const int koff = (((ks * 4 + fq) ^ sw_r) << 4);
      f16x8 af[4], bf[NB];
#pragma unroll
      for (int t = 0; t < 4; ++t) af[t] = *(const f16x8*)(sa + a_row_off + t * 16 * RB + koff);
#pragma unroll
      for (int t = 0; t < NB; ++t) bf[t] = *(const f16x8*)(sb + b_row_off + t * 16 * RB + koff);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }

  if (ABL == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if (EPI2) {
    // ---- wave-private epilogue: each wave stages 16 rows x WN cols of its own tile (fp32) and writes
    // full rows; no block barriers after the first one
    constexpr int WN = BN / 2, CSW = WN + 4, NCH = WN / 8;
    constexpr int RPS = 64 / NCH;                    // rows per store step (8 for WN=64, 6 for WN=80)
    float* ws = (float*)smem + wave * (16 * CSW);
    const int cc = lane % NCH, rr = lane / NCH;
    const bool lactive = lane < RPS * NCH;
    const int ncol = n0 + wn * WN + cc * 8;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (p.bias != nullptr && lactive) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (ncol + e < p.N) bv[e] = p.bias[ncol + e];
    }
    const int nvalid = min(8, p.N - ncol);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[(fq * 4 + r) * CSW + j * 16 + fr] = acc[i][j][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lactive) {
#pragma unroll
        for (int r = rr; r < 16; r += RPS) {
          const int m = m0 + wm * 64 + i * 16 + r;
          if (m < p.M && nvalid > 0) {
            const float4 v0 = *(const float4*)(ws + r * CSW + cc * 8), v1 = *(const float4*)(ws + r * CSW + cc * 8 + 4);
            const float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
            f16* op = p.out + (int64_t)m * p.ldo + ncol;
            if (nvalid == 8 && ((p.ldo & 7) == 0)) {
              U4H8 t;
#pragma unroll
              for (int e = 0; e < 8; ++e) t.e[e] = (f16)v[e];
              *(u32x4*)op = t.u;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (e < nvalid) op[e] = (f16)v[e];
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  // ---- epilogue: two passes of 128 rows through fp32 LDS staging ---------------------------------
  constexpr int CS = BN + 4;
  float* cs = (float*)smem;
  constexpr int NCHUNK = BN / 8;
  constexpr int RSTEP = NT2 / NCHUNK;
  const int cc = tid % NCHUNK, r0 = tid / NCHUNK;
  const bool tactive = tid < RSTEP * NCHUNK;
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = 0.f;
  const int ncol = n0 + cc * 8;
  if (p.bias != nullptr && tactive) {
#pragma unroll
    for (int e = 0; e < 8; ++e) if (ncol + e < p.N) bv[e] = p.bias[ncol + e];
  }
  for (int pass = 0; pass < 4; ++pass) {
    __builtin_amdgcn_s_barrier();  // previous readers of the LDS region are done (K loop / previous pass)
    if (wm == pass) {
      const int wr = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cs[(wr + i * 16 + fq * 4 + r) * CS + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tactive) {
      for (int r = r0; r < 64; r += RSTEP) {
        const int m = m0 + pass * 64 + r;
        if (m >= p.M) break;
        const float4 v0 = *(const float4*)(cs + r * CS + cc * 8), v1 = *(const float4*)(cs + r * CS + cc * 8 + 4);
        const float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3], v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
        f16* op = p.out + (int64_t)m * p.ldo + ncol;
        const int nvalid = min(8, p.N - ncol);
        if (nvalid == 8 && ((p.ldo & 7) == 0)) {
          U4H8 t;
#pragma unroll
          for (int e = 0; e < 8; ++e) t.e[e] = (f16)v[e];
          *(u32x4*)op = t.u;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) if (e < nvalid) op[e] = (f16)v[e];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
__global__ void ref_kernel(const f16* A, const f16* W, const float* bias, float* out, int M, int N, int K) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(int64_t)m * K + k] * (float)W[(int64_t)n * K + k];
  out[(int64_t)m * N + n] = s + (bias ? bias[n] : 0.f);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int BN, int BK, int MINW, int ABL, int PRIO, int EPI2 = 0>
void launch(const GP& p, hipStream_t s) {
  constexpr int STAGE = (BM + BN) * BK * 2;
  constexpr int EPI = 64 * (BN + 4) * 4;
  constexpr int LDS = (3 * STAGE > EPI) ? 3 * STAGE : EPI;
  static bool attr = false;
  if (!attr) { auto kfn = gemm2_kernel<BN, BK, MINW, ABL, PRIO, EPI2>; CK(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); attr = true; }
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm2_kernel<BN, BK, MINW, ABL, PRIO, EPI2>), dim3(nbm * nbn), dim3(NT2), LDS, s, p);
}

static void fill(std::vector<f16>& v, float scale, unsigned seed) {
  unsigned s = seed * 2654435761u + 12345u;
  for (auto& x : v) { s = s * 1664525u + 1013904223u; x = (f16)((((s >> 8) & 0xFFFF) / 32768.0f - 1.0f) * scale); }
}

int run_case(int M, int N, int K, int BN, bool check, int iters, int var = 0) {
  const int abl = var;
  std::vector<f16> hA((size_t)M * K), hW((size_t)N * K);
  std::vector<float> hb(N);
  fill(hA, 1.0f, 1 + M); fill(hW, 1.0f / sqrtf((float)K), 2 + N);
  for (int i = 0; i < N; ++i) hb[i] = 0.01f * (i % 37) - 0.1f;
  f16 *dA, *dW, *dO; float *db, *dR;
  CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&dO, (size_t)M * N * 2));
  CK(hipMalloc(&db, N * 4));
  CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dO, 0xFF, (size_t)M * N * 2));
  GP p{dA, K, dW, K, dO, N, M, N, K, db};
  auto go = [&]() {
    switch (var) {
      case 0: if (BN == 128) launch<128, 64, 2, 0, 0>(p, 0); else launch<160, 64, 2, 0, 0>(p, 0); break;
      case 1: if (BN == 128) launch<128, 32, 4, 0, 0>(p, 0); else launch<160, 32, 4, 0, 0>(p, 0); break;
      case 2: if (BN == 128) launch<128, 64, 2, 0, 1>(p, 0); else launch<160, 64, 2, 0, 1>(p, 0); break;
      case 3: if (BN == 128) launch<128, 32, 4, 0, 1>(p, 0); else launch<160, 32, 4, 0, 1>(p, 0); break;
      case 4: if (BN == 128) launch<128, 32, 4, 1, 0>(p, 0); else launch<160, 32, 4, 1, 0>(p, 0); break;
      case 5: if (BN == 128) launch<128, 32, 4, 0, 0, 1>(p, 0); else launch<160, 32, 4, 0, 0, 1>(p, 0); break;
      case 6: if (BN == 128) launch<128, 32, 4, 0, 2, 1>(p, 0); else launch<160, 32, 4, 0, 2, 1>(p, 0); break;
    }
  };
  go();
  CK(hipDeviceSynchronize());
  int bad = 0;
  if (check) {
    CK(hipMalloc(&dR, (size_t)M * N * 4));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, M), dim3(256), 0, 0, dA, dW, db, dR, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<f16> ho((size_t)M * N); std::vector<float> hr((size_t)M * N);
    CK(hipMemcpy(ho.data(), dO, ho.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), dR, hr.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (size_t i = 0; i < ho.size(); ++i) {
      const double e = fabs((double)(float)ho[i] - hr[i]);
      const double tol = 2e-3 * fabs(hr[i]) + 2e-3;
      if (!(e <= tol)) { if (bad < 5) printf("  mismatch at m=%zu n=%zu got %f want %f\n", i / N, i % N, (float)ho[i], hr[i]); ++bad; }
      if (e > maxerr) maxerr = e;
    }
    printf("check M=%d N=%d K=%d BN=%d: maxerr=%.3e bad=%d\n", M, N, K, BN, maxerr, bad);
    CK(hipFree(dR));
  }
  if (iters > 0) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) go();
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; ++i) go();
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= iters;
    printf("perf var=%d M=%d N=%d K=%d BN=%d: %.4f ms  %.1f TFLOP/s\n", abl, M, N, K, BN, ms, 2.0 * M * N * K / ms / 1e9);
  }
  CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dO)); CK(hipFree(db));
  return bad;
}

int main() {
  int bad = 0;
  for (int var = 6; var < 7; ++var) {
    bad += run_case(256, 128, 64, 128, true, 0, var);
    bad += run_case(256, 128, 320, 128, true, 0, var);
    bad += run_case(300, 200, 136, 128, true, 0, var);
    bad += run_case(1000, 320, 320, 160, true, 0, var);
    bad += run_case(77, 640, 1280, 160, true, 0, var);
    bad += run_case(2048, 1280, 2560, 128, true, 0, var);
    bad += run_case(512, 320, 32, 160, true, 0, var);
    bad += run_case(512, 72, 128, 128, true, 0, var);
  }
  printf("TOTAL bad=%d\n", bad);
  for (int var : {5, 6}) {
    run_case(131072, 2560, 320, 128, false, 20, var);
    run_case(131072, 320, 320, 160, false, 20, var);
    run_case(131072, 640, 320, 160, false, 20, var);
    run_case(131072, 320, 1280, 160, false, 20, var);
    run_case(32768, 640, 2560, 160, false, 20, var);
    run_case(8192, 1280, 5120, 128, false, 20, var);
    run_case(8192, 8192, 8192, 128, false, 5, var);
  }
  return bad ? 1 : 0;
}
