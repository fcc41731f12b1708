// gemm2: the main MFMA GEMM / implicit-GEMM 3x3 convolution kernel for gfx950 (CDNA4).
//
//   out[M,N] = epilogue(alpha * A[M,K] @ W[N,K]^T), fp16 in, fp32 accumulate
//
// Structure (MI355X-first, see cdna_hip_programming.md §5):
//  * 256(M) x BN(N) x 32(K) tiles, BN = 128 or 160 (160 tiles N = 320/640/960/1920 without waste),
//    512 threads = 8 waves as 4(M) x 2(N), wave tile 64 x BN/2 of v_mfma_f32_16x16x32_f16.
//  * operands go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction): no
//    staging VGPRs, no ds_write pass.  A 3-stage ring with COUNTED vmcnt keeps one K-tile in flight across
//    the single s_barrier per K-tile; out-of-range rows / K-tail chunks / conv padding are buffer-OOB
//    lanes, which the DMA fills with zeros.
//  * LDS rows are 64 B (4 chunks of 16 B); the DMA destination is lane-linear, so the bank swizzle
//    (chunk ^= 2*bit3(row)) is applied on the per-lane SOURCE address and on the ds_read_b128 side.
//  * 72-78 KB LDS and <= 128 VGPRs per block -> 2 blocks (16 waves) per CU: one block's epilogue /
//    barrier stalls overlap the other's MFMAs.
//  * A operand loaders: plain row-major (optionally two sources split along K = fused channel concat),
//    or NHWC 3x3 window gather (stride 1/2, zero padding, fused nearest-2x upsample).
//  * epilogue through fp32 LDS staging in 64-row passes, 16-B coalesced stores: bias, per-row-group
//    bias, GEGLU, residual, fp32 out, transposed out (V^T projection), single rounding to fp16.
//  * XCD-aware bijective tile order: each XCD's private L2 sees a contiguous run of tiles sharing A panels.
#include "common.h"

namespace {

constexpr int BK2 = 32;
constexpr int RB = BK2 * 2;      // LDS row bytes
constexpr int RPI = 1024 / RB;   // rows per LDS-DMA instruction (16)
constexpr uint32_t OOB = 0xFFFFFFF0u;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int swz_of(int row) { return ((row >> 3) & 1) * 2; }

// BM2 = 256 (8 waves, 2 blocks per CU) for problems with >= ~1000 tiles, BM2 = 128 (4 waves, up to 3 blocks
// per CU) to double the tile count of smaller problems.
template <int BM2, int BN, bool CONV>
__global__ __launch_bounds__(BM2 * 2, (BM2 == 256 ? 4 : 2)) void gemm2_kernel(const anip_gemm_params p) {
  constexpr int NT2 = BM2 * 2;                 // threads: one wave per 32 tile rows
  constexpr int NW = NT2 / 64;                 // waves, arranged (BM2/64) along M x 2 along N
  constexpr int NB = BN / 32;                  // 16-col MFMA tiles per wave along N
  constexpr int A_BYTES = BM2 * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA_I = BM2 / RPI / NW;         // A DMA instructions per wave per K-tile (2)
  constexpr int NB_TOT = BN / RPI;             // B DMA instructions per K-tile (8 or 10)
  constexpr int NB_I = (NB_TOT + NW - 1) / NW; // max per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int nbm = (p.M + BM2 - 1) / BM2, nbn = (p.N + BN - 1) / BN, nblk = nbm * nbn;
  int swz;
  {
    const int bid = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = swz / nbn, bn = swz % nbn, m0 = bm * BM2, n0 = bn * BN;

  const f16* Ap = (const f16*)p.A;
  const f16* Wp = (const f16*)p.W;
  if (p.batch > 1) {
    Ap += (int64_t)blockIdx.y * p.strideA;
    Wp += (int64_t)blockIdx.y * p.strideW;
  }
  // buffer resources (wave-uniform).  num_records covers the addressable extent; lanes whose voffset is
  // >= num_records (OOB) make the DMA write zeros.
  uint32_t a_bytes, a2_bytes = 0;
  if (CONV) a_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, (int64_t)p.Nimg * p.Hin * p.Win * p.Cin * 2);
  else a_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, ((int64_t)(p.M - 1) * p.lda + (p.A2 ? p.K1 : p.K)) * 2);
  if (!CONV && p.A2) a2_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, ((int64_t)(p.M - 1) * p.lda2 + (p.K - p.K1)) * 2);
  const uint32_t w_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, ((int64_t)(p.N - 1) * p.ldw + p.K) * 2);
  auto rA = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, a_bytes, 0x00020000);
  auto rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, a2_bytes, 0x00020000);
  auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, w_bytes, 0x00020000);

  // ---- per-lane DMA source bookkeeping ------------------------------------------------------------
  const int lr = lane >> 2, ls = lane & 3;     // row within the 16-row DMA group, 16-B slot within the row
  uint32_t a_off[NA_I];                        // plain: byte offset of (row, chunk g) at k = 0; conv: pixel base
  int a_g[NA_I];
  int a_y0[NA_I], a_x0[NA_I];
  bool a_ok[NA_I];
#pragma unroll
  for (int i = 0; i < NA_I; ++i) {
    const int row = (wave * NA_I + i) * RPI + lr;
    const int g = ls ^ swz_of(row);
    const int m = m0 + row;
    a_g[i] = g;
    a_ok[i] = m < p.M;
    if (CONV) {
      const int hw = p.Hout * p.Wout;
      const int mm = a_ok[i] ? m : 0;
      const int img = mm / hw, rem = mm - img * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      a_off[i] = (uint32_t)img * (uint32_t)(p.Hin * p.Win);
      a_y0[i] = oy * p.stride - p.pad;
      a_x0[i] = ox * p.stride - p.pad;
    } else {
      a_off[i] = (uint32_t)m;
      a_y0[i] = a_x0[i] = 0;
    }
  }
  uint32_t b_off[NB_I];
#pragma unroll
  for (int i = 0; i < NB_I; ++i) {
    const int j = wave + NW * i;
    const int row = j * RPI + lr;
    const int g = ls ^ swz_of(row);
    const int n = n0 + row;
    b_off[i] = (j < NB_TOT && n < p.N) ? (uint32_t)(((int64_t)n * p.ldw + g * 8) * 2) : OOB;
  }
  const int my_b = (NB_TOT - wave + NW - 1) / NW;  // B DMA instructions this wave issues (wave-uniform)

  auto issue = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE;
    char* sb = sa + A_BYTES;
    const int k0 = kt * BK2;
    const bool ktail = k0 + BK2 > p.K;         // wave-uniform
    if (CONV) {
      const int tap = k0 / p.Cin, c0 = k0 - tap * p.Cin;   // tap uniform over the K-tile (Cin % 32 == 0)
      const int dy = tap / 3, dx = tap - dy * 3;
      const int He = p.upsample ? 2 * p.Hin : p.Hin, We = p.upsample ? 2 * p.Win : p.Win;
#pragma unroll
      for (int i = 0; i < NA_I; ++i) {
        int y = a_y0[i] + dy, x = a_x0[i] + dx;
        const bool ok = a_ok[i] && y >= 0 && y < He && x >= 0 && x < We;
        if (p.upsample) { y >>= 1; x >>= 1; }
        const uint32_t vo = ok ? ((a_off[i] + (uint32_t)(y * p.Win + x)) * (uint32_t)p.Cin + (uint32_t)(c0 + a_g[i] * 8)) * 2u : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sa + (wave * NA_I + i) * 1024), 16, vo, 0, 0, 0);
      }
    } else {
      const bool second = (p.A2 != nullptr) && (k0 >= p.K1);   // wave-uniform (K1 % 32 == 0)
      const int kk = second ? k0 - p.K1 : k0;
      const int64_t ld = second ? p.lda2 : p.lda;
      const int klim = second ? p.K - p.K1 : (p.A2 ? p.K1 : p.K);
#pragma unroll
      for (int i = 0; i < NA_I; ++i) {
        uint32_t vo = a_ok[i] ? (uint32_t)(((int64_t)a_off[i] * ld + kk + a_g[i] * 8) * 2) : OOB;
        if (ktail && kk + a_g[i] * 8 >= klim) vo = OOB;
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sa + (wave * NA_I + i) * 1024), 16, vo, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sa + (wave * NA_I + i) * 1024), 16, vo, 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NB_I; ++i) {
      if (i < my_b) {
        uint32_t vo = b_off[i];
        if (ktail) {
          const int row = (wave + NW * i) * RPI + lr;
          if (k0 + (ls ^ swz_of(row)) * 8 >= p.K) vo = OOB;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(sb + (wave + NW * i) * 1024), 16, vo, (uint32_t)k0 * 2u, 0, 0);
      }
    }
  };

  // ---- fragment read offsets ------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int koff = (fq ^ swz_of(fr)) << 4;
  const int a_row_off = (wm * 64 + fr) * RB + koff;
  const int b_row_off = (wn * (BN / 2) + fr) * RB + koff;

  f32x4 acc[4][NB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK2 - 1) / BK2;
  issue(0, 0);
  if (nk > 1) issue(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's part of tile kt has landed; leave only tile kt+1's DMA in flight
    if (kt + 1 < nk) {
      if (my_b == NB_I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_I + NB_I) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA_I + NB_I - 1) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // tile kt complete for all waves; stage (kt+2)%3 no longer being read
    if (kt + 2 < nk) issue(kt + 2, (kt + 2) % 3);
    const char* sa = smem + (kt % 3) * STAGE;
    const char* sb = sa + A_BYTES;
    f16x8 af[4], bf[NB];
#pragma unroll
    for (int t = 0; t < 4; ++t) af[t] = *(const f16x8*)(sa + a_row_off + t * 16 * RB);
#pragma unroll
    for (int t = 0; t < NB; ++t) bf[t] = *(const f16x8*)(sb + b_row_off + t * 16 * RB);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
  }

  // ---- epilogue: passes of 64 rows through an fp32 LDS staging tile ---------------------------------
  constexpr int CS = BN + 4;
  float* cs = (float*)smem;
  const float alpha = p.alpha;
  const bool geglu = p.act == 1;
  const int64_t obatch = (p.batch > 1) ? (int64_t)blockIdx.y * p.strideO : 0;
  for (int pass = 0; pass < BM2 / 64; ++pass) {
    __builtin_amdgcn_s_barrier();  // readers of this LDS region (K loop / previous pass) are done
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cs[(i * 16 + fq * 4 + r) * CS + wn * (BN / 2) + j * 16 + fr] = acc[i][j][r] * alpha;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int mbase = m0 + pass * 64;
    if (p.trans_out) {
      // out[n][m]: thread -> (column n, 8 consecutive rows)
      constexpr int CPS = NT2 / BN;            // 8-row chunks per sweep (4 or 3)
      const int n = tid % BN, ch0 = tid / BN;
      if (ch0 < CPS && n0 + n < p.N) {
        const float bn_ = p.bias ? p.bias[n0 + n] : 0.f;
        for (int ch = ch0; ch < 8; ch += CPS) {
          const int m = mbase + ch * 8;
          if (m >= p.M) break;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = cs[(ch * 8 + e) * CS + n] + bn_;
          f16* op = (f16*)p.out + obatch + (int64_t)(n0 + n) * p.ldo + m;
          if (m + 8 <= p.M && ((p.ldo & 7) == 0)) {
            U4H8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t.e[e] = (f16)v[e];
            *(u32x4*)op = t.u;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (m + e < p.M) op[e] = (f16)v[e];
          }
        }
      }
    } else if (geglu) {
      // packed columns per 128-tile: [64 x value | 64 x gate] -> 64 output columns
      if (BN == 128) {
        const int cc = tid & 7;
        const int pn = n0 + cc * 8, ncol = bn * 64 + cc * 8;
        for (int row = tid >> 3; row < 64; row += NT2 / 8) {
        const int m = mbase + row;
        if (m < p.M && ncol < p.N / 2) {
          const float* hrow = cs + row * CS + cc * 8;
          const float* grow = hrow + 64;
          U4H8 t;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float hh = hrow[e], gg = grow[e];
            if (p.bias != nullptr) { hh += p.bias[pn + e]; gg += p.bias[pn + 64 + e]; }
            t.e[e] = (f16)(hh * gelu_erf_f(gg));
          }
          f16* op = (f16*)p.out + obatch + (int64_t)m * p.ldo + ncol;
          if ((p.ldo & 7) == 0) *(u32x4*)op = t.u;
          else {
#pragma unroll
            for (int e = 0; e < 8; ++e) op[e] = t.e[e];
          }
        }
        }
      }
    } else {
      constexpr int NCHUNK = BN / 8, RSTEP = NT2 / NCHUNK;
      const int cc = tid % NCHUNK, r0 = tid / NCHUNK;
      const int ncol = n0 + cc * 8;
      const int nvalid = min(8, p.N - ncol);
      if (r0 < RSTEP && nvalid > 0) {
        float bv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bv[e] = (p.bias != nullptr && e < nvalid) ? p.bias[ncol + e] : 0.f;
        for (int r = r0; r < 64; r += RSTEP) {
          const int m = mbase + r;
          if (m >= p.M) break;
          const float4 v0 = *(const float4*)(cs + r * CS + cc * 8), v1 = *(const float4*)(cs + r * CS + cc * 8 + 4);
          float v[8] = {v0.x + bv[0], v0.y + bv[1], v0.z + bv[2], v0.w + bv[3],
                        v1.x + bv[4], v1.y + bv[5], v1.z + bv[6], v1.w + bv[7]};
          if (p.rowbias != nullptr) {
            const float* rbp = p.rowbias + ((int64_t)m / p.rows_per_group) * p.ld_rowbias + ncol;
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e < nvalid) v[e] += rbp[e];
          }
          if (p.residual != nullptr) {
            const f16* rp = (const f16*)p.residual + (int64_t)m * p.ldr + ncol;
            if (nvalid == 8 && ((p.ldr & 7) == 0)) {
              U4H8 t;
              t.u = *(const u32x4*)rp;
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)t.e[e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (e < nvalid) v[e] += (float)rp[e];
            }
          }
          const int64_t o = obatch + (int64_t)m * p.ldo + ncol;
          if (p.out_f32) {
            float* op = (float*)p.out + o;
            if (nvalid == 8 && ((p.ldo & 3) == 0)) {
              *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (e < nvalid) op[e] = v[e];
            }
          } else {
            f16* op = (f16*)p.out + o;
            if (nvalid == 8 && ((p.ldo & 7) == 0)) {
              U4H8 t;
#pragma unroll
              for (int e = 0; e < 8; ++e) t.e[e] = (f16)v[e];
              *(u32x4*)op = t.u;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e)
                if (e < nvalid) op[e] = (f16)v[e];
            }
          }
        }
      }
    }
  }
}

template <int BM2, int BN, bool CONV>
int launch_gemm2(const anip_gemm_params& p, hipStream_t stream) {
  constexpr int NT2 = BM2 * 2;
  constexpr int STAGE = (BM2 + BN) * RB;
  constexpr int EPI = 64 * (BN + 4) * 4;
  constexpr int LDS = (3 * STAGE > EPI) ? 3 * STAGE : EPI;
  static bool attr_done = false;
  if (!attr_done) {
    auto kfn = gemm2_kernel<BM2, BN, CONV>;
    if (hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      anip_set_error("anip_gemm: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return -2;
    }
    attr_done = true;
  }
  const int nbm = (p.M + BM2 - 1) / BM2, nbn = (p.N + BN - 1) / BN;
  hipLaunchKernelGGL((gemm2_kernel<BM2, BN, CONV>), dim3((unsigned)(nbm * nbn), (unsigned)p.batch, 1), dim3(NT2), LDS, stream, p);
  return 1;
}

}  // namespace

// returns 1 if the problem was launched on gemm2, 0 if it is not eligible (caller falls back to the
// small-problem kernel), < 0 on error
int anip_gemm2_try(const anip_gemm_params& p, hipStream_t stream) {
  if (p.M < 1024) return 0;
  if (p.conv) {
    if (p.Cin % BK2 != 0) return 0;
  } else {
    if (p.A2 != nullptr && (p.K1 % BK2) != 0) return 0;
  }
  int bn = 128;
  if (p.act != 1) {
    const int64_t pad128 = (int64_t)((p.N + 127) / 128) * 128, pad160 = (int64_t)((p.N + 159) / 160) * 160;
    if (pad160 < pad128) bn = 160;
  }
  const int64_t nb = p.batch > 1 ? p.batch : 1;
  const int64_t tiles256 = (int64_t)((p.M + 255) / 256) * ((p.N + bn - 1) / bn) * nb;
  if (tiles256 * 2 < 128) return 0;
  if (p.trans_out && (p.act == 1 || p.out_f32 || p.rowbias || p.residual)) return 0;
  const bool big = tiles256 >= 1024;   // >= 2 full rounds of 2 x 256 resident 256-row blocks
  if (big) {
    if (bn == 128) return p.conv ? launch_gemm2<256, 128, true>(p, stream) : launch_gemm2<256, 128, false>(p, stream);
    return p.conv ? launch_gemm2<256, 160, true>(p, stream) : launch_gemm2<256, 160, false>(p, stream);
  }
  if (bn == 128) return p.conv ? launch_gemm2<128, 128, true>(p, stream) : launch_gemm2<128, 128, false>(p, stream);
  return p.conv ? launch_gemm2<128, 160, true>(p, stream) : launch_gemm2<128, 160, false>(p, stream);
}
