// Reference attention for gfx950, second generation (round 4): ref_attn_dma_kernel<D>.
//
// Same mathematics and operand conventions as ref_attn_kernel (attention.hip): flash-style attention of 32 queries per
// wave against 64-key tiles of [self tokens ++ reference-bank tokens], S^T = K Q^T and O^T += V^T P^T with
// v_mfma_f32_32x32x16_f16, online softmax with a lazily raised running maximum.  What changed, each item decided by a
// measurement of this round (tools/exp_valu_rates.*, profiles/r04/):
//  * a workgroup is 8 waves = 256 queries of one (frame, head): a K / V^T tile is fetched once for twice the queries;
//  * K / V^T tiles travel global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) into a 3-stage ring,
//    two tiles ahead of the compute, under COUNTED vmcnt and ONE s_barrier per tile: no staging registers, no ds_write
//    pass, no address arithmetic in the loop.  The DMA destination is lane-linear, so bank-conflict freedom is arranged
//    on the SOURCE side: K rows keep their natural 2 D-byte pitch when D/8 is odd (D = 40: 5 chunks) and get one pad
//    chunk otherwise; V^T rows (128 B = 8 chunks) are stored with chunk ^= (row >> 1) & 7;
//  * V^T needs no key permutation any more: the K rows of a tile are READ in the order (bits 2 <-> 3 of the key index
//    swapped) that makes the accumulator registers of S^T line up with natural 8-key chunks of V^T;
//  * the O^T accumulators live in AGPRs (inline-asm MFMA with "a" operands): the unit mix 14 MFMA | 72 VALU measured
//    388 -> 334 cycles per SIMD with the P V accumulators out of the VGPR file (VALU and MFMA no longer compete for
//    its ports);
//  * Q arrives PRE-MULTIPLIED by scale * log2(e) (the alpha of the to_q projection GEMM: one rounding, as before), and
//    when D % 16 == 8 the running maximum rides in two spare contraction slots (-m as an fp16 hi/lo pair on the Q side
//    against a constant [1, 1, 0...] chunk on the K side): the score MFMA delivers s - m directly and the softmax is
//    v_exp_f32 + v_cvt_pkrtz only — 32 v_fma per tile fewer on the VALU, which together with the transcendental unit
//    bounds this kernel.
// Shapes: T % 256 == 0, D in {40, 80, 160}, 16-B aligned rows; everything else stays on ref_attn_kernel.
#include <stdlib.h>

#include "attn_args.h"

namespace {

constexpr int NT2 = 512;     // 8 waves
constexpr int KV = 64;       // keys per tile
constexpr int NS = 3;        // LDS ring stages
constexpr float THR = 8.0f;  // lazy rescale: the running max is raised when a tile exceeds it by more than 2^THR

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

typedef __fp16 fp16x2_t __attribute__((ext_vector_type(2)));
union H2U {
  fp16x2_t h;
  unsigned int u;
};
__device__ __forceinline__ unsigned int pk_f16(float a, float b) {
  H2U t;
  t.h = __builtin_amdgcn_cvt_pkrtz(a, b);
  return t.u;
}
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// one LDS-DMA instruction: 64 lanes x 16 B from base + soff + voff[lane] to lds + 16 lane
__device__ __forceinline__ void dma16(const void* base, char* lds, uint32_t voff, uint32_t soff) {
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7FFFFFF0, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(lds), 16, voff, soff, 0, 0);
}

template <int D>
struct Geo {
  static constexpr int DQ = (D + 15) / 16;            // 16-wide contraction steps of S^T = K Q^T
  static constexpr bool FOLD = (D % 16) == 8;         // spare contraction slots: the running max rides in the MFMA
  static constexpr bool ONES = (D % 32) != 0;         // spare O^T row: the softmax denominator falls out of P V
  static constexpr int DC = D / 8;                    // 16-B chunks per K row
  static constexpr int KSTR = DC | 1;                 // chunks per K row in LDS (odd: conflict-free b128 reads)
  static constexpr int KP = (KV * KSTR + 63) / 64;    // DMA pieces (1 KiB) per K tile
  static constexpr int K_BYTES = KP * 1024;
  static constexpr int DO = (D + (ONES ? 1 : 0) + 31) / 32;  // 32-row tiles of O^T
  static constexpr int VP = D / 8;                    // DMA pieces per V^T tile: D rows x 8 chunks
  static constexpr int V_BYTES = DO * 32 * 128;
  static constexpr int NP = KP + VP;
  static constexpr int NPW = (NP + 7) / 8;            // pieces per wave (waves 0 .. NP % 8 - 1 issue NPW, the others NPW - 1 when NP % 8 != 0)
  static constexpr int AUG = K_BYTES + V_BYTES;       // per-stage constant chunk [1, 1, 0, 0, 0, 0, 0, 0] (FOLD)
  static constexpr int STAGE = AUG + 64;
  static constexpr int LDS = NS * STAGE;
  static constexpr int LDL = (int)((DO * 16 + 15) / 16);
};

#define ANIP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

}  // namespace

template <int D>
__global__ __launch_bounds__(NT2, (D <= 40 ? 4 : 2)) void ref_attn_dma_kernel(const RefAttnArgs a) {
  using G = Geo<D>;
  constexpr int DQ = G::DQ, DO = G::DO, KSTR = G::KSTR, DC = G::DC;
  constexpr bool FOLD = G::FOLD, ONES = G::ONES;
  constexpr int LT = D / 32, LR = D % 32;   // O^T tile / row of the ones-row
  constexpr int L_HI = (LR >> 2) & 1, L_REG = 4 * (LR >> 3) + (LR & 3);
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, hi = lane >> 5;
  const int T = a.T;
  // block -> (query block, head, frame); all query blocks of a (frame, head) on one XCD (see attention.hip)
  int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  {
    const int nqb = gridDim.x, nfh = gridDim.y * gridDim.z;
    if ((nfh & 7) == 0) {
      const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = lin & 7, kq = lin >> 3;
      const int fh = (kq / nqb) * 8 + xcd;
      qb = kq % nqb;
      h = fh % gridDim.y;
      n = fh / gridDim.y;
    }
  }
  const int q = qb * 256 + wave * 32 + ql;
  const int ref = a.ref_index ? a.ref_index[n] : -1;
  const int nts = T / KV;
  const int ntiles = nts * (ref >= 0 ? 2 : 1);

  // ---- one-time LDS constants: ones / zero rows of V^T (rows D .. DO*32-1, never written by the DMA), the [1,1,0..] chunk ----
  for (int st = 0; st < NS; ++st) {
    char* vb_ = smem + st * G::STAGE + G::K_BYTES + D * 128;
    constexpr int PADW = (DO * 32 - D) * 128 / 4;      // dwords
    for (int i = tid; i < PADW; i += NT2) ((uint32_t*)vb_)[i] = (ONES && i < 32) ? 0x3C003C00u : 0u;
    if (tid < 4) ((uint32_t*)(smem + st * G::STAGE + G::AUG))[tid] = (tid == 0) ? 0x3C003C00u : 0u;
  }

  // ---- Q fragments (B operand of S^T = K Q^T): lane (ql, hi) holds Q[q][16 kk + 8 hi .. +7] -----------------------------
  f16x8 qf[DQ];
  {
    const f16* qp = a.q + ((int64_t)n * T + q) * a.ldq + h * D;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const int d0 = kk * 16 + hi * 8;
      U4H8 t;
      t.u = u32x4{0u, 0u, 0u, 0u};
      if (d0 < D) t.u = *(const u32x4*)(qp + d0);
      qf[kk] = t.h;
    }
  }

  // ---- DMA sources: piece p = wave + 8 i of a tile ([0, KP): K, [KP, NP): V^T; LDS destination stage + 1024 p either way) ----
  const f16* kb_s = a.k + (int64_t)n * T * a.ldk + (int64_t)h * a.k_hs;
  const f16* kb_r = ref >= 0 ? a.kref + (int64_t)ref * T * a.ldkr + (int64_t)h * a.kr_hs : kb_s;
  const f16* vb_s = a.vt + (int64_t)h * D * a.ldvt + (int64_t)n * T;
  const f16* vb_r = ref >= 0 ? a.vtref + (int64_t)h * D * a.ldvtr + (int64_t)ref * T : vb_s;
  const char* base_s[G::NPW];              // wave-uniform: operand base of the piece (self / reference segment)
  const char* base_r[G::NPW];
  uint32_t step_s[G::NPW], step_r[G::NPW]; // wave-uniform: bytes from one 64-key tile to the next
  uint32_t off_s[G::NPW], off_r[G::NPW];   // per lane: byte offset of this lane's 16-B chunk inside a tile
#pragma unroll
  for (int i = 0; i < G::NPW; ++i) {
    const int p = wave + 8 * i;
    const bool isk = p < G::KP;
    base_s[i] = (const char*)(isk ? kb_s : vb_s);
    base_r[i] = (const char*)(isk ? kb_r : vb_r);
    step_s[i] = isk ? (uint32_t)(KV * a.ldk * 2) : 128u;
    step_r[i] = isk ? (uint32_t)(KV * a.ldkr * 2) : 128u;
    // K: LDS slot s <-> (key s / KSTR, chunk s % KSTR), pad slots fetch any valid address (they are never read)
    const int sk = 64 * p + lane;
    int key = sk / KSTR, c = sk - key * KSTR;
    if (c >= DC || key >= KV) { key = 0; c = 0; }
    // V^T: LDS slot s <-> (row s >> 3, chunk (s & 7) ^ ((row >> 1) & 7))
    const int sv = 64 * (p - G::KP) + lane;
    const int r0 = sv >> 3, r = r0 < 0 ? 0 : (r0 > D - 1 ? D - 1 : r0), ch = (sv & 7) ^ ((r >> 1) & 7);
    off_s[i] = isk ? (uint32_t)((key * (int)a.ldk + c * 8) * 2) : (uint32_t)((r * (int)a.ldvt + ch * 8) * 2);
    off_r[i] = isk ? (uint32_t)((key * (int)a.ldkr + c * 8) * 2) : (uint32_t)((r * (int)a.ldvtr + ch * 8) * 2);
  }
  auto issue_tile = [&](int t) {
    const bool second = t >= nts;
    const uint32_t tt = (uint32_t)(second ? t - nts : t);
    char* sb = smem + (t % NS) * G::STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < G::NPW; ++i) {
      if (wave + 8 * i < G::NP) {
        dma16(second ? base_r[i] : base_s[i], sb + i * 8192, second ? off_r[i] : off_s[i], tt * (second ? step_r[i] : step_s[i]));
      }
    }
  };
  const bool full_cnt = (G::NP % 8 == 0) || (wave < G::NP % 8);   // this wave issues NPW pieces per tile (else NPW - 1)

  // ---- per-lane LDS read offsets (bytes, relative to the stage) ----------------------------------------------------------
  const int krow = (ql & 0x13) | ((ql & 4) << 1) | ((ql & 8) >> 1);       // key read by A-operand row ql: bits 2 <-> 3
  const int koff = krow * KSTR * 16 + hi * 16;                            // + kk * 32 (+ 32 KSTR 16 for the second 32 keys)
  constexpr int KHALF = 32 * KSTR * 16;
  // last contraction step with the folded maximum: the hi = 1 half-wave reads the constant chunk instead of K
  const int klast0 = (FOLD && hi) ? G::AUG : koff + (DQ - 1) * 32;
  const int klast1 = (FOLD && hi) ? G::AUG : koff + (DQ - 1) * 32 + KHALF;
  int voff[4];
#pragma unroll
  for (int gk = 0; gk < 4; ++gk) voff[gk] = G::K_BYTES + ql * 128 + (((2 * gk + hi) ^ ((ql >> 1) & 7)) << 4);

  f32x16 o[DO];
#pragma unroll
  for (int dt = 0; dt < DO; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = FOLD ? 0.f : -INFINITY;   // FOLD: the value encoded in qf[DQ-1] of the hi = 1 lanes (0 until the first tile is seen)
  float l_run = 0.f;                      // !ONES: denominator by VALU adds

#if defined(__HIP_DEVICE_COMPILE__)   // (register constraints are meaningless to the host pass, which then drops the kernel stub)
#pragma unroll
  for (int kk = 0; kk < DQ; ++kk) asm volatile("" ::"v"(qf[kk]));   // Q has arrived: no compiler-placed vmcnt(0) behind the first DMA
#endif
  __syncthreads();                        // constants visible (and no DMA in flight yet: the fence drains nothing)
  issue_tile(0);
  if (ntiles > 1) issue_tile(1);

  for (int t = 0; t < ntiles; ++t) {
    // my pieces of tile t have landed: everything but the pieces of tile t + 1 (issued later) is complete
    if (t + 1 < ntiles) {
      if (full_cnt) ANIP_VMCNT(G::NPW);
      else ANIP_VMCNT(G::NPW - 1);
    } else {
      ANIP_VMCNT(0);
    }
    __builtin_amdgcn_s_barrier();         // tile t complete for all waves; stage (t + 2) % NS (read at t - 1) is free
    asm volatile("" ::: "memory");
    if (t + 2 < ntiles) issue_tile(t + 2);
    const char* st = smem + (t % NS) * G::STAGE;

    // ---- S^T = K Q^T (- m): two 32-key x 32-query tiles -------------------------------------------------------------------
    f32x16 s0, s1;
#if defined(ANIP_ATTN_AGPR) && defined(__HIP_DEVICE_COMPILE__)
    // The `a` operand below makes the compiler select the AGPR form for every BUILTIN MFMA of this kernel (the P V ones:
    // O^T then lives in the accumulator file, and all their hazards stay the compiler's business); the score MFMAs must
    // deliver into VGPRs (the VALU reads them), so they are written out by hand: operands complete (the compiler waits for
    // the LDS reads feeding an asm statement), alternating accumulators, zero as the first addend, and 12 wait states
    // behind the last one before the VALU may read its result (8 passes + margin; the block is opaque to the hazard recognizer).
    {
      f16x8 ka[DQ], kb[DQ];
#pragma unroll
      for (int kk = 0; kk < DQ; ++kk) {
        const bool last = kk == DQ - 1;
        ka[kk] = *(const f16x8*)(st + (last ? klast0 : koff + kk * 32));
        kb[kk] = *(const f16x8*)(st + (last ? klast1 : koff + kk * 32 + KHALF));
      }
      asm volatile("" ::"a"(0.0f));
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %4, 0\n v_mfma_f32_32x32x16_f16 %1, %3, %4, 0" : "=&v"(s0), "=&v"(s1) : "v"(ka[0]), "v"(kb[0]), "v"(qf[0]));
#pragma unroll
      for (int kk = 1; kk < DQ; ++kk)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %4, %0\n v_mfma_f32_32x32x16_f16 %1, %3, %4, %1" : "+v"(s0), "+v"(s1) : "v"(ka[kk]), "v"(kb[kk]), "v"(qf[kk]));
      asm volatile("s_nop 7\n s_nop 3" : "+v"(s0), "+v"(s1));
    }
#else
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const bool last = kk == DQ - 1;
      const f16x8 a0 = *(const f16x8*)(st + (last ? klast0 : koff + kk * 32));
      const f16x8 a1 = *(const f16x8*)(st + (last ? klast1 : koff + kk * 32 + KHALF));
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[kk], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[kk], s1, 0, 0, 0);
    }
#endif
    // ---- tile maximum (one query per lane; lane ^ 32 holds the other 32 keys) ---------------------------------------------
    float mx = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s0[r]), s0[r + 1]);
    mx = fmaxf(fmaxf(mx, s0[15]), s1[0]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s1[r]), s1[r + 1]);
    mx = fmaxf(mx, s1[15]);
    mx = xor32_max(mx);
    if (FOLD) {
      // scores are relative to m_run already
      if (t == 0 || __any(mx > THR)) {
        const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
        const float m_new = m_run + delta;
        const f16 mh = (f16)m_new;
        const f16 ml = (f16)(m_new - (float)mh);
        const float m_enc = (float)mh + (float)ml;       // what the MFMA will subtract from now on
        const float d_eff = m_enc - m_run;
        m_run = m_enc;
        if (t != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-d_eff);
#pragma unroll
          for (int dt = 0; dt < DO; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] -= d_eff;
          s1[r] -= d_eff;
        }
        if (hi) {
          f16x8 aug = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          aug[0] = -mh;
          aug[1] = -ml;
          qf[DQ - 1] = aug;
        }
      }
    } else {
      if (__any(mx - m_run > THR)) {     // always taken on the first tile (m_run = -inf)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int dt = 0; dt < DO; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s0[r] -= m_run;
        s1[r] -= m_run;
      }
    }
    // ---- P^T fragments (B operand): slot (hi, j) of 16-key group gk <-> accumulator register 8 (gk & 1) + j of tile gk >> 1 ----
    U4H8 pb[4];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      const float p00 = __builtin_amdgcn_exp2f(s0[j]), p01 = __builtin_amdgcn_exp2f(s0[j + 1]);
      const float p10 = __builtin_amdgcn_exp2f(s0[8 + j]), p11 = __builtin_amdgcn_exp2f(s0[9 + j]);
      const float p20 = __builtin_amdgcn_exp2f(s1[j]), p21 = __builtin_amdgcn_exp2f(s1[j + 1]);
      const float p30 = __builtin_amdgcn_exp2f(s1[8 + j]), p31 = __builtin_amdgcn_exp2f(s1[9 + j]);
      if (!ONES) l_run += ((p00 + p01) + (p10 + p11)) + ((p20 + p21) + (p30 + p31));
      pb[0].u[j >> 1] = pk_f16(p00, p01);
      pb[1].u[j >> 1] = pk_f16(p10, p11);
      pb[2].u[j >> 1] = pk_f16(p20, p21);
      pb[3].u[j >> 1] = pk_f16(p30, p31);
    }
    // ---- O^T += V^T P^T, accumulators in AGPRs; consecutive MFMAs on different accumulators -----------------------------------
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
#pragma unroll
      for (int dt = 0; dt < DO; ++dt) {
        const f16x8 av = *(const f16x8*)(st + voff[gk] + dt * 4096);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb[gk].h, o[dt], 0, 0, 0);
      }
    }
  }


  // ---- epilogue -------------------------------------------------------------------------------------------------------------
  float l_tot;
  if (ONES) {
    const float lv = o[ONES ? LT : 0][L_REG];                  // row D of O^T: held by the half-wave with hi == L_HI
    const float lp = __shfl_xor(lv, 32, 64);
    l_tot = (hi == L_HI) ? lv : lp;
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  f16* op = a.out + ((int64_t)n * T + q) * a.ldo + h * D;
#pragma unroll
  for (int dt = 0; dt < DO; ++dt)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const int d0 = dt * 32 + rq * 8 + hi * 4;
      if (d0 < D) {
        f16x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (f16)(o[dt][rq * 4 + e] * inv);
        *(f16x4*)(op + d0) = v;
      }
    }
}

namespace {

template <int D>
int launch_dma(const RefAttnArgs& a, int Nf, hipStream_t stream) {
  using G = Geo<D>;
  static bool attr_done[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_done[dev]) {
    if (hipFuncSetAttribute((const void*)ref_attn_dma_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS) != hipSuccess) {
      anip_set_error("anip_ref_attention: cannot raise the dynamic LDS limit to %d bytes", G::LDS);
      return -2;
    }
    attr_done[dev] = true;
  }
  dim3 grid((unsigned)(a.T / 256), (unsigned)a.heads, (unsigned)Nf);
  AnipProfScope prof_(ANIP_K_REF_ATTN, (void*)stream);
  hipLaunchKernelGGL((ref_attn_dma_kernel<D>), grid, dim3(NT2), G::LDS, stream, a);
  return 1;
}

}  // namespace

int anip_ref_attn_dma_try(const RefAttnArgs& a, int Nf, int d, hipStream_t stream) {
  static const int off = getenv("ANIP_ATTN_DMA") ? (atoi(getenv("ANIP_ATTN_DMA")) == 0) : 0;   // ANIP_ATTN_DMA=0: A/B against ref_attn_kernel
  if (off) return 0;
  if (a.T % 256 != 0 || !a.vt_vec_ok || (a.ref_index != nullptr && !a.vtref_vec_ok)) return 0;
  const bool fits32 = (int64_t)a.T * a.ldk * 2 < (1ll << 31) && (int64_t)a.T * a.ldkr * 2 < (1ll << 31) &&
                      (int64_t)(d + 1) * a.ldvt * 2 < (1ll << 31) && (int64_t)(d + 1) * a.ldvtr * 2 < (1ll << 31);
  if (!fits32) return 0;
  switch (d) {
    case 40: return launch_dma<40>(a, Nf, stream);
    case 80: return launch_dma<80>(a, Nf, stream);
    case 160: return launch_dma<160>(a, Nf, stream);
    default: return 0;
  }
}
