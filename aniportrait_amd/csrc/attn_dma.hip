// Reference attention for gfx950, second generation (round 4): ref_attn_dma_kernel<D, NW>.
//
// Same mathematics and operand conventions as ref_attn_kernel (attention.hip): flash-style attention of 32-query groups
// against 64-key tiles of [self tokens ++ reference-bank tokens], S^T = K Q^T with v_mfma_f32_32x32x16_f16, online softmax
// with a lazily raised running maximum, O^T += V^T P^T.  What changed, each item decided by a measurement of this round
// (tools/exp_valu_rates.*, profiles/r04/j_attn_dma_component_ablation.txt):
//  * the kernel is CLOCK-limited on real data (the same launch runs 1.45 ms on random and 1.05 ms on all-zero operands), and
//    leaving single components out of the tile loop prices them (random data, of 1405 us): all MFMAs 533 us, the K / V^T
//    fragment reads LDS -> VGPR 517 us, the global -> LDS tile traffic 310 us, the 32 v_exp_f32 210 us, the tile maximum 63 us.
//    So the design moves as few bytes per query as it can:
//  * a workgroup is 8 waves x 32 queries = 256 queries of one (frame, head): a K / V^T tile is fetched once for all of them;
//  * K / V^T tiles travel global -> LDS by LDS-DMA (buffer_load ... lds, 1 KiB per wave instruction) into a 3-stage ring,
//    two tiles ahead of the compute, under COUNTED vmcnt and ONE s_barrier per tile: no staging registers, no ds_write
//    pass, no address arithmetic in the loop.  The DMA destination is lane-linear, so bank-conflict freedom is arranged
//    on the SOURCE side: K rows keep their natural 2 D-byte pitch when D/8 is odd (D = 40: 5 chunks) and get one pad
//    chunk otherwise; V^T rows (128 B = 8 chunks) are stored with chunk ^= (row >> 1) & {7, 5};
//  * V^T needs no key permutation any more: the K rows of a tile are READ in the order (bits 2 <-> 3 of the key index
//    swapped) that makes the accumulator registers of S^T line up with natural 8-key chunks of V^T;
//  * Q arrives PRE-MULTIPLIED by scale * log2(e) (the alpha of the to_q projection GEMM: one rounding, as before), and
//    when D % 16 == 8 the running maximum rides in two spare contraction slots (-m as an fp16 hi/lo pair on the Q side
//    against a constant [1, 1, 0...] chunk on the K side): the score MFMA delivers s - m directly and the softmax is
//    v_exp_f32 + v_cvt_pkrtz only;
//  * at d = 40 the P V product runs on 16x16x32 tiles (48 instead of 64 padded rows of O^T: 12 four-pass MFMAs instead of 8
//    eight-pass ones, 6 instead of 8 V^T fragment reads per tile), the P fragments re-dealt by 8 v_permlane16_swap.
// (O^T in AGPRs measured 388 -> 334 cycles per unit in the instruction-mix benchmark, but the compiler splits a 128-register
//  budget evenly between the two files as soon as a kernel may need AGPRs and spills; not pursued in HIP.)
// Shapes: T % 256 == 0, D in {40, 80, 160}, 16-B aligned rows; everything else stays on ref_attn_kernel.
#include <stdlib.h>

#include "attn_args.h"

namespace {

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
// (a device function: written inside the kernel's lambda, the builtin makes the HOST pass drop the kernel stub silently)
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
  // O^T += V^T P^T on 16x16x32 tiles when that needs fewer padded rows (d = 40: 48 instead of 64)
  static constexpr bool PV16 = ((D + (ONES ? 1 : 0) + 15) / 16) * 16 < DO * 32;
  static constexpr int DT = (D + (ONES ? 1 : 0) + 15) / 16;  // 16-row tiles of O^T (PV16)
  static constexpr int VROWS = PV16 ? DT * 16 : DO * 32;
  static constexpr int VSWZ = PV16 ? 5 : 7;           // V^T chunk swizzle: chunk ^= (row >> 1) & VSWZ (conflict-free b128 reads of the tile shape used)
  static constexpr int VP = D / 8;                    // DMA pieces per V^T tile: D rows x 8 chunks
  static constexpr int V_BYTES = VROWS * 128;
  static constexpr int NP = KP + VP;
  static constexpr int AUG = K_BYTES + V_BYTES;       // per-stage constant chunk [1, 1, 0, 0, 0, 0, 0, 0] (FOLD)
  static constexpr int STAGE = AUG + 64;
  static constexpr int LDS = NS * STAGE;
};

#define ANIP_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

}  // namespace

template <int D, int NW>
__global__ __launch_bounds__(NW * 64, (D <= 40 ? 4 : 2)) void ref_attn_dma_kernel(const RefAttnArgs a) {
  using G = Geo<D>;
  constexpr int NT2 = NW * 64;
  constexpr int NPW = (G::NP + NW - 1) / NW;   // DMA pieces per wave and tile (waves >= NP % NW issue one fewer when NP % NW != 0)
  constexpr int DQ = G::DQ, DO = G::DO, KSTR = G::KSTR, DC = G::DC, DT = G::DT;
  constexpr bool FOLD = G::FOLD, ONES = G::ONES, PV16 = G::PV16;
  // 32-query groups per wave.  2 (every LDS fragment feeds two groups, a tile is fetched for 512 queries, but 168 VGPRs = half
  // the waves per SIMD) measured the same speed as 1 at d = 40 (748 vs 750 TFLOP/s): the loops stay written over g
  constexpr int QH = 1;
  constexpr int LT = D / 32, LR = D % 32;   // O^T tile / row of the ones-row (32-row tiling)
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
  const int q0 = qb * (32 * NW) + wave * 32;  // first query of this wave
  const int ref = a.ref_index ? a.ref_index[n] : -1;
  const int ns = a.frame_mod > 0 ? n % a.frame_mod : n;   // frame whose q / k / v^T this frame attends with (its own output row stays n)
  const int nts = T / KV;
  const int ntiles = nts * (ref >= 0 ? 2 : 1);

  // ---- one-time LDS constants: ones / zero rows of V^T (rows D .. VROWS-1, never written by the DMA), the [1,1,0..] chunk ----
  for (int st = 0; st < NS; ++st) {
    char* vb_ = smem + st * G::STAGE + G::K_BYTES + D * 128;
    constexpr int PADW = (G::VROWS - D) * 128 / 4;     // dwords
    for (int i = tid; i < PADW; i += NT2) ((uint32_t*)vb_)[i] = (ONES && i < 32) ? 0x3C003C00u : 0u;
    if (tid < 4) ((uint32_t*)(smem + st * G::STAGE + G::AUG))[tid] = (tid == 0) ? 0x3C003C00u : 0u;
  }

  // ---- Q fragments (B operand of S^T = K Q^T): lane (ql, hi) holds Q[q][16 kk + 8 hi .. +7] -----------------------------
  f16x8 qf[QH][DQ];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    const f16* qp = a.q + ((int64_t)ns * T + q0 + 32 * g + ql) * a.ldq + h * D;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const int d0 = kk * 16 + hi * 8;
      U4H8 t;
      t.u = u32x4{0u, 0u, 0u, 0u};
      if (d0 < D) t.u = *(const u32x4*)(qp + d0);
      qf[g][kk] = t.h;
    }
  }

  // ---- DMA sources: piece p = wave + 8 i of a tile ([0, KP): K, [KP, NP): V^T; LDS destination stage + 1024 p either way) ----
  const f16* kb_s = a.k + (int64_t)ns * T * a.ldk + (int64_t)h * a.k_hs;
  const f16* kb_r = ref >= 0 ? a.kref + (int64_t)ref * T * a.ldkr + (int64_t)h * a.kr_hs : kb_s;
  const f16* vb_s = a.vt + (int64_t)h * D * a.ldvt + (int64_t)ns * T;
  const f16* vb_r = ref >= 0 ? a.vtref + (int64_t)h * D * a.ldvtr + (int64_t)ref * T : vb_s;
  const char* base_s[NPW];              // wave-uniform: operand base of the piece (self / reference segment)
  const char* base_r[NPW];
  uint32_t step_s[NPW], step_r[NPW]; // wave-uniform: bytes from one 64-key tile to the next
  uint32_t off_s[NPW], off_r[NPW];   // per lane: byte offset of this lane's 16-B chunk inside a tile
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int p = wave + NW * i;
    const bool isk = p < G::KP;
    base_s[i] = (const char*)(isk ? kb_s : vb_s);
    base_r[i] = (const char*)(isk ? kb_r : vb_r);
    step_s[i] = isk ? (uint32_t)(KV * a.ldk * 2) : 128u;
    step_r[i] = isk ? (uint32_t)(KV * a.ldkr * 2) : 128u;
    // K: LDS slot s <-> (key s / KSTR, chunk s % KSTR), pad slots fetch any valid address (they are never read)
    const int sk = 64 * p + lane;
    int key = sk / KSTR, c = sk - key * KSTR;
    if (c >= DC || key >= KV) { key = 0; c = 0; }
    // V^T: LDS slot s <-> (row s >> 3, chunk (s & 7) ^ ((row >> 1) & VSWZ))
    const int sv = 64 * (p - G::KP) + lane;
    const int r0 = sv >> 3, r = r0 < 0 ? 0 : (r0 > D - 1 ? D - 1 : r0), ch = (sv & 7) ^ ((r >> 1) & G::VSWZ);
    off_s[i] = isk ? (uint32_t)((key * (int)a.ldk + c * 8) * 2) : (uint32_t)((r * (int)a.ldvt + ch * 8) * 2);
    off_r[i] = isk ? (uint32_t)((key * (int)a.ldkr + c * 8) * 2) : (uint32_t)((r * (int)a.ldvtr + ch * 8) * 2);
  }
  auto issue_tile = [&](int t) {
    const bool second = t >= nts;
    const uint32_t tt = (uint32_t)(second ? t - nts : t);
    char* sb = smem + (t % NS) * G::STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      if (wave + NW * i < G::NP) {
        dma16(second ? base_r[i] : base_s[i], sb + i * (NW * 1024), second ? off_r[i] : off_s[i], tt * (second ? step_r[i] : step_s[i]));
      }
    }
  };
  const bool full_cnt = (G::NP % NW == 0) || (wave < G::NP % NW);   // this wave issues NPW pieces per tile (else NPW - 1)

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
  // PV16: A operand of the 16x16x32 MFMA: lane (m = lane & 15, g = lane >> 4) holds V^T[16 dt + m][8 keys of k-slot group g];
  // after the v_permlane16_swap of the P fragments group g carries chunk {0, 2, 1, 3}[g] of the 32-key half kc
  const int m16 = lane & 15, g16 = lane >> 4;
  int voff16[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
    voff16[kc] = G::K_BYTES + m16 * 128 + (((4 * kc + ((g16 & 1) * 2 + (g16 >> 1))) ^ ((m16 >> 1) & 5)) << 4);

  f32x16 o[QH][PV16 ? 1 : DO];
  f32x4 o16[QH][PV16 ? DT : 1][2];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
#pragma unroll
    for (int dt = 0; dt < (PV16 ? 1 : DO); ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][dt][r] = 0.f;
#pragma unroll
    for (int dt = 0; dt < (PV16 ? DT : 1); ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o16[g][dt][0][r] = o16[g][dt][1][r] = 0.f;
  }
  float m_run[QH];   // FOLD: the value encoded in qf[g][DQ-1] of the hi = 1 lanes (0 until the first tile is seen)
  float l_run[QH];   // !ONES: denominator by VALU adds
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    m_run[g] = FOLD ? 0.f : -INFINITY;
    l_run[g] = 0.f;
  }

#if defined(__HIP_DEVICE_COMPILE__)   // (register constraints are meaningless to the host pass, which then drops the kernel stub)
#pragma unroll
  for (int g = 0; g < QH; ++g)
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) asm volatile("" ::"v"(qf[g][kk]));   // Q has arrived: no compiler-placed vmcnt(0) behind the first DMA
#endif
  // ---- K / V^T fragments of a tile (read next to their MFMAs: fetching them one phase ahead — K of tile t + 1 under P V of tile t,
  // V^T under the softmax, the barrier between softmax and P V — measured 5-20 % SLOWER: the registers cost more than the latency)
  constexpr int NVF = PV16 ? 2 * DT : 4 * DO;
  f16x8 kf0[DQ], kf1[DQ], vf[NVF];
  auto load_k = [&](const char* st) {
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const bool last = kk == DQ - 1;
      kf0[kk] = *(const f16x8*)(st + (last ? klast0 : koff + kk * 32));
      kf1[kk] = *(const f16x8*)(st + (last ? klast1 : koff + kk * 32 + KHALF));
    }
  };
  auto load_v = [&](const char* st) {
    if constexpr (PV16) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vf[kc * DT + dt] = *(const f16x8*)(st + voff16[kc] + dt * 2048);
    } else {
#pragma unroll
      for (int gk = 0; gk < 4; ++gk)
#pragma unroll
        for (int dt = 0; dt < DO; ++dt) vf[gk * DO + dt] = *(const f16x8*)(st + voff[gk] + dt * 4096);
    }
  };
  // counted wait for this wave's pieces of one tile, the pieces of the NEXT tile (if any were issued) staying in flight
  auto wait_tile = [&](bool next_in_flight) {
    if (!next_in_flight) ANIP_VMCNT(0);
    else if (full_cnt) ANIP_VMCNT(NPW);
    else ANIP_VMCNT(NPW - 1);
  };

  __syncthreads();                        // constants visible (and no DMA in flight yet: the fence drains nothing)
  issue_tile(0);
  if (ntiles > 1) issue_tile(1);

  for (int t = 0; t < ntiles; ++t) {
    const char* st = smem + (t % NS) * G::STAGE;
    // my pieces of tile t have landed: everything but the pieces of tile t + 1 (issued later) is complete
    wait_tile(t + 1 < ntiles);
    __builtin_amdgcn_s_barrier();         // tile t complete for all waves; stage (t + 2) % NS (read at t - 1) is free
    asm volatile("" ::: "memory");
    if (t + 2 < ntiles) issue_tile(t + 2);
    load_k(st);

    // ---- S^T = K Q^T (- m): two 32-key x 32-query tiles per query group; every K fragment feeds all groups ------------------
    f32x16 s0[QH], s1[QH];
#pragma unroll
  for (int g = 0; g < QH; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[g][r] = s1[g][r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const f16x8 a0 = kf0[kk], a1 = kf1[kk];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
        s0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[g][kk], s0[g], 0, 0, 0);
        s1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[g][kk], s1[g], 0, 0, 0);
      }
    }
    U4H8 pb[QH][4];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
      // ---- tile maximum (one query per lane; lane ^ 32 holds the other 32 keys) -------------------------------------------
      float mx = fmaxf(fmaxf(s0[g][0], s0[g][1]), s0[g][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s0[g][r]), s0[g][r + 1]);
      mx = fmaxf(fmaxf(mx, s0[g][15]), s1[g][0]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s1[g][r]), s1[g][r + 1]);
      mx = fmaxf(mx, s1[g][15]);
      mx = xor32_max(mx);
      // O^T *= alpha (alpha per query, held in the 32x32 layout: lane <-> query lane & 31)
      auto scale_o = [&](float alpha) {
        if constexpr (PV16) {
          const auto al = __builtin_amdgcn_permlane16_swap(__float_as_uint(alpha), __float_as_uint(alpha), false, false);
          const float a0 = __uint_as_float(al[0]), a1 = __uint_as_float(al[1]);   // queries (lane & 15) / 16 + (lane & 15)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o16[g][dt][0][r] *= a0;
              o16[g][dt][1][r] *= a1;
            }
        } else {
#pragma unroll
          for (int dt = 0; dt < DO; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[g][dt][r] *= alpha;
        }
      };
      if (FOLD) {
        // scores are relative to m_run already
        if (t == 0 || __any(mx > THR)) {
          const float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
          const float m_new = m_run[g] + delta;
          const f16 mh = (f16)m_new;
          const f16 ml = (f16)(m_new - (float)mh);
          const float m_enc = (float)mh + (float)ml;       // what the MFMA will subtract from now on
          const float d_eff = m_enc - m_run[g];
          m_run[g] = m_enc;
          if (t != 0) scale_o(__builtin_amdgcn_exp2f(-d_eff));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            s0[g][r] -= d_eff;
            s1[g][r] -= d_eff;
          }
          if (hi) {
            f16x8 aug = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
            aug[0] = -mh;
            aug[1] = -ml;
            qf[g][DQ - 1] = aug;
          }
        }
      } else {
        if (__any(mx - m_run[g] > THR)) {     // always taken on the first tile (m_run = -inf)
          const float m_new = fmaxf(m_run[g], mx);
          const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
          m_run[g] = m_new;
          l_run[g] *= alpha;
          scale_o(alpha);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[g][r] -= m_run[g];
          s1[g][r] -= m_run[g];
        }
      }
      // ---- P^T fragments (B operand): slot (hi, j) of 16-key group gk <-> accumulator register 8 (gk & 1) + j of tile gk >> 1 ----
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float p00 = __builtin_amdgcn_exp2f(s0[g][j]), p01 = __builtin_amdgcn_exp2f(s0[g][j + 1]);
        const float p10 = __builtin_amdgcn_exp2f(s0[g][8 + j]), p11 = __builtin_amdgcn_exp2f(s0[g][9 + j]);
        const float p20 = __builtin_amdgcn_exp2f(s1[g][j]), p21 = __builtin_amdgcn_exp2f(s1[g][j + 1]);
        const float p30 = __builtin_amdgcn_exp2f(s1[g][8 + j]), p31 = __builtin_amdgcn_exp2f(s1[g][9 + j]);
        if (!ONES) l_run[g] += ((p00 + p01) + (p10 + p11)) + ((p20 + p21) + (p30 + p31));
        pb[g][0].u[j >> 1] = pk_f16(p00, p01);
        pb[g][1].u[j >> 1] = pk_f16(p10, p11);
        pb[g][2].u[j >> 1] = pk_f16(p20, p21);
        pb[g][3].u[j >> 1] = pk_f16(p30, p31);
      }
    }
    load_v(st);
    // ---- O^T += V^T P^T; every V^T fragment feeds all query groups; consecutive MFMAs on different accumulators ----------------
    if constexpr (PV16) {
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        // B operands of the two 16-query tiles of a group: lanes 16-31 / 48-63 of fragment 2 kc trade places with lanes
        // 0-15 / 32-47 of fragment 2 kc + 1
        U4H8 b0[QH], b1[QH];
#pragma unroll
  for (int g = 0; g < QH; ++g)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const auto sw = __builtin_amdgcn_permlane16_swap(pb[g][2 * kc].u[w], pb[g][2 * kc + 1].u[w], false, false);
            b0[g].u[w] = sw[0];
            b1[g].u[w] = sw[1];
          }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const f16x8 av = vf[kc * DT + dt];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
            o16[g][dt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b0[g].h, o16[g][dt][0], 0, 0, 0);
            o16[g][dt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, b1[g].h, o16[g][dt][1], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll
      for (int gk = 0; gk < 4; ++gk) {
#pragma unroll
        for (int dt = 0; dt < DO; ++dt) {
          const f16x8 av = vf[gk * DO + dt];
#pragma unroll
          for (int g = 0; g < QH; ++g) o[g][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb[g][gk].h, o[g][dt], 0, 0, 0);
        }
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------------------------------------
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    if constexpr (PV16) {
      // lane (n = lane & 15, g16 = lane >> 4) holds O^T[16 dt + 4 g16 + r][query 16 qt + n]; the denominator is row D
      static_assert(!PV16 || ONES, "the 16-row tiling is only chosen when the ones-row supplies the denominator");
      constexpr int LDT = D / 16, LG = (D % 16) >> 2, LRR = D & 3;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        const float l_tot = __shfl(o16[g][LDT][qt][LRR], LG * 16 + m16, 64);
        const float inv = 1.0f / l_tot;
        f16* op = a.out + ((int64_t)n * T + q0 + 32 * g + qt * 16 + m16) * a.ldo + h * D;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int d0 = dt * 16 + g16 * 4;
          if (d0 < D) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(o16[g][dt][qt][e] * inv);
            *(f16x4*)(op + d0) = v;
          }
        }
      }
    } else {
      float l_tot;
      if (ONES) {
        const float lv = o[g][ONES ? LT : 0][L_REG];                  // row D of O^T: held by the half-wave with hi == L_HI
        const float lp = __shfl_xor(lv, 32, 64);
        l_tot = (hi == L_HI) ? lv : lp;
      } else {
        l_tot = l_run[g] + __shfl_xor(l_run[g], 32, 64);
      }
      const float inv = 1.0f / l_tot;
      f16* op = a.out + ((int64_t)n * T + q0 + 32 * g + ql) * a.ldo + h * D;
#pragma unroll
      for (int dt = 0; dt < DO; ++dt)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int d0 = dt * 32 + rq * 8 + hi * 4;
          if (d0 < D) {
            f16x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (f16)(o[g][dt][rq * 4 + e] * inv);
            *(f16x4*)(op + d0) = v;
          }
        }
    }
  }
}

namespace {

template <int D, int NW>
int launch_dma(const RefAttnArgs& a, int Nf, hipStream_t stream) {
  using G = Geo<D>;
  if (anip_raise_lds_limit((const void*)ref_attn_dma_kernel<D, NW>, G::LDS) != 0) {
    anip_set_error("anip_ref_attention: cannot raise the dynamic LDS limit to %d bytes", G::LDS);
    return -2;
  }
  dim3 grid((unsigned)(a.T / (32 * NW)), (unsigned)a.heads, (unsigned)Nf);
  AnipProfScope prof_(ANIP_K_REF_ATTN, (void*)stream);
  hipLaunchKernelGGL((ref_attn_dma_kernel<D, NW>), grid, dim3(NW * 64), G::LDS, stream, a);
  return 1;
}

}  // namespace

int anip_ref_attn_dma_try(const RefAttnArgs& a, int Nf, int d, hipStream_t stream) {
  if (a.T % 256 != 0 || !a.vt_vec_ok || (a.ref_index != nullptr && !a.vtref_vec_ok)) return 0;
  const bool fits32 = (int64_t)a.T * a.ldk * 2 < (1ll << 31) && (int64_t)a.T * a.ldkr * 2 < (1ll << 31) &&
                      (int64_t)(d + 1) * a.ldvt * 2 < (1ll << 31) && (int64_t)(d + 1) * a.ldvtr * 2 < (1ll << 31);
  if (!fits32) return 0;
  switch (d) {
    // 8 waves per workgroup; 4 (four independent workgroups per CU, twice the tile traffic) measured the same at d = 40
    // and 10 % slower at d = 80
    case 40: return launch_dma<40, 8>(a, Nf, stream);
    case 80: return launch_dma<80, 8>(a, Nf, stream);
    case 160: return launch_dma<160, 8>(a, Nf, stream);
    default: return 0;
  }
}
