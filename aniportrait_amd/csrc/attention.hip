// Attention kernels for gfx950 (CDNA4).
//
// 1. ref_attn_kernel<D>: flash-style spatial attention with an optional second key/value segment
//    (the ReferenceNet bank).  One workgroup = 4 waves x 32 queries of one (frame, head); K/V tiles
//    of 64 keys are staged global -> registers -> LDS (double buffered) and shared by the 4 waves.
//    The score tile is computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_f16, so every lane
//    owns one query column: row max / row sum are in-register reductions plus one lane^32 exchange,
//    the probabilities are already laid out as the B operand of the second MFMA (O^T += V^T P^T),
//    and the online-softmax rescale of O is one scalar per lane.  V is consumed pre-transposed
//    (V^T[d][token], produced by the projection GEMM with swapped operands), so its LDS image needs no
//    transpose — only a fixed permutation of the 16-key groups that matches the accumulator layout.
// 2. temporal_attn_kernel: attention over the F (<= 32) frames of one pixel & head; purely HBM-bound
//    strided gather, one wave per problem, everything staged in LDS.
#include <stdlib.h>

#include "attn_args.h"

namespace {

constexpr int NT = 256;
constexpr int KV = 64;       // keys per tile
constexpr int VROW = KV + 8; // fp16 elements per V^T LDS row (144 B = 9 x 16 B: odd -> conflict-free b128 reads)

// raw v_exp_f32 (no denormal fix-up sequence around it: inputs here are <= ~THR and results below 2^-126
// may flush to zero) and packed round-toward-zero fp32 -> fp16 conversion of the probabilities.  The
// truncation bias is common to the numerator (P V) and — with the ones-row trick below — the denominator.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
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

// max(x[lane], x[lane ^ 32]) with one v_permlane32_swap (lanes 32..63 of the first operand <-> lanes 0..31 of the
// second) instead of a ds_bpermute round trip through the LDS on the softmax's critical path
__device__ __forceinline__ float xor32_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Online softmax is the VALU-bound part of this kernel at d = 40 (rocprofv3: SQ_ACTIVE_INST_VALU 88 % of the
// kernel's cycles vs 23 % MFMA busy before this restructuring), so the per-score VALU work is cut to
// max3 / fma / v_exp / cvt_pk:
//  * the running max is only raised (and O rescaled) when some query's tile max exceeds it by more than
//    2^RESCALE_LOG2 (lazy rescale): probabilities stay <= 2^RESCALE_LOG2, exact in fp16/fp32 either way;
//  * when D is not a multiple of 32, row D of the zero-padded V^T tile is set to ones, so the softmax
//    denominator falls out of the P V MFMA (row D of O^T) instead of 32 VALU adds per tile.
constexpr float RESCALE_LOG2 = 8.0f;

// FAST: T % 64 == 0 and 16-B aligned V^T rows (every C2 shape): no key masking, branch-free 16-B loads with
// 32-bit per-thread offsets from a wave-uniform base.
// QH: 32-query groups per wave (1 or 2).  With QH = 2 a wave owns 64 queries: every K / V^T fragment read from LDS
// feeds two MFMAs, and the two groups are independent dependency chains issued in the order
//   QK(0) QK(1) | softmax(0) | PV(0) | softmax(1) | PV(1)
// so group 1's score MFMAs run in the matrix pipe under group 0's exponentials and group 0's P V MFMAs under group
// 1's: the in-order wave overlaps its own VALU and MFMA work instead of relying on a co-resident wave being in the
// other phase (measured on the QH = 1 kernel at d = 40: per-tile SIMD time = VALU time + MFMA time, i.e. no overlap).
template <int D, bool FAST, int QH>
__global__ __launch_bounds__(NT, (D > 96 ? 1 : 2)) void ref_attn_kernel(const RefAttnArgs a) {
  constexpr int DQ = (D + 15) / 16;        // 16-wide contraction chunks of Q K^T
  constexpr int DO = (D + 31) / 32;        // 32-row output tiles of O^T
  constexpr int KROW = DQ * 16 + 8;        // fp16 per K LDS row; (2*DQ+1) 16-B slots: odd
  constexpr int DC = D / 8;                // 16-B chunks per head row
  constexpr int NCH = (KV * DC + NT - 1) / NT;  // staged chunks per thread (K and V^T each)
  constexpr bool ONES = (D % 32) != 0;     // spare padded V^T row available for the denominator
  constexpr int LT = D / 32, LR = D % 32;  // O^T tile / row of the ones-row
  constexpr int L_HI = (LR >> 2) & 1, L_REG = 4 * (LR >> 3) + (LR & 3);
  constexpr int QPB = 128 * QH;            // queries per block
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  __shared__ __attribute__((aligned(16))) f16 sK[2][KV * KROW];
  __shared__ __attribute__((aligned(16))) f16 sV[2][DO * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int T = a.T;
  // Block -> (query block, head, frame).  Workgroups are dealt round-robin to the 8 XCDs in launch order (x fastest),
  // and every XCD has its own L2: with the natural order the query blocks of one (frame, head) — which all stream the
  // same K / V^T — land on all 8 L2s and each fetches that K / V^T from the fabric (rocprofv3 FETCH_SIZE: 2.1 GB per
  // launch at 64x64, d = 40, against 0.26 GB of operands).  When the (frame, head) count divides by 8, the launch order
  // is re-read so that XCD x owns (frame, head) pairs x, x + 8, ... with all their query blocks.
  int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
  {
    const int nqb = gridDim.x, nfh = gridDim.y * gridDim.z;
    if ((nfh & 7) == 0) {
      const int lin = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
      const int xcd = lin & 7, k = lin >> 3;
      const int fh = (k / nqb) * 8 + xcd;
      qb = k % nqb;
      h = fh % gridDim.y;
      n = fh / gridDim.y;
    }
  }
  int q[QH];
  bool qvalid[QH];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    q[g] = qb * QPB + wave * (32 * QH) + g * 32 + ql;
    qvalid[g] = q[g] < T;
  }
  const int ref = a.ref_index ? a.ref_index[n] : -1;
  const int ns = a.frame_mod > 0 ? n % a.frame_mod : n;   // source frame of q / k / v^T (attn_args.h)

  // zero the padding that is never overwritten: K columns [D, DQ*16) and V^T rows [D, DO*32)
  for (int i = tid; i < 2 * KV * KROW; i += NT) (&sK[0][0])[i] = (f16)0.f;
  for (int i = tid; i < 2 * DO * 32 * VROW; i += NT) (&sV[0][0])[i] = (f16)0.f;
  if (ONES) {
    __syncthreads();
    for (int i = tid; i < 2 * VROW; i += NT) sV[i / VROW][D * VROW + (i % VROW)] = (f16)1.f;
  }

  // Q fragments (B operand of S^T = K Q^T): lane (q = ql, hi) holds Q[q][16 kk + 8 hi .. +7]
  f16x8 qf[QH][DQ];
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    const f16* qp = a.q + ((int64_t)ns * T + (qvalid[g] ? q[g] : 0)) * a.ldq + h * D;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const int d0 = kk * 16 + hi * 8;
      U4H8 t;
      t.u = u32x4{0u, 0u, 0u, 0u};
      if (qvalid[g] && d0 < D) t.u = *(const u32x4*)(qp + d0);
      qf[g][kk] = t.h;
    }
  }

  const int nts = (T + KV - 1) / KV;
  const int ntiles = nts * (ref >= 0 ? 2 : 1);

  // per-thread staging bookkeeping, hoisted out of the tile loop
  bool ch_ok[NCH];
  int k_key[NCH], k_dc8[NCH], v_dr[NCH], v_k8[NCH];
  int k_lds[NCH], v_lds_lo[NCH], v_lds_hi[NCH];
  uint32_t koff_s[NCH], koff_r[NCH], voff_s[NCH], voff_r[NCH];  // FAST: element offsets inside a tile
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
    ch_ok[i] = c < KV * DC;
    const int cc = ch_ok[i] ? c : 0;       // idle slots re-read chunk 0 (valid memory) and skip the LDS store
    const int key = cc / DC, dc = cc - key * DC;
    k_key[i] = key;
    k_dc8[i] = dc * 8;
    k_lds[i] = key * KROW + dc * 8;
    const int dr = cc >> 3, kc = cc & 7;
    v_dr[i] = dr;
    v_k8[i] = kc * 8;
    // 16-key group permutation: [k0-3 | k8-11 | k4-7 | k12-15]
    const int gb = (kc >> 1) * 16;
    v_lds_lo[i] = dr * VROW + gb + ((kc & 1) ? 4 : 0);
    v_lds_hi[i] = dr * VROW + gb + ((kc & 1) ? 12 : 8);
    koff_s[i] = (uint32_t)(key * (int)a.ldk + dc * 8);
    koff_r[i] = (uint32_t)(key * (int)a.ldkr + dc * 8);
    voff_s[i] = (uint32_t)(dr * (int)a.ldvt + kc * 8);
    voff_r[i] = (uint32_t)(dr * (int)a.ldvtr + kc * 8);
  }

  u32x4 rk[NCH], rv[NCH];
  auto load_tile = [&](int t) {
    const bool second = t >= nts;
    const int tt = second ? t - nts : t;
    const int64_t ldk = second ? a.ldkr : a.ldk;
    const int64_t ldv = second ? a.ldvtr : a.ldvt;
    const int64_t tok0 = (int64_t)(second ? ref : ns) * T + (int64_t)tt * KV;
    const f16* kb = (second ? a.kref : a.k) + tok0 * ldk + h * (second ? a.kr_hs : a.k_hs);
    const f16* vb = (second ? a.vtref : a.vt) + (int64_t)h * D * ldv + tok0;
    if (FAST) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        rk[i] = *(const u32x4*)(kb + (second ? koff_r[i] : koff_s[i]));
        rv[i] = *(const u32x4*)(vb + (second ? voff_r[i] : voff_s[i]));
      }
      return;
    }
    const bool vvec = second ? a.vtref_vec_ok : a.vt_vec_ok;
    const int left = T - tt * KV;          // valid keys in this tile (>= KV except in a segment's last tile)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      rk[i] = u32x4{0u, 0u, 0u, 0u};
      rv[i] = u32x4{0u, 0u, 0u, 0u};
      if (ch_ok[i]) {
        if (k_key[i] < left) rk[i] = *(const u32x4*)(kb + (int64_t)k_key[i] * ldk + k_dc8[i]);
        const f16* vp = vb + (int64_t)v_dr[i] * ldv + v_k8[i];
        if (vvec && v_k8[i] + 8 <= left) {
          rv[i] = *(const u32x4*)vp;
        } else {
          U4H8 t8;
#pragma unroll
          for (int e = 0; e < 8; ++e) t8.e[e] = (v_k8[i] + e < left) ? vp[e] : (f16)0.f;
          rv[i] = t8.u;
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (ch_ok[i]) {
        *(u32x4*)(&sK[buf][k_lds[i]]) = rk[i];
        *(u32x2*)(&sV[buf][v_lds_lo[i]]) = u32x2{rv[i].x, rv[i].y};
        *(u32x2*)(&sV[buf][v_lds_hi[i]]) = u32x2{rv[i].z, rv[i].w};
      }
    }
  };

  f32x16 o[QH][DO];
  float m_run[QH];           // reference max of the exponent (raised lazily), raw-score units
  float l_run[QH];           // denominator when there is no spare V^T row (!ONES)
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    m_run[g] = -INFINITY;
    l_run[g] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DO; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[g][dt][r] = 0.f;
  }
  const float c2 = a.scale_log2e;
  const int ka_off = ql * KROW + hi * 8;
  const int va_off = ql * VROW + hi * 8;

  __syncthreads();  // padding zeros / ones visible before the first tile is written
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile(t + 1);

    // ---- S^T = K Q^T : two 32-key x 32-query tiles per query group ---------------------------------
    f32x16 s0[QH], s1[QH];
#pragma unroll
    for (int g = 0; g < QH; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[g][r] = s1[g][r] = 0.f;
    const f16* kbuf = &sK[buf][ka_off];
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const f16x8 a0 = *(const f16x8*)(kbuf + kk * 16);
      const f16x8 a1 = *(const f16x8*)(kbuf + 32 * KROW + kk * 16);
#pragma unroll
      for (int g = 0; g < QH; ++g) {
        s0[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[g][kk], s0[g], 0, 0, 0);
        s1[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[g][kk], s1[g], 0, 0, 0);
      }
    }
    // mask keys beyond the segment length (last tile of a segment only)
    const int tt = t >= nts ? t - nts : t;
    if (!FAST && tt * KV + KV > T) {
#pragma unroll
      for (int g = 0; g < QH; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = tt * KV + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= T) s0[g][r] = -INFINITY;
          if (key + 32 >= T) s1[g][r] = -INFINITY;
        }
    }
    // V^T fragments of the tile (A operand of O^T += V^T P^T), shared by the query groups
    const f16* vbuf = &sV[buf][va_off];
#pragma unroll
    for (int g = 0; g < QH; ++g) {
      // ---- online softmax (one query per lane; partner lane^32 holds the other 32 keys) ------------
      float mx = fmaxf(fmaxf(s0[g][0], s0[g][1]), s0[g][2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s0[g][r]), s0[g][r + 1]);
      mx = fmaxf(fmaxf(mx, s0[g][15]), s1[g][0]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) mx = fmaxf(fmaxf(mx, s1[g][r]), s1[g][r + 1]);
      mx = fmaxf(mx, s1[g][15]);
      mx = xor32_max(mx);
      if (__any((mx - m_run[g]) * c2 > RESCALE_LOG2)) {   // wave-uniform; always taken on the first tile
        const float m_new = fmaxf(m_run[g], mx);
        const float alpha = fast_exp2((m_run[g] - m_new) * c2);
        m_run[g] = m_new;
        l_run[g] *= alpha;
#pragma unroll
        for (int dt = 0; dt < DO; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[g][dt][r] *= alpha;
      }
      const float mb = m_run[g] * c2;
      // P^T fragments (B operand): k-slot (hi, j) of 16-key group gk <-> accumulator reg 8*(gk&1)+j of tile gk>>1
      U4H8 pb[4];
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float p00 = fast_exp2(fmaf(s0[g][j], c2, -mb)), p01 = fast_exp2(fmaf(s0[g][j + 1], c2, -mb));
        const float p10 = fast_exp2(fmaf(s0[g][8 + j], c2, -mb)), p11 = fast_exp2(fmaf(s0[g][9 + j], c2, -mb));
        const float p20 = fast_exp2(fmaf(s1[g][j], c2, -mb)), p21 = fast_exp2(fmaf(s1[g][j + 1], c2, -mb));
        const float p30 = fast_exp2(fmaf(s1[g][8 + j], c2, -mb)), p31 = fast_exp2(fmaf(s1[g][9 + j], c2, -mb));
        if (!ONES) l_run[g] += ((p00 + p01) + (p10 + p11)) + ((p20 + p21) + (p30 + p31));
        pb[0].u[j >> 1] = pk_f16(p00, p01);
        pb[1].u[j >> 1] = pk_f16(p10, p11);
        pb[2].u[j >> 1] = pk_f16(p20, p21);
        pb[3].u[j >> 1] = pk_f16(p30, p31);
      }
      if (QH > 1) __builtin_amdgcn_sched_barrier(0);   // keep this group's P V MFMAs ahead of the next group's softmax
      // ---- O^T += V^T P^T ----------------------------------------------------------------------
#pragma unroll
      for (int dt = 0; dt < DO; ++dt) {
#pragma unroll
        for (int gk = 0; gk < 4; ++gk) {
          const f16x8 av = *(const f16x8*)(vbuf + dt * 32 * VROW + gk * 16);
          o[g][dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb[gk].h, o[g][dt], 0, 0, 0);
        }
      }
      if (QH > 1) __builtin_amdgcn_sched_barrier(0);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
#pragma unroll
  for (int g = 0; g < QH; ++g) {
    float l_tot;
    if (ONES) {
      const float lv = o[g][LT][L_REG];              // row D of O^T: held by the half-wave with hi == L_HI
      const float lp = __shfl_xor(lv, 32, 64);
      l_tot = (hi == L_HI) ? lv : lp;
    } else {
      l_tot = l_run[g] + __shfl_xor(l_run[g], 32, 64);
    }
    const float inv = 1.0f / l_tot;
    if (qvalid[g]) {
      f16* op = a.out + ((int64_t)n * T + q[g]) * a.ldo + h * D;
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

// one 32-query group per wave (two — the QH = 2 instantiation — measured no faster at twice the registers, round 2)
template <int D>
int launch_ref_attn(const RefAttnArgs& a, int Nf, hipStream_t stream) {
  dim3 grid((unsigned)((a.T + 127) / 128), (unsigned)a.heads, (unsigned)Nf);
  AnipProfScope prof_(ANIP_K_REF_ATTN, (void*)stream);
  const bool fits32 = (int64_t)KV * a.ldk < (1ll << 31) && (int64_t)KV * a.ldkr < (1ll << 31) &&
                      (int64_t)(D + 32) * a.ldvt < (1ll << 31) && (int64_t)(D + 32) * a.ldvtr < (1ll << 31);
  const bool fast = (a.T % KV) == 0 && a.vt_vec_ok && (a.ref_index == nullptr || a.vtref_vec_ok) && fits32;
  if (fast) hipLaunchKernelGGL((ref_attn_kernel<D, true, 1>), grid, dim3(NT), 0, stream, a);
  else hipLaunchKernelGGL((ref_attn_kernel<D, false, 1>), grid, dim3(NT), 0, stream, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// temporal attention (VersatileAttention, src/models/motion_module.py:351-388): for every (b, pixel, head) a
// softmax attention over the F <= 32 frames.  HBM-bound: 8 B per channel and token.
// One 256-thread block per (b, pixel, channel group of <= 320 channels = hpg whole heads): the q | k | v segments of
// the group are contiguous 2*CG-byte pieces of every frame's token row, staged with full-line 16-B loads into
// LDS [F][CG]; then thread (query frame i, head h) — h fastest, so a wave's lanes read a handful of distinct k / v
// rows (LDS broadcast) and write neighbouring output segments — computes its 1 x F score row with v_dot2_f32_f16,
// the softmax in registers and the d outputs 8 channels at a time (fp16 inputs, fp32 accumulation).
// ---------------------------------------------------------------------------------------------------
// (round 2) The kernel is VALU-bound (rocprofv3: SQ_ACTIVE_INST_VALU > 100 % of a SIMD's cycles over its waves, no
// MFMA), and 4/5 of that was P V: one v_cvt + one v_fma per (key, channel).  V is therefore staged as KEY PAIRS —
// (v[2j][c], v[2j+1][c]) packed in one dword, interleaved with v_perm while the two frames' 16-B chunks are on their
// way to LDS — and the probabilities are packed to fp16 pairs once per query, so P V is one v_dot2_f32_f16 per
// (key pair, channel) with fp32 accumulation: 4x fewer VALU instructions.  (The probabilities of the reference's fp16
// SDPA are fp16 too.)
template <int FMAX>
__global__ __launch_bounds__(NT) void temporal_attn_kernel(const f16* __restrict__ qkv, f16* __restrict__ out, int B,
                                                          int F, int T, int heads, int d, float scale_log2e, int hpg,
                                                          int SL) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  constexpr int FP = FMAX / 2;              // key pairs
  const int tid = threadIdx.x;
  const int C = heads * d, CG = hpg * d, ngroups = heads / hpg;
  const int g = blockIdx.x % ngroups;
  const int64_t bt = blockIdx.x / ngroups;
  const int t = (int)(bt % T), b = (int)(bt / T);
  const int F2 = (F + 1) >> 1;               // frame pairs staged
  f16* sq = (f16*)dsm;                       // [2*F2][CG]
  f16* sk = sq + 2 * F2 * CG;                // [2*F2][CG]
  f16* sv = sk + 2 * F2 * CG;                // [F2][CG] key pairs: 2 halfs per channel
  const int cpr = CG >> 3;                   // 16-B chunks per (frame, q|k|v) segment
  const int total = F2 * 3 * cpr;            // items: (frame pair, segment, chunk)
  for (int base = 0; base < total; base += NT * 4) {
    u32x4 va[4], vb[4];
    int fp_[4], seg_[4], off_[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = base + u * NT + tid;
      const bool ok = c < total;
      const int cc = ok ? c : 0;
      const int fp = cc / (3 * cpr), r = cc - fp * (3 * cpr);
      const int seg = r / cpr, off = r - seg * cpr;
      fp_[u] = ok ? fp : -1;
      seg_[u] = seg;
      off_[u] = off;
      const int f0 = 2 * fp, f1 = min(2 * fp + 1, F - 1);   // odd F: the missing partner re-reads the last frame
      va[u] = *(const u32x4*)(qkv + (((int64_t)b * F + f0) * T + t) * (3 * (int64_t)C) + seg * C + g * CG + off * 8);
      vb[u] = *(const u32x4*)(qkv + (((int64_t)b * F + f1) * T + t) * (3 * (int64_t)C) + seg * C + g * CG + off * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (fp_[u] < 0) continue;
      if (seg_[u] < 2) {
        f16* dst = (seg_[u] == 0 ? sq : sk) + (2 * fp_[u]) * CG + off_[u] * 8;
        *(u32x4*)dst = va[u];
        *(u32x4*)(dst + CG) = vb[u];
      } else {
        // (a0 a1 | a2 a3 | ..) x (b0 b1 | ..) -> (a0 b0 | a1 b1 | a2 b2 | ...): low / high halves of each dword pair
        u32x4 lo, hi;
        lo.x = __builtin_amdgcn_perm(vb[u].x, va[u].x, 0x05040100u);
        lo.y = __builtin_amdgcn_perm(vb[u].x, va[u].x, 0x07060302u);
        lo.z = __builtin_amdgcn_perm(vb[u].y, va[u].y, 0x05040100u);
        lo.w = __builtin_amdgcn_perm(vb[u].y, va[u].y, 0x07060302u);
        hi.x = __builtin_amdgcn_perm(vb[u].z, va[u].z, 0x05040100u);
        hi.y = __builtin_amdgcn_perm(vb[u].z, va[u].z, 0x07060302u);
        hi.z = __builtin_amdgcn_perm(vb[u].w, va[u].w, 0x05040100u);
        hi.w = __builtin_amdgcn_perm(vb[u].w, va[u].w, 0x07060302u);
        f16* dst = sv + ((int64_t)fp_[u] * CG + off_[u] * 8) * 2;
        *(u32x4*)dst = lo;
        *(u32x4*)(dst + 8) = hi;
      }
    }
  }
  __syncthreads();
  // thread = (item, d-slice): item = (query frame i, head h), h fastest; the head's d/8 chunks are split over SL
  // adjacent lanes (SL a power of two) whose partial scores are summed with xor-shuffles
  const int item = tid / SL, sl = tid - item * SL;
  if (item >= F * hpg) return;
  const int i = item / hpg, h = item - i * hpg;
  const int dc = d >> 3;
  const int c0 = (dc * sl) / SL, c1 = (dc * (sl + 1)) / SL;
  float s[FMAX];
#pragma unroll
  for (int j = 0; j < FMAX; ++j) s[j] = 0.f;
  for (int c = c0; c < c1; ++c) {
    U4H8 qa;
    qa.u = *(const u32x4*)(sq + i * CG + h * d + c * 8);
    const f16x2 q0 = {qa.e[0], qa.e[1]}, q1 = {qa.e[2], qa.e[3]}, q2 = {qa.e[4], qa.e[5]}, q3 = {qa.e[6], qa.e[7]};
#pragma unroll
    for (int j = 0; j < FMAX; ++j) {
      if (j < F) {
        U4H8 kb;
        kb.u = *(const u32x4*)(sk + j * CG + h * d + c * 8);
        float a = s[j];
        a = __builtin_amdgcn_fdot2(q0, f16x2{kb.e[0], kb.e[1]}, a, false);
        a = __builtin_amdgcn_fdot2(q1, f16x2{kb.e[2], kb.e[3]}, a, false);
        a = __builtin_amdgcn_fdot2(q2, f16x2{kb.e[4], kb.e[5]}, a, false);
        a = __builtin_amdgcn_fdot2(q3, f16x2{kb.e[6], kb.e[7]}, a, false);
        s[j] = a;
      }
    }
  }
  for (int o = 1; o < SL; o <<= 1) {
#pragma unroll
    for (int j = 0; j < FMAX; ++j)
      if (j < F) s[j] += __shfl_xor(s[j], o, 64);
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < FMAX; ++j)
    if (j < F) mx = fmaxf(mx, s[j]);
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < FMAX; ++j) {
    s[j] = (j < F) ? __builtin_amdgcn_exp2f((s[j] - mx) * scale_log2e) : 0.f;   // keys >= F (odd-F partner): weight 0
    sum += s[j];
  }
  const float inv = 1.0f / sum;
  f16x2 pp[FP];
#pragma unroll
  for (int jp = 0; jp < FP; ++jp) pp[jp] = f16x2{(f16)s[2 * jp], (f16)s[2 * jp + 1]};
  f16* op = out + (((int64_t)b * F + i) * T + t) * (int64_t)C + g * CG + h * d;
  for (int c = c0; c < c1; ++c) {
    float acc[8];
#pragma unroll
    for (int x = 0; x < 8; ++x) acc[x] = 0.f;
#pragma unroll
    for (int jp = 0; jp < FP; ++jp) {
      if (jp < F2) {
        const f16* vp = sv + ((int64_t)jp * CG + h * d + c * 8) * 2;
        union { u32x4 u; f16x2 p[4]; } v0, v1;
        v0.u = *(const u32x4*)vp;
        v1.u = *(const u32x4*)(vp + 8);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          acc[x] = __builtin_amdgcn_fdot2(pp[jp], v0.p[x], acc[x], false);
          acc[4 + x] = __builtin_amdgcn_fdot2(pp[jp], v1.p[x], acc[4 + x], false);
        }
      }
    }
    U4H8 ov;
#pragma unroll
    for (int x = 0; x < 8; ++x) ov.e[x] = (f16)(acc[x] * inv);
    *(u32x4*)(op + c * 8) = ov.u;
  }
}


// ---------------------------------------------------------------------------------------------------
// temporal attention on MFMA for the 16-frame windows every configuration of the path runs (F == 16, round 4).
// temporal_attn_kernel above is VALU-bound (one v_dot2 per key pair and channel pair: 3.3 TB/s of 336 MB at the 64x64 level);
// here a (pixel, head) problem is five to fifteen 16x16x32 MFMAs:
//   S^T = K Q^T  (16 keys x 16 queries, contraction over d in ceil(d / 32) steps; lanes beyond d contribute zeros),
//   softmax over the keys of a query = over the 4 accumulator registers and the 4 lane groups of a column,
//   O^T = V^T P^T (d x 16 queries; the 16 keys occupy k-slot groups 0 and 2 of the 32: the probabilities reach the B layout by
//   one v_permlane16_swap per dword, V^T is staged transposed — [channel][key] — with zeros read for the other two groups).
// One block per (b, pixel, channel group of <= 320 channels) as before: q | k | v of the 16 frames staged with full-line loads
// (V transposed on the way, frame pairs interleaved with v_perm), wave w takes heads w, w + 4, ..; the output tile goes back
// through LDS so that every frame's row segment leaves in 16-B pieces.
// ---------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(NT) void temporal_attn16_kernel(const f16* __restrict__ qkv, f16* __restrict__ out, int T, int heads,
                                                            float scale_log2e, int hpg) {
  constexpr int F = 16;
  constexpr int DQ = (D + 31) / 32;          // contraction steps of K Q^T
  constexpr int DT = (D + 15) / 16;          // 16-row tiles of O^T
  constexpr int VS = 32;                     // bytes per V^T row: 16 keys = two 16-B halves, swapped in rows with bit 3 set (conflict-free b128 reads down 16 rows)
  extern __shared__ __attribute__((aligned(16))) char tsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = heads * D, CG = hpg * D, ngroups = heads / hpg;
  const int RS = CG * 2 + 16;                // bytes per q / k / o row (pad: conflict-free b128 reads down a column of rows)
  const int g = blockIdx.x % ngroups;
  const int64_t bt = blockIdx.x / ngroups;
  const int t = (int)(bt % T);
  const int64_t b = bt / T;
  char* sq = tsm;
  char* sk = sq + F * RS;
  char* so = sq;                             // the output tile takes the place of q: head h's columns are read (q) and written (o) by
                                             // the one wave that owns head h, after its S^T is complete
  char* svt = sk + F * RS;                   // (CG + 16) rows: a head's last O^T tile may read up to 15 rows past its channels
  const int cpr = CG >> 3;                   // 16-B chunks per (frame, q | k | v) segment
  const int64_t fstride = (int64_t)T * 3 * C;
  const int nqk = 2 * F * cpr, nv = (F / 2) * cpr;
  constexpr int IQK = (2 * F * 40 + NT - 1) / NT, IV = ((F / 2) * 40 + NT - 1) / NT;   // register slots at the widest group (320 channels)
  u32x4 rqk[IQK], rva[IV], rvb[IV];
  auto issue_loads = [&](int t) {
    const f16* base = qkv + ((b * F) * (int64_t)T + t) * (3 * (int64_t)C) + g * CG;
#pragma unroll
    for (int i = 0; i < IQK; ++i) {
      const int it = tid + i * NT;
      if (it < nqk) {
        const int seg = it / (F * cpr), r = it - seg * (F * cpr);
        const int f = r / cpr, c8 = r - f * cpr;
        rqk[i] = *(const u32x4*)(base + f * fstride + seg * C + c8 * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < IV; ++i) {
      const int it = tid + i * NT;
      if (it < nv) {
        const int fp = it / cpr, c8 = it - fp * cpr;
        rva[i] = *(const u32x4*)(base + (2 * fp) * fstride + 2 * C + c8 * 8);
        rvb[i] = *(const u32x4*)(base + (2 * fp + 1) * fstride + 2 * C + c8 * 8);
      }
    }
  };
  auto stage = [&]() {
    // q, k: rows as they are; v: transposed, a frame pair per dword
#pragma unroll
    for (int i = 0; i < IQK; ++i) {
      const int it = tid + i * NT;
      if (it < nqk) {
        const int seg = it / (F * cpr), r = it - seg * (F * cpr);
        const int f = r / cpr, c8 = r - f * cpr;
        *(u32x4*)((seg ? sk : sq) + f * RS + c8 * 16) = rqk[i];
      }
    }
#pragma unroll
    for (int i = 0; i < IV; ++i) {
      const int it = tid + i * NT;
      if (it < nv) {
        const int fp = it / cpr, c8 = it - fp * cpr;
        const u32x4 va = rva[i], vb = rvb[i];
        char* dst = svt + (c8 * 8) * VS + (((fp >> 2) ^ (c8 & 1)) << 4) + (fp & 3) * 4;   // rows c8*8 .. +7 share bit 3 = c8 & 1
        // (a0 a1 | a2 a3 | ..) x (b0 b1 | ..) -> channel e: (a_e, b_e)
        *(uint32_t*)(dst + 0 * VS) = __builtin_amdgcn_perm(vb.x, va.x, 0x05040100u);
        *(uint32_t*)(dst + 1 * VS) = __builtin_amdgcn_perm(vb.x, va.x, 0x07060302u);
        *(uint32_t*)(dst + 2 * VS) = __builtin_amdgcn_perm(vb.y, va.y, 0x05040100u);
        *(uint32_t*)(dst + 3 * VS) = __builtin_amdgcn_perm(vb.y, va.y, 0x07060302u);
        *(uint32_t*)(dst + 4 * VS) = __builtin_amdgcn_perm(vb.z, va.z, 0x05040100u);
        *(uint32_t*)(dst + 5 * VS) = __builtin_amdgcn_perm(vb.z, va.z, 0x07060302u);
        *(uint32_t*)(dst + 6 * VS) = __builtin_amdgcn_perm(vb.w, va.w, 0x05040100u);
        *(uint32_t*)(dst + 7 * VS) = __builtin_amdgcn_perm(vb.w, va.w, 0x07060302u);
      }
    }
  };
  // every load of the block in flight before the first LDS write (written as load -> store pairs the block reached 3.1 TB/s at
  // the 64x64 level whatever its arithmetic cost; a block walking several pixels with the next one's rows prefetched: no faster)
  issue_loads(t);
  stage();
  __syncthreads();
  const int col = lane & 15, lg = lane >> 4;     // accumulator column (query / key row of an operand) and lane group
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
  for (int h = wave; h < hpg; h += NT / 64) {
    const int hb = h * D * 2;
    // ---- S^T = K Q^T ---------------------------------------------------------------------------------------------------
    f32x4 sT = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const int ch = kk * 32 + lg * 8;            // first channel of this lane's 8
      U4H8 ka, qb;
      ka.u = zero4;
      qb.u = zero4;
      if (ch < D) {
        ka.u = *(const u32x4*)(sk + col * RS + hb + ch * 2);
        qb.u = *(const u32x4*)(sq + col * RS + hb + ch * 2);
      }
      sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka.h, qb.h, sT, 0, 0, 0);
    }
    // ---- softmax over the 16 keys of query `col`: registers (keys 4 lg .. + 3) x lane groups -------------------------------
    float mx = fmaxf(fmaxf(sT[0], sT[1]), fmaxf(sT[2], sT[3]));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) pr[r] = __builtin_amdgcn_exp2f((sT[r] - mx) * scale_log2e);
    // ---- P^T as the B operand: lane group 0 / 1 <- keys 0-7, 2 / 3 <- keys 8-15 (the duplicates meet zeros of V^T) ---------
    H2U p01, p23;
    p01.h = __builtin_amdgcn_cvt_pkrtz(pr[0], pr[1]);
    p23.h = __builtin_amdgcn_cvt_pkrtz(pr[2], pr[3]);
    // the denominator is the sum of the TRUNCATED probabilities the P V product sees (round 5; the fp32 sum of the untruncated
    // ones biased every output down by up to 2^-11 relative — the reference-attention kernels take it from the same MFMA)
    float sum = ((float)p01.h[0] + (float)p01.h[1]) + ((float)p23.h[0] + (float)p23.h[1]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const auto s0 = __builtin_amdgcn_permlane16_swap(p01.u, p01.u, false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(p23.u, p23.u, false, false);
    U4H8 pb;
    pb.u = u32x4{s0[0], s1[0], s0[1], s1[1]};     // keys 4 e .. + 3 of the even group, then of the odd group
    // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      U4H8 va;
      va.u = zero4;
      const int vrow = h * D + dt * 16 + col;
      if ((lg & 1) == 0) va.u = *(const u32x4*)(svt + vrow * VS + (((lg >> 1) ^ ((vrow >> 3) & 1)) << 4));
      f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(va.h, pb.h, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const int c0 = dt * 16 + lg * 4;            // channels c0 .. c0 + 3 of the head, query `col`
      if (c0 < D) {
        union { u32x2 u; f16 e[4]; } ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(o[r] * inv);
        *(u32x2*)(so + col * RS + hb + c0 * 2) = ov.u;
      }
    }
  }
  __syncthreads();
  f16* ob = out + ((b * F) * (int64_t)T + t) * (int64_t)C + g * CG;
  for (int it = tid; it < F * cpr; it += NT) {
    const int f = it / cpr, c8 = it - f * cpr;
    *(u32x4*)(ob + f * ((int64_t)T * C) + c8 * 8) = *(const u32x4*)(so + f * RS + c8 * 16);
  }
}

}  // namespace

extern "C" int anip_ref_attention_ex(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt,
                                     int64_t ldvt, const void* kref, int64_t ldkr, const void* vtref, int64_t ldvtr,
                                     const int* ref_index, void* out, int64_t ldo, int Nf, int T, int heads, int d,
                                     float scale, int64_t k_head_stride, int64_t kref_head_stride, int flags,
                                     void* stream) {
  ANIP_REQUIRE(q && k && vt && out, "anip_ref_attention: null pointer");
  ANIP_REQUIRE(Nf > 0 && T > 0 && heads > 0, "anip_ref_attention: bad sizes");
  ANIP_REQUIRE((ldq & 7) == 0 && (ldk & 7) == 0 && (ldo & 3) == 0, "anip_ref_attention: ldq/ldk %% 8, ldo %% 4 required");
  ANIP_REQUIRE(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)vt & 15) == 0 && ((uintptr_t)out & 7) == 0,
               "anip_ref_attention: misaligned base pointer");
  if (ref_index != nullptr) {
    ANIP_REQUIRE(kref && vtref, "anip_ref_attention: ref_index given without kref/vtref");
    ANIP_REQUIRE((ldkr & 7) == 0 && ((uintptr_t)kref & 15) == 0 && ((uintptr_t)vtref & 15) == 0, "anip_ref_attention: misaligned reference bank");
  }
  RefAttnArgs a;
  a.q = (const f16*)q; a.ldq = ldq;
  a.k = (const f16*)k; a.ldk = ldk;
  a.vt = (const f16*)vt; a.ldvt = ldvt;
  a.kref = (const f16*)kref; a.ldkr = ldkr;
  a.k_hs = k_head_stride > 0 ? k_head_stride : d;
  a.kr_hs = kref_head_stride > 0 ? kref_head_stride : d;
  a.vtref = (const f16*)vtref; a.ldvtr = ldvtr;
  a.ref_index = ref_index;
  a.out = (f16*)out; a.ldo = ldo;
  a.T = T; a.heads = heads;
  a.frame_mod = (int)((unsigned)flags >> 16);
  ANIP_REQUIRE(a.frame_mod == 0 || (a.frame_mod > 0 && Nf % a.frame_mod == 0), "anip_ref_attention: Nf=%d is not a multiple of the frame modulus %d",
               Nf, a.frame_mod);
  const bool q_log2 = (flags & ANIP_ATTN_Q_LOG2_SCALED) != 0;   // scores are base-2 exponents already: `scale` is not applied
  a.scale_log2e = q_log2 ? 1.0f : scale * 1.4426950408889634f;
  a.vt_vec_ok = ((T & 7) == 0) && ((ldvt & 7) == 0);
  a.vtref_vec_ok = ((T & 7) == 0) && ((ldvtr & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
  if (q_log2) {
    const int r = anip_ref_attn_dma_try(a, Nf, d, s);     // T % 256 == 0, d in {40, 80, 160}: the LDS-DMA kernel (attn_dma.hip)
    if (r < 0) return r;
    if (r > 0) {
      ANIP_LAUNCH_CHECK("anip_ref_attention");
      return 0;
    }
  }
  switch (d) {
    case 8: launch_ref_attn<8>(a, Nf, s); break;
    case 16: launch_ref_attn<16>(a, Nf, s); break;
    case 32: launch_ref_attn<32>(a, Nf, s); break;
    case 40: launch_ref_attn<40>(a, Nf, s); break;
    case 64: launch_ref_attn<64>(a, Nf, s); break;
    case 80: launch_ref_attn<80>(a, Nf, s); break;
    case 88: launch_ref_attn<88>(a, Nf, s); break;  // PoseGuider self-attention (16 heads x 88, pose_guider.py:86-89)
    case 160: launch_ref_attn<160>(a, Nf, s); break;
    default:
      anip_set_error("anip_ref_attention: unsupported head dim %d (have 8,16,32,40,64,80,88,160)", d);
      return -1;
  }
  ANIP_LAUNCH_CHECK("anip_ref_attention");
  return 0;
}

extern "C" int anip_ref_attention(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt,
                                  int64_t ldvt, const void* kref, int64_t ldkr, const void* vtref, int64_t ldvtr,
                                  const int* ref_index, void* out, int64_t ldo, int Nf, int T, int heads, int d,
                                  float scale, int64_t k_head_stride, int64_t kref_head_stride, void* stream) {
  return anip_ref_attention_ex(q, ldq, k, ldk, vt, ldvt, kref, ldkr, vtref, ldvtr, ref_index, out, ldo, Nf, T, heads, d, scale,
                               k_head_stride, kref_head_stride, 0, stream);
}

extern "C" int anip_temporal_attention(const void* qkv, void* out, int B, int F, int T, int heads, int d,
                                       float scale, void* stream) {
  ANIP_REQUIRE(qkv && out, "anip_temporal_attention: null pointer");
  ANIP_REQUIRE(B > 0 && T > 0 && heads > 0 && F > 0 && F <= 32, "anip_temporal_attention: need 1 <= F <= 32 (F=%d)", F);
  ANIP_REQUIRE((d & 7) == 0 && d > 0, "anip_temporal_attention: head dim %d must be a multiple of 8", d);
  ANIP_REQUIRE((((uintptr_t)qkv | (uintptr_t)out) & 15) == 0, "anip_temporal_attention: pointers must be 16-B aligned");
  // heads per channel group: whole heads, <= 320 channels, one (query frame, head) item per thread
  int hpg = 1;
  for (int c = heads; c >= 1; --c)
    if (heads % c == 0 && c * d <= 320 && c * F <= NT) { hpg = c; break; }
  ANIP_REQUIRE(hpg * F <= NT, "anip_temporal_attention: F=%d too large", F);
  const float sl2 = scale * 1.4426950408889634f;
  const int64_t blocks = (int64_t)B * T * (heads / hpg);
  ANIP_REQUIRE(blocks < (1ll << 31), "anip_temporal_attention: grid too large");
  if (F == 16 && (d == 40 || d == 80 || d == 160)) {
    const int CG = hpg * d;
    const size_t lds16 = (size_t)2 * 16 * (CG * 2 + 16) + (size_t)(CG + 16) * 32;
    const unsigned nblk = (unsigned)blocks;
    AnipProfScope prof_(ANIP_K_TEMPORAL_ATTN, (void*)stream);
    if (d == 40)
      hipLaunchKernelGGL(temporal_attn16_kernel<40>, dim3(nblk), dim3(NT), lds16, (hipStream_t)stream, (const f16*)qkv, (f16*)out, T, heads, sl2, hpg);
    else if (d == 80)
      hipLaunchKernelGGL(temporal_attn16_kernel<80>, dim3(nblk), dim3(NT), lds16, (hipStream_t)stream, (const f16*)qkv, (f16*)out, T, heads, sl2, hpg);
    else
      hipLaunchKernelGGL(temporal_attn16_kernel<160>, dim3(nblk), dim3(NT), lds16, (hipStream_t)stream, (const f16*)qkv, (f16*)out, T, heads, sl2, hpg);
    ANIP_LAUNCH_CHECK("anip_temporal_attention");
    return 0;
  }
  const int F2 = (F + 1) / 2;
  const size_t lds = (size_t)3 * (2 * F2) * hpg * d * sizeof(f16);   // q, k: 2*F2 rows; v: F2 rows of key pairs
  ANIP_REQUIRE(lds <= 65536, "anip_temporal_attention: LDS budget exceeded (F=%d d=%d)", F, d);
  int SL = 1;                               // d-slices per item: power of two, <= chunks per head, all 256 threads used
  while (SL * 2 * hpg * F <= NT && SL * 2 <= (d >> 3)) SL *= 2;
  {
    AnipProfScope prof_(ANIP_K_TEMPORAL_ATTN, (void*)stream);
    if (F <= 8)
      hipLaunchKernelGGL(temporal_attn_kernel<8>, dim3((unsigned)blocks), dim3(NT), lds, (hipStream_t)stream, (const f16*)qkv,
                         (f16*)out, B, F, T, heads, d, sl2, hpg, SL);
    else if (F <= 16)
      hipLaunchKernelGGL(temporal_attn_kernel<16>, dim3((unsigned)blocks), dim3(NT), lds, (hipStream_t)stream, (const f16*)qkv,
                         (f16*)out, B, F, T, heads, d, sl2, hpg, SL);
    else
      hipLaunchKernelGGL(temporal_attn_kernel<32>, dim3((unsigned)blocks), dim3(NT), lds, (hipStream_t)stream, (const f16*)qkv,
                         (f16*)out, B, F, T, heads, d, sl2, hpg, SL);
  }
  ANIP_LAUNCH_CHECK("anip_temporal_attention");
  return 0;
}
