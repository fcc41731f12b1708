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
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int KV = 64;       // keys per tile
constexpr int VROW = KV + 8; // fp16 elements per V^T LDS row (144 B = 9 x 16 B: odd -> conflict-free b128 reads)

struct RefAttnArgs {
  const f16* q; int64_t ldq;
  const f16* k; int64_t ldk;
  const f16* vt; int64_t ldvt;
  const f16* kref; int64_t ldkr;
  const f16* vtref; int64_t ldvtr;
  const int* ref_index;
  f16* out; int64_t ldo;
  int T, heads;
  float scale_log2e;
  int vt_vec_ok, vtref_vec_ok;
};

template <int D>
__global__ __launch_bounds__(NT, (D > 96 ? 1 : 2)) void ref_attn_kernel(const RefAttnArgs a) {
  constexpr int DQ = (D + 15) / 16;        // 16-wide contraction chunks of Q K^T
  constexpr int DO = (D + 31) / 32;        // 32-row output tiles of O^T
  constexpr int KROW = DQ * 16 + 8;        // fp16 per K LDS row; (2*DQ+1) 16-B slots: odd
  constexpr int DC = D / 8;                // 16-B chunks per head row
  constexpr int NCH = (KV * DC + NT - 1) / NT;  // staged chunks per thread (K and V^T each)
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  __shared__ __attribute__((aligned(16))) f16 sK[2][KV * KROW];
  __shared__ __attribute__((aligned(16))) f16 sV[2][DO * 32 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int T = a.T;
  const int h = blockIdx.y, n = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + ql;
  const bool qvalid = q < T;
  const int ref = a.ref_index ? a.ref_index[n] : -1;

  // zero the padding that is never overwritten: K columns [D, DQ*16) and V^T rows [D, DO*32)
  for (int i = tid; i < 2 * KV * KROW; i += NT) (&sK[0][0])[i] = (f16)0.f;
  for (int i = tid; i < 2 * DO * 32 * VROW; i += NT) (&sV[0][0])[i] = (f16)0.f;

  // Q fragments (B operand of S^T = K Q^T): lane (q = ql, hi) holds Q[q][16 kk + 8 hi .. +7]
  f16x8 qf[DQ];
  {
    const f16* qp = a.q + ((int64_t)n * T + (qvalid ? q : 0)) * a.ldq + h * D;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const int d0 = kk * 16 + hi * 8;
      U4H8 t;
      t.u = u32x4{0u, 0u, 0u, 0u};
      if (qvalid && d0 < D) t.u = *(const u32x4*)(qp + d0);
      qf[kk] = t.h;
    }
  }

  const int nts = (T + KV - 1) / KV;
  const int ntiles = nts * (ref >= 0 ? 2 : 1);

  u32x4 rk[NCH], rv[NCH];
  auto load_tile = [&](int t) {
    const bool second = t >= nts;
    const int tt = second ? t - nts : t;
    const f16* kb = second ? a.kref : a.k;
    const int64_t ldk = second ? a.ldkr : a.ldk;
    const f16* vb = second ? a.vtref : a.vt;
    const int64_t ldv = second ? a.ldvtr : a.ldvt;
    const int64_t tok0 = (int64_t)(second ? ref : n) * T;
    const bool vvec = second ? a.vtref_vec_ok : a.vt_vec_ok;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      rk[i] = u32x4{0u, 0u, 0u, 0u};
      rv[i] = u32x4{0u, 0u, 0u, 0u};
      if (c < KV * DC) {
        {  // K chunk: key = c / DC, d-chunk = c % DC
          const int key = c / DC, dc = c - key * DC;
          const int kg = tt * KV + key;
          if (kg < T) rk[i] = *(const u32x4*)(kb + (tok0 + kg) * ldk + h * D + dc * 8);
        }
        {  // V^T chunk: d row = c / 8, keys 8*(c%8) .. +7
          const int dr = c >> 3, kc = c & 7;
          const int kg = tt * KV + kc * 8;
          const f16* vp = vb + (int64_t)(h * D + dr) * ldv + tok0 + kg;
          if (vvec && kg + 8 <= T) {
            rv[i] = *(const u32x4*)vp;
          } else {
            U4H8 t8;
#pragma unroll
            for (int e = 0; e < 8; ++e) t8.e[e] = (kg + e < T) ? vp[e] : (f16)0.f;
            rv[i] = t8.u;
          }
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + i * NT;
      if (c < KV * DC) {
        const int key = c / DC, dc = c - key * DC;
        *(u32x4*)(&sK[buf][key * KROW + dc * 8]) = rk[i];
        // 16-key group permutation: [k0-3 | k8-11 | k4-7 | k12-15]
        const int dr = c >> 3, kc = c & 7;
        const int gb = (kc >> 1) * 16;
        const int plo = gb + ((kc & 1) ? 4 : 0), phi = gb + ((kc & 1) ? 12 : 8);
        *(u32x2*)(&sV[buf][dr * VROW + plo]) = u32x2{rv[i].x, rv[i].y};
        *(u32x2*)(&sV[buf][dr * VROW + phi]) = u32x2{rv[i].z, rv[i].w};
      }
    }
  };

  f32x16 o[DO];
#pragma unroll
  for (int dt = 0; dt < DO; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float c2 = a.scale_log2e;

  __syncthreads();  // padding zeros visible before the first tile is written
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const bool more = (t + 1) < ntiles;
    if (more) load_tile(t + 1);

    // ---- S^T = K Q^T : two 32-key x 32-query tiles -------------------------------------------
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < DQ; ++kk) {
      const f16x8 a0 = *(const f16x8*)(&sK[buf][ql * KROW + kk * 16 + hi * 8]);
      const f16x8 a1 = *(const f16x8*)(&sK[buf][(32 + ql) * KROW + kk * 16 + hi * 8]);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, qf[kk], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, qf[kk], s1, 0, 0, 0);
    }
    // mask keys beyond the segment length (last tile of a segment only)
    const int tt = t >= nts ? t - nts : t;
    if (tt * KV + KV > T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = tt * KV + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= T) s0[r] = -INFINITY;
        if (key + 32 >= T) s1[r] = -INFINITY;
      }
    }
    // ---- online softmax (one query per lane; partner lane^32 holds the other 32 keys) ------------
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f((m_run - m_new) * c2);
    const float mb = m_new * c2;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = exp2f(s0[r] * c2 - mb);
      s1[r] = exp2f(s1[r] * c2 - mb);
      psum += s0[r] + s1[r];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DO; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    // P^T fragments (B operand): k-slot (hi, j) of 16-key group g <-> accumulator reg 8*(g&1)+j of tile g>>1
    f16x8 pb[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pb[0][j] = (f16)s0[j];
      pb[1][j] = (f16)s0[8 + j];
      pb[2][j] = (f16)s1[j];
      pb[3][j] = (f16)s1[8 + j];
    }
    // ---- O^T += V^T P^T ------------------------------------------------------------------------
#pragma unroll
    for (int dt = 0; dt < DO; ++dt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f16x8 av = *(const f16x8*)(&sV[buf][(dt * 32 + ql) * VROW + g * 16 + hi * 8]);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, pb[g], o[dt], 0, 0, 0);
      }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (qvalid) {
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
}

template <int D>
int launch_ref_attn(const RefAttnArgs& a, int Nf, hipStream_t stream) {
  dim3 grid((unsigned)((a.T + 127) / 128), (unsigned)a.heads, (unsigned)Nf);
  AnipProfScope prof_(ANIP_K_REF_ATTN, (void*)stream);
  hipLaunchKernelGGL(ref_attn_kernel<D>, grid, dim3(NT), 0, stream, a);
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// temporal attention: one wave per (b, pixel, head); F <= 32 frames
// dynamic LDS per wave: q,k,v [F][d] fp16 + s [F][F] fp32
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void temporal_attn_kernel(const f16* __restrict__ qkv, f16* __restrict__ out, int B,
                                                          int F, int T, int heads, int d, float scale, int wpb,
                                                          int64_t nprob) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = heads * d, dc = d >> 3;
  const size_t per_wave = ((size_t)3 * F * d * 2 + (size_t)F * F * 4 + 15) & ~(size_t)15;
  const bool has_wave = wave < wpb;
  const int64_t prob = (int64_t)blockIdx.x * wpb + wave;
  const bool valid = has_wave && prob < nprob;
  char* base = dsm + (size_t)(has_wave ? wave : 0) * per_wave;
  f16* sq = (f16*)base;
  f16* sk = sq + F * d;
  f16* sv = sk + F * d;
  float* ss = (float*)(sv + F * d);
  // problem -> (b, t, h), head fastest so neighbouring waves read neighbouring columns
  const int hh = valid ? (int)(prob % heads) : 0;
  const int64_t bt = valid ? prob / heads : 0;
  const int t = (int)(bt % T);
  const int b = (int)(bt / T);

  if (valid) {
    for (int c = lane; c < F * dc; c += 64) {
      const int f = c / dc, ch = c - f * dc;
      const f16* row = qkv + (((int64_t)b * F + f) * T + t) * (3 * (int64_t)C) + hh * d + ch * 8;
      *(u32x4*)(sq + f * d + ch * 8) = *(const u32x4*)(row);
      *(u32x4*)(sk + f * d + ch * 8) = *(const u32x4*)(row + C);
      *(u32x4*)(sv + f * d + ch * 8) = *(const u32x4*)(row + 2 * C);
    }
  }
  __syncthreads();
  if (valid) {
    for (int e = lane; e < F * F; e += 64) {
      const int i = e / F, j = e - i * F;
      float acc = 0.f;
      for (int ch = 0; ch < dc; ++ch) {
        U4H8 qa, kb;
        qa.u = *(const u32x4*)(sq + i * d + ch * 8);
        kb.u = *(const u32x4*)(sk + j * d + ch * 8);
#pragma unroll
        for (int x = 0; x < 8; ++x) acc += (float)qa.e[x] * (float)kb.e[x];
      }
      ss[e] = acc * scale;
    }
  }
  __syncthreads();
  if (valid && lane < F) {
    float* r = ss + lane * F;
    float mx = r[0];
    for (int j = 1; j < F; ++j) mx = fmaxf(mx, r[j]);
    float sum = 0.f;
    for (int j = 0; j < F; ++j) {
      const float p = __expf(r[j] - mx);
      r[j] = p;
      sum += p;
    }
    const float inv = 1.0f / sum;
    for (int j = 0; j < F; ++j) r[j] *= inv;
  }
  __syncthreads();
  if (valid) {
    for (int oidx = lane; oidx < F * dc; oidx += 64) {
      const int i = oidx / dc, ch = oidx - i * dc;
      float acc[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) acc[x] = 0.f;
      for (int j = 0; j < F; ++j) {
        const float p = ss[i * F + j];
        U4H8 vv;
        vv.u = *(const u32x4*)(sv + j * d + ch * 8);
#pragma unroll
        for (int x = 0; x < 8; ++x) acc[x] += p * (float)vv.e[x];
      }
      U4H8 ov;
#pragma unroll
      for (int x = 0; x < 8; ++x) ov.e[x] = (f16)acc[x];
      *(u32x4*)(out + (((int64_t)b * F + i) * T + t) * (int64_t)C + hh * d + ch * 8) = ov.u;
    }
  }
}

}  // namespace

extern "C" int anip_ref_attention(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt,
                                  int64_t ldvt, const void* kref, int64_t ldkr, const void* vtref, int64_t ldvtr,
                                  const int* ref_index, void* out, int64_t ldo, int Nf, int T, int heads, int d,
                                  float scale, void* stream) {
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
  a.vtref = (const f16*)vtref; a.ldvtr = ldvtr;
  a.ref_index = ref_index;
  a.out = (f16*)out; a.ldo = ldo;
  a.T = T; a.heads = heads;
  a.scale_log2e = scale * 1.4426950408889634f;
  a.vt_vec_ok = ((T & 7) == 0) && ((ldvt & 7) == 0);
  a.vtref_vec_ok = ((T & 7) == 0) && ((ldvtr & 7) == 0);
  hipStream_t s = (hipStream_t)stream;
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

extern "C" int anip_temporal_attention(const void* qkv, void* out, int B, int F, int T, int heads, int d,
                                       float scale, void* stream) {
  ANIP_REQUIRE(qkv && out, "anip_temporal_attention: null pointer");
  ANIP_REQUIRE(B > 0 && T > 0 && heads > 0 && F > 0 && F <= 32, "anip_temporal_attention: need 1 <= F <= 32 (F=%d)", F);
  ANIP_REQUIRE((d & 7) == 0 && d > 0, "anip_temporal_attention: head dim %d must be a multiple of 8", d);
  const size_t per_wave = ((size_t)3 * F * d * 2 + (size_t)F * F * 4 + 15) & ~(size_t)15;
  int wpb = 4;
  while (wpb > 1 && per_wave * wpb > 65536) wpb >>= 1;
  ANIP_REQUIRE(per_wave * wpb <= 65536, "anip_temporal_attention: LDS budget exceeded (F=%d d=%d)", F, d);
  const int64_t nprob = (int64_t)B * T * heads;
  const int64_t blocks = cdiv64(nprob, wpb);
  {
    AnipProfScope prof_(ANIP_K_TEMPORAL_ATTN, (void*)stream);
    hipLaunchKernelGGL(temporal_attn_kernel, dim3((unsigned)blocks), dim3(NT), per_wave * wpb, (hipStream_t)stream,
                       (const f16*)qkv, (f16*)out, B, F, T, heads, d, scale, wpb, nprob);
  }
  ANIP_LAUNCH_CHECK("anip_temporal_attention");
  return 0;
}
