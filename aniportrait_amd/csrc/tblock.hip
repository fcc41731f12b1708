// Fused front half of a motion-module attention block at C = 320 (the 64x64 level of the denoising UNet), gfx950:
//
//     a = TemporalSelfAttention( LayerNorm(h) + pe[frame] )          (without the output projection)
//
// i.e. norms[i] -> pos_encoder -> to_q / to_k / to_v -> softmax(q k^T / sqrt d) v over the 16 frames of every pixel
// (src/models/motion_module.py:236-259 TemporalTransformerBlock.forward, :283-296 PositionalEncoding, :351-388
// VersatileAttention.forward) in ONE launch; the block's to_out + residual stays the GEMM it was.
//
// Why: the four launches LayerNorm(+PE) -> [to_q;to_k;to_v] GEMM -> temporal attention -> to_out moved
// 1.09 GB per block instance for 168 MB of input + output (the normalised rows and the M x 960 q|k|v matrix are written and
// read back), 300 us per instance, 250 instances per clip.  Here neither leaves the CU: 84 MB in, 84 MB out.
//
// Design — everything of a pixel lives in ONE wave's registers:
//   * a wave owns 2 pixels x 16 frames = two 16-row MFMA tiles; a 256-thread block = 4 waves = 8 adjacent pixels, two blocks
//     per CU (their prologues and tile loops de-phase);
//   * prologue: the 32 rows are loaded straight into the B-operand layout of v_mfma_f32_16x16x32_f16 (lane (frame, fq) holds
//     channels 32 ks + 8 fq .. + 7 of every 32-deep contraction step ks: 80 packed VGPRs), LayerNorm statistics are a lane-local
//     sum + two cross-lane adds, the normalised rows (+ beta + pe[frame], one fp32 table) are rounded to fp16 in place — the
//     arithmetic and the rounding point of the stand-alone layernorm_kernel (the row sums form in another order);
//   * the 960 x 320 projection matrix is host-packed per head PAIR [q 80 rows | k 80 | v 80] x 4 and streamed through a 5-stage
//     LDS ring (80 rows x 64 k = 10 KB per stage) by LDS-DMA under counted vmcnt, one s_barrier per stage; every 1-KB weight
//     fragment read feeds two MFMAs (the two pixels), so the fragment traffic is half the LDS read rate at MFMA peak;
//   * q and k come out of  D = W x^T  with lane (frame, 4 channels) — which IS the A / B operand layout of S^T = K Q^T when
//     two 16-channel tiles are paired per contraction step (the channel order inside a step is the same permutation on both
//     sides); v comes out of  D = x W^T  with lane (channel, 4 frames) — which IS the A operand layout of O^T = V^T P^T with
//     the 16 keys in k-slots {8 g + e, e < 4}; the probabilities leave the S^T accumulators in exactly the matching B layout.
//     No LDS staging, no cross-lane traffic between projection and attention; the only LDS round trip is the a tile on its way
//     out (wave-private, so that every row segment leaves in 16-B pieces);
//   * a head is 40 channels = 2.5 tiles: the middle tile of a pair (A 32-39 | B 0-7) enters each head's score MFMA masked by
//     lane group, and one P V MFMA serves both heads (P_A in k-slots e < 4 against V rows 0-7, P_B in e >= 4 against rows 8-15).
// Softmax: exponent base 2 on fp32 scores, probabilities rounded to fp16 (nearest), denominator = sum of the ROUNDED values.
//
// Measured on MI355X at the 64x64 level (M = 131072 rows; profiles/r05/a_*, h_pmc_sq_fused.json): 104.8 / 125.8 us warm / cache-cold
// against 252.5 / 266.0 for LayerNorm + qkv GEMM + temporal attention; 95 MB read + 97 MB written per launch; MFMA-busy 0.36 at an
// effective 2.15 GHz.  The bound is the weight stream: 614 KB of LDS-DMA per 128 rows = 0.63 GB per launch at ~6 TB/s (the chip's
// rate for weights every CU re-reads out of L2 — the fused FFN sits on the same 6.4 TB/s).  The same rows-in-registers scheme serves
// rowgemm320_kernel below (norm1 -> q | k | v^T; GroupNorm's affine form -> proj_in).
#include "common.h"

namespace {

#define TB_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define TB_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

constexpr int TB_C = 320;
constexpr int TB_NS = 5;                      // ring stages
constexpr int TB_STAGE = 80 * 128;            // 80 weight rows x 64 k
constexpr int TB_NSTAGES = 60;                // 4 head pairs x {q, k, v} x 5 K-tiles
constexpr int TB_ORS = 176;                   // bytes per row of the wave-private output staging tile (160 + 16)
constexpr int TB_OST = TB_NS * TB_STAGE;
// waves per workgroup: 4 = two workgroups per CU whose prologues and tile loops de-phase.  8 (one workgroup per CU, half the
// LDS-DMA instructions and weight bytes per wave; the kernels are templated on it) measured SLOWER in one call on MI355X:
// temporal front 111 -> 118 us, q|k|v^T projections 117 -> 124 us, GroupNorm + proj_in 74 -> 80 us (profiles/r05/c_fbench_nw*).
constexpr int TB_NW = 4;
constexpr int TB_LDS = TB_OST + TB_NW * 32 * TB_ORS;

struct TbArgs {
  const f16* x;        // [B*16*T][320]
  const float* gamma;  // [320]
  const float* bpe;    // [16][320]  beta + pe[frame]
  const f16* w;        // [960][320] packed per head pair: q 80 | k 80 | v 80
  f16* out;            // [B*16*T][320]
  int T, tpb;          // pixels per frame, blocks per batch sample (T / 8)
  float eps, scale_log2e;
};

// (a device function: written inside a lambda, the builtin makes the HOST pass drop the kernel stub silently — see attn_dma.hip)
template <int NSTAGES, int NW>
__device__ __forceinline__ void tb_issue_stage(const f16* w, char* smem, int sn, int wave, uint32_t lane_voff) {
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (uint32_t)((NSTAGES / 5) * 80 * TB_C * 2), 0x00020000);
  const int g = sn / 5, kt = sn - g * 5;
  const uint32_t soff = (uint32_t)(g * (80 * TB_C * 2) + kt * 128);
  char* dst = smem + (sn % TB_NS) * TB_STAGE;
#pragma unroll
  for (int i = 0; i < (10 + NW - 1) / NW; ++i) {
    const int j = wave + NW * i;               // 1-KB piece: weight rows 8 j .. 8 j + 7 of the group
    if (j < 10) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, TB_LDS_PTR(dst + j * 1024), 16, lane_voff + (uint32_t)(j * 8 * TB_C * 2), soff, 0, 0);
  }
}

union TbP4 {
  u32x2 u;
  f16 e[4];
};
__device__ __forceinline__ u32x2 tb_pack4(const f32x4 v) {
  TbP4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t.e[r] = (f16)v[r];
  return t.u;
}
__device__ __forceinline__ f16x8 tb_pack8(const u32x2 lo, const u32x2 hi) {
  U4H8 t;
  t.u = u32x4{lo.x, lo.y, hi.x, hi.y};
  return t.h;
}

// one tile group: acc[j][px] (16 x 16 tiles j = 0..4 of the group's 80 weight rows, pixel px) over the five 64-deep K-tiles
//   SWAP = false:  D = W x^T   lane (frame = lane & 15, channels 4 (lane >> 4) + r)       (q, k)
//   SWAP = true:   D = x W^T   lane (channel = lane & 15, frames 4 (lane >> 4) + r)       (v)
template <bool SWAP, int NSTAGES, int NW>
__device__ __forceinline__ void tb_group(const f16* w, char* smem, int& s, const int wave, const uint32_t lane_voff, const int fr,
                                         const int fq, const f16x8 (&xf)[2][10], f32x4 (&acc)[5][2]) {
#pragma unroll
  for (int j = 0; j < 5; ++j) acc[j][0] = acc[j][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 5; ++kt) {
    // this wave's pieces of stage s have landed: everything but the TB_NS - 2 stages issued after it is complete
    if (s >= NSTAGES - (TB_NS - 2)) TB_VMCNT(0);
    else if (wave < 10 % NW) TB_VMCNT(((10 + NW - 1) / NW) * (TB_NS - 2));     // waves that issue one piece more per stage
    else TB_VMCNT((10 / NW) * (TB_NS - 2));
    __builtin_amdgcn_s_barrier();            // stage s complete for all waves; the slot read at s - 1 is free
    asm volatile("" ::: "memory");
    if (s + TB_NS - 1 < NSTAGES) tb_issue_stage<NSTAGES, NW>(w, smem, s + TB_NS - 1, wave, lane_voff);
    const char* st = smem + (s % TB_NS) * TB_STAGE;
    // all ten fragment reads of the stage go out before its first MFMA (left to the scheduler they are issued in pairs right in
    // front of their consumers, every pair's LDS latency exposed)
    f16x8 wf[2][5];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ((ks * 4 + fq) ^ (fr & 7)) << 4;
#pragma unroll
      for (int j = 0; j < 5; ++j) wf[ks][j] = *(const f16x8*)(st + (j * 16 + fr) * 128 + ko);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          if constexpr (SWAP) acc[j][px] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[px][kt * 2 + ks], wf[ks][j], acc[j][px], 0, 0, 0);
          else acc[j][px] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[px][kt * 2 + ks], acc[j][px], 0, 0, 0);
        }
    __builtin_amdgcn_sched_barrier(0);
    ++s;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 8 / NW) void temporal_qkv_attn_kernel(const TbArgs a) {
  constexpr int C = TB_C;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.x / a.tpb;
  const int t0 = (blockIdx.x - b * a.tpb) * (2 * NW) + wave * 2;

  // ---- weight ring: the first TB_NS - 1 stages go out before anything else ------------------------------------------------
  const uint32_t lane_voff = (uint32_t)((lane >> 3) * C * 2 + (((lane & 7) ^ (lane >> 3)) << 4));
#pragma unroll
  for (int sn = 0; sn < TB_NS - 1; ++sn) tb_issue_stage<TB_NSTAGES, NW>(a.w, smem, sn, wave, lane_voff);

  // ---- the wave's 2 x 16 rows in B-operand layout, LayerNorm + pe in registers ------------------------------------------------
  f16x8 xf[2][10];
  {
    const f16* xp = a.x + ((int64_t)(b * 16 + fr) * a.T + t0) * C + fq * 8;
    U4H8 xr[2][10];
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) xr[px][ks].u = *(const u32x4*)(xp + px * C + ks * 32);
    float mean[2], rstd[2];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) sm += (float)xr[px][ks].e[e];
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      mean[px] = sm / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = (float)xr[px][ks].e[e] - mean[px];
          q += d * d;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rstd[px] = rsqrtf(q / (float)C + a.eps);
    }
    const float* gp = a.gamma + fq * 8;
    const float* bp = a.bpe + fr * C + fq * 8;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
      const float4 g0 = *(const float4*)(gp + ks * 32), g1 = *(const float4*)(gp + ks * 32 + 4);
      const float4 b0 = *(const float4*)(bp + ks * 32), b1 = *(const float4*)(bp + ks * 32 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        U4H8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (f16)(((float)xr[px][ks].e[e] - mean[px]) * rstd[px] * gg[e] + bb[e]);
        xf[px][ks] = o.h;
      }
    }
  }

  char* ost = smem + TB_OST + wave * (32 * TB_ORS);
  const u32x2 z2 = u32x2{0u, 0u};
  int s = 0;
#pragma unroll 1
  for (int hp = 0; hp < 4; ++hp) {
    f32x4 acc[5][2];
    u32x2 qp[2][5], kp[2][5];
    tb_group<false, TB_NSTAGES, NW>(a.w, smem, s, wave, lane_voff, fr, fq, xf, acc);
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int j = 0; j < 5; ++j) qp[px][j] = tb_pack4(acc[j][px]);
    tb_group<false, TB_NSTAGES, NW>(a.w, smem, s, wave, lane_voff, fr, fq, xf, acc);
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int j = 0; j < 5; ++j) kp[px][j] = tb_pack4(acc[j][px]);

    // ---- S^T = K Q^T per (pixel, head), softmax over the 16 keys of query `fr` (4 registers x 4 lane groups) ----------------
    u32x2 pop[2][2];     // P^T of (pixel, head): lane (query fr, keys 4 fq .. + 3) as fp16
    float inv[2][2];
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const f16x8 qmid = tb_pack8(qp[px][2], z2);
#pragma unroll
      for (int hd = 0; hd < 2; ++hd) {
        const bool mine = hd == 0 ? (fq < 2) : (fq >= 2);          // the middle tile's lanes of this head
        const f16x8 k0 = hd == 0 ? tb_pack8(kp[px][0], kp[px][1]) : tb_pack8(kp[px][3], kp[px][4]);
        const f16x8 q0 = hd == 0 ? tb_pack8(qp[px][0], qp[px][1]) : tb_pack8(qp[px][3], qp[px][4]);
        const f16x8 k1 = tb_pack8(mine ? kp[px][2] : z2, z2);
        f32x4 sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, q0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        sT = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qmid, sT, 0, 0, 0);
        float mx = fmaxf(fmaxf(sT[0], sT[1]), fmaxf(sT[2], sT[3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        TbP4 p;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p.e[r] = (f16)__builtin_amdgcn_exp2f((sT[r] - mx) * a.scale_log2e);
          sum += (float)p.e[r];
        }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        inv[px][hd] = 1.0f / sum;
        pop[px][hd] = p.u;
      }
    }

    tb_group<true, TB_NSTAGES, NW>(a.w, smem, s, wave, lane_voff, fr, fq, xf, acc);
    // ---- O^T = V^T P^T: lane (query fr, channels 16 j + 4 fq + r of the pair), scaled, into the staging tile -------------------
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const f16x8 pA = tb_pack8(pop[px][0], z2), pB = tb_pack8(pop[px][1], z2), pAB = tb_pack8(pop[px][0], pop[px][1]);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const u32x2 v = tb_pack4(acc[j][px]);
        f16x8 va = tb_pack8(v, z2);
        if (j == 2) va = fr < 8 ? tb_pack8(v, z2) : tb_pack8(z2, v);
        const f16x8 pb = j < 2 ? pA : (j == 2 ? pAB : pB);
        const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const float sc = j < 2 ? inv[px][0] : (j == 2 ? (fq < 2 ? inv[px][0] : inv[px][1]) : inv[px][1]);
        TbP4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(o[r] * sc);
        *(u32x2*)(ost + (px * 16 + fr) * TB_ORS + j * 32 + fq * 8) = ov.u;
      }
    }
    // ---- the wave's 32 x 80 tile leaves in 16-B pieces (10 per row) ------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int id = i * 64 + lane;
      const int row = id / 10, pc = id - row * 10;
      const u32x4 v = *(const u32x4*)(ost + row * TB_ORS + pc * 16);
      f16* op = a.out + ((int64_t)(b * 16 + (row & 15)) * a.T + t0 + (row >> 4)) * C + hp * 80 + pc * 8;
      *(u32x4*)op = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Row-stationary projections at C = 320 (round 5): the same register-resident rows, the same weight ring, no attention —
//   MODE 0  q | k | v^T = LayerNorm(h) [to_q; to_k; to_v]^T  of a spatial transformer block's attn1 in the three layouts the
//           reference-attention kernel reads (q token-major x alpha, k head-major, v transposed): one launch instead of
//           layernorm + three GEMMs (src/models/attention.py:383-401 norm1 -> attn1 projections under
//           src/models/mutual_self_attention.py:147-186)
//   MODE 1  out = (x * scale[frame] + shift[frame]) W^T + bias: GroupNorm's affine apply (statistics from gn_stats, finalised
//           into a per-(frame, channel) table) inside the proj_in 1x1 convolution of Transformer3DModel / the temporal
//           transformer (src/models/transformer_3d.py:128-139, src/models/motion_module.py:185-204): no gn_apply pass
// A wave owns 32 consecutive rows, a 256-thread block 128; weights [NG * 80][320] in the caller's row order.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int RG_VRS = 80;                     // bytes per channel row of the transposed staging tile (64 + 16)
constexpr int RG_STG = 80 * RG_VRS;            // wave-private staging: max(32 rows x 176 B, 80 channels x 80 B)
constexpr int RG_LDS = TB_OST + TB_NW * RG_STG;

struct RgArgs {
  const f16* x;            // [M][320]
  const float* gamma;      // MODE 0: LayerNorm weight / bias [320]
  const float* beta;
  const float* sst;        // MODE 1: [frames][320][2] (scale, shift)
  int rows_per_frame;      // MODE 1
  const f16* w;            // [NG * 80][320]
  const float* bias;       // MODE 1: [320] or null
  f16* out0;               // MODE 0: q [M][320];  MODE 1: out [M][320]
  f16* out1;               // MODE 0: k head-major [8][M][40]
  f16* out2;               // MODE 0: v^T [320][ldvt]
  int64_t ldvt;
  int M;
  float eps, alpha;
};

template <int MODE, int NW>
__global__ __launch_bounds__(NW * 64, 8 / NW) void rowgemm320_kernel(const RgArgs a) {
  constexpr int C = TB_C;
  constexpr int NG = MODE == 0 ? 12 : 4;
  constexpr int NSTG = NG * 5;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int m0 = blockIdx.x * (32 * NW) + wave * 32;   // the wave's rows m0 .. m0 + 31 (M % (32 * NW) == 0)

  const uint32_t lane_voff = (uint32_t)((lane >> 3) * C * 2 + (((lane & 7) ^ (lane >> 3)) << 4));
#pragma unroll
  for (int sn = 0; sn < TB_NS - 1; ++sn) tb_issue_stage<NSTG, NW>(a.w, smem, sn, wave, lane_voff);

  f16x8 xf[2][10];
  {
    const f16* xp = a.x + (int64_t)(m0 + fr) * C + fq * 8;
    U4H8 xr[2][10];
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) xr[px][ks].u = *(const u32x4*)(xp + px * 16 * C + ks * 32);
    if constexpr (MODE == 0) {
      float mean[2], rstd[2];
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) sm += (float)xr[px][ks].e[e];
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        mean[px] = sm / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = (float)xr[px][ks].e[e] - mean[px];
            q += d * d;
          }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        rstd[px] = rsqrtf(q / (float)C + a.eps);
      }
      const float* gp = a.gamma + fq * 8;
      const float* bp = a.beta + fq * 8;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        const float4 g0 = *(const float4*)(gp + ks * 32), g1 = *(const float4*)(gp + ks * 32 + 4);
        const float4 b0 = *(const float4*)(bp + ks * 32), b1 = *(const float4*)(bp + ks * 32 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          U4H8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.e[e] = (f16)(((float)xr[px][ks].e[e] - mean[px]) * rstd[px] * gg[e] + bb[e]);
          xf[px][ks] = o.h;
        }
      }
    } else {
      // (scale, shift) of the wave's frame (32 | rows_per_frame): same arithmetic as gn_apply_kernel, x * scale + shift
      const float* sp = a.sst + ((int64_t)(m0 / a.rows_per_frame) * C + fq * 8) * 2;
#pragma unroll
      for (int ks = 0; ks < 10; ++ks) {
        float4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = *(const float4*)(sp + ks * 64 + i * 4);
        const float sc[8] = {t[0].x, t[0].z, t[1].x, t[1].z, t[2].x, t[2].z, t[3].x, t[3].z};
        const float sh[8] = {t[0].y, t[0].w, t[1].y, t[1].w, t[2].y, t[2].w, t[3].y, t[3].w};
#pragma unroll
        for (int px = 0; px < 2; ++px) {
          U4H8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o.e[e] = (f16)((float)xr[px][ks].e[e] * sc[e] + sh[e]);
          xf[px][ks] = o.h;
        }
      }
    }
  }

  char* ost = smem + TB_OST + wave * RG_STG;
  int s = 0;
#pragma unroll 1
  for (int g = 0; g < NG; ++g) {
    f32x4 acc[5][2];
    const bool trans = MODE == 0 && g >= 8;
    if (!trans) {
      tb_group<false, NSTG, NW>(a.w, smem, s, wave, lane_voff, fr, fq, xf, acc);
      // lane (row fr of pixel group px, columns 16 j + 4 fq + r of the group) -> staging [32 rows][176 B]
      float bias4[5][4];
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[j][r] = (MODE == 1 && a.bias != nullptr) ? a.bias[g * 80 + j * 16 + fq * 4 + r] : 0.f;
      const float al = (MODE == 0 && g < 4) ? a.alpha : 1.0f;
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          TbP4 ov;
#pragma unroll
          for (int r = 0; r < 4; ++r) ov.e[r] = (f16)(acc[j][px][r] * al + bias4[j][r]);
          *(u32x2*)(ost + (px * 16 + fr) * TB_ORS + j * 32 + fq * 8) = ov.u;
        }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int id = i * 64 + lane;
        const int row = id / 10, pc = id - row * 10;
        const u32x4 v = *(const u32x4*)(ost + row * TB_ORS + pc * 16);
        f16* op;
        if (MODE == 0 && g >= 4) {           // k head-major: head 2 (g - 4) + pc / 5, 40 channels = five 16-B pieces per token
          const int head = 2 * (g - 4) + pc / 5;
          op = a.out1 + ((int64_t)head * a.M + m0 + row) * 40 + (pc % 5) * 8;
        } else {
          op = a.out0 + (int64_t)(m0 + row) * C + g * 80 + pc * 8;
        }
        *(u32x4*)op = v;
      }
    } else {
      tb_group<true, NSTG, NW>(a.w, smem, s, wave, lane_voff, fr, fq, xf, acc);
      // lane (channel 16 j + fr of the group, rows 16 px + 4 fq + r) -> staging [80 channels][80 B], 64 B = the wave's 32 rows
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int j = 0; j < 5; ++j) *(u32x2*)(ost + (j * 16 + fr) * RG_VRS + px * 32 + fq * 8) = tb_pack4(acc[j][px]);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int id = i * 64 + lane;
        const int ch = id >> 2, pc = id & 3;
        const u32x4 v = *(const u32x4*)(ost + ch * RG_VRS + pc * 16);
        *(u32x4*)(a.out2 + (int64_t)((g - 8) * 80 + ch) * a.ldvt + m0 + pc * 8) = v;
      }
    }
  }
}

}  // namespace

// LayerNorm(+pe) -> [to_q; to_k; to_v] -> temporal self-attention over F = 16 frames at C = 320 (8 heads x 40), one launch.
extern "C" int anip_temporal_qkv_attention(const void* x, const float* gamma, const float* beta_pe, const void* w_packed,
                                           void* out, int B, int F, int T, int C, int heads, float eps, float scale,
                                           void* stream) {
  ANIP_REQUIRE(x && gamma && beta_pe && w_packed && out, "anip_temporal_qkv_attention: null pointer");
  ANIP_REQUIRE(anip_temporal_qkv_attention_supported(F, T, C, heads) == 1,
               "anip_temporal_qkv_attention: only F = 16, C = 320, heads = 8, T %% 8 == 0 is built (F=%d T=%d C=%d heads=%d)", F, T, C,
               heads);
  ANIP_REQUIRE(B > 0 && (int64_t)B * (T / 8) < (1ll << 31), "anip_temporal_qkv_attention: bad B=%d", B);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta_pe | (uintptr_t)w_packed | (uintptr_t)out) & 15) == 0,
               "anip_temporal_qkv_attention: pointers must be 16-B aligned");
  TbArgs a;
  a.x = (const f16*)x; a.gamma = gamma; a.bpe = beta_pe; a.w = (const f16*)w_packed; a.out = (f16*)out;
  a.T = T; a.tpb = T / (2 * TB_NW); a.eps = eps; a.scale_log2e = scale * 1.4426950408889634f;
  if (anip_raise_lds_limit((const void*)temporal_qkv_attn_kernel<TB_NW>, TB_LDS) != 0) {
    anip_set_error("anip_temporal_qkv_attention: cannot raise the dynamic LDS limit to %d bytes", TB_LDS);
    return -2;
  }
  {
    AnipProfScope prof_(ANIP_K_GEMM, stream);
    hipLaunchKernelGGL(temporal_qkv_attn_kernel<TB_NW>, dim3((unsigned)(B * a.tpb)), dim3(TB_NW * 64), TB_LDS, (hipStream_t)stream, a);
  }
  ANIP_LAUNCH_CHECK("anip_temporal_qkv_attention");
  return 0;
}

extern "C" int anip_temporal_qkv_attention_supported(int F, int T, int C, int heads) {
  return (F == 16 && C == TB_C && heads == 8 && T > 0 && (T % 8) == 0) ? 1 : 0;
}

extern "C" int anip_rowgemm320_supported(int64_t M, int C, int64_t rows_per_frame) {
  return (C == TB_C && M > 0 && (M % (32 * TB_NW)) == 0 && M * (int64_t)TB_C * 2 < (1ll << 40) && (rows_per_frame == 0 || rows_per_frame % 32 == 0))
             ? 1 : 0;
}

// LayerNorm(x) -> q (token-major, x alpha) | k (head-major [heads][M][40]) | v^T ([320][ldvt]) at C = 320, 8 heads, one launch
extern "C" int anip_ln_qkv_projection(const void* x, const float* gamma, const float* beta, float eps, const void* w_qkv,
                                      void* q, float q_alpha, void* k_head_major, void* vt, int64_t ldvt, int64_t M, int C,
                                      int heads, void* stream) {
  ANIP_REQUIRE(x && gamma && beta && w_qkv && q && k_head_major && vt, "anip_ln_qkv_projection: null pointer");
  ANIP_REQUIRE(heads == 8 && anip_rowgemm320_supported(M, C, 0) == 1 && M < (1ll << 31) && ldvt >= M && (ldvt & 7) == 0,
               "anip_ln_qkv_projection: only C = 320, heads = 8, M %% 128 == 0, ldvt %% 8 == 0 is built (M=%lld C=%d heads=%d)",
               (long long)M, C, heads);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w_qkv | (uintptr_t)q | (uintptr_t)k_head_major |
                 (uintptr_t)vt) & 15) == 0, "anip_ln_qkv_projection: pointers must be 16-B aligned");
  RgArgs a = {};
  a.x = (const f16*)x; a.gamma = gamma; a.beta = beta; a.w = (const f16*)w_qkv;
  a.out0 = (f16*)q; a.out1 = (f16*)k_head_major; a.out2 = (f16*)vt; a.ldvt = ldvt; a.M = (int)M; a.eps = eps; a.alpha = q_alpha;
  if (anip_raise_lds_limit((const void*)rowgemm320_kernel<0, TB_NW>, RG_LDS) != 0) {
    anip_set_error("anip_ln_qkv_projection: cannot raise the dynamic LDS limit to %d bytes", RG_LDS);
    return -2;
  }
  {
    AnipProfScope prof_(ANIP_K_GEMM, stream);
    hipLaunchKernelGGL((rowgemm320_kernel<0, TB_NW>), dim3((unsigned)(M / (32 * TB_NW))), dim3(TB_NW * 64), RG_LDS, (hipStream_t)stream, a);
  }
  ANIP_LAUNCH_CHECK("anip_ln_qkv_projection");
  return 0;
}

// out = (x * scale[frame] + shift[frame]) W^T + bias at C = 320 -> 320: GroupNorm's apply inside the following 1x1 convolution
extern "C" int anip_affine_linear320(const void* x, const float* scale_shift, int64_t rows_per_frame, const void* w,
                                     const float* bias, void* out, int64_t M, int C, void* stream) {
  ANIP_REQUIRE(x && scale_shift && w && out, "anip_affine_linear320: null pointer");
  ANIP_REQUIRE(rows_per_frame > 0 && anip_rowgemm320_supported(M, C, rows_per_frame) == 1 && M % rows_per_frame == 0 && M < (1ll << 31),
               "anip_affine_linear320: only C = 320, M %% 128 == 0, rows_per_frame %% 32 == 0 is built (M=%lld C=%d rows_per_frame=%lld)",
               (long long)M, C, (long long)rows_per_frame);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)scale_shift | (uintptr_t)w | (uintptr_t)out) & 15) == 0 && (((uintptr_t)bias) & 3) == 0,
               "anip_affine_linear320: pointers must be 16-B aligned");
  RgArgs a = {};
  a.x = (const f16*)x; a.sst = scale_shift; a.rows_per_frame = (int)rows_per_frame; a.w = (const f16*)w; a.bias = bias;
  a.out0 = (f16*)out; a.M = (int)M; a.alpha = 1.0f;
  if (anip_raise_lds_limit((const void*)rowgemm320_kernel<1, TB_NW>, RG_LDS) != 0) {
    anip_set_error("anip_affine_linear320: cannot raise the dynamic LDS limit to %d bytes", RG_LDS);
    return -2;
  }
  {
    AnipProfScope prof_(ANIP_K_GEMM, stream);
    hipLaunchKernelGGL((rowgemm320_kernel<1, TB_NW>), dim3((unsigned)(M / (32 * TB_NW))), dim3(TB_NW * 64), RG_LDS, (hipStream_t)stream, a);
  }
  ANIP_LAUNCH_CHECK("anip_affine_linear320");
  return 0;
}
