// Fused GEGLU feed-forward for gfx950 (CDNA4):  out = residual + b2 + W2 · ( (x W1v^T + b1v) * gelu(x W1g^T + b1g) )
// — diffusers FeedForward(activation_fn="geglu") as used by every (Temporal)BasicTransformerBlock and motion module
// (src/models/attention.py:361, src/models/motion_module.py:233) plus the block's residual add.
//
// Status (round 1): parity-green on MI355X (tests/test_hip_ops.py::test_ffn_geglu_fused at M = 128 / 4096 / 5000; bit-identical to
// the two-GEMM path at M = 131072) and 435 us vs 644 us for the two launches in isolation (tools/exp_ffn.py),
// but the single end-to-end bench run that still fit in the round's GPU budget was slower with it (1.70 vs 1.62 s per
// clip), so the engine only uses it with ANIP_FUSED_FFN=1.  Known weakness: every 64-deep K-tile iteration drains the
// DMA queue (vmcnt(0) + barrier) with only 16 MFMAs per wave to cover it.
//
// Round 2: enabled by default (engine._FUSED_FFN) — with every DDIM step inside the captured graph it is 1-4 % faster end
// to end than the two-GEMM path.  A variant with a continuous 3-stage W1 ring under counted vmcnt and the W2 chunk
// loaded straight into registers (no per-K-tile drain) was built and measured: 516 us vs 440 us for this kernel
// (profiles/r02: exp_ffn) — the G phase is not latency-bound, so the variant was dropped.
//
// Round 5: (a) the block's LayerNorm moved into the prologue (ffn_geglu_kernel<C, LN = true>, anip_ffn_geglu_ln: no layernorm
// launch, one input tensor): 423 -> 387 us per layer, 250 layers per clip.  (b) The W1 ring of DESIGN 5.3 was built — 32-deep
// tiles through five 8-KB slots, three tiles ahead of the compute under counted vmcnt, the H tile in the two slots of a chunk's
// last tiles, W2 waited for by queue order alone: x 80 + ring 40 + W2 40 = 160 KB — parity-green, 166 instead of 246 VGPRs, and
// NO faster: 385 / 401 us (warm / cold) against 389 / 424 for this two-stage form, whole clip 1288.7 | 1273.7 | 1276.3 ms
// ring | stages | ring inside one call (profiles/r05/e_*).  The kernel moves 2.46 MB of weights through LDS-DMA per 128 rows =
// 2.5 GB per launch = 6.4 TB/s: that is the chip's LDS-DMA rate for weights every CU re-reads out of L2 (MI355X_MICROARCH.md,
// "ldsdma-fill"), not the latency of a transfer — which confirms round 2's result ("the G phase is not latency-bound") and
// retires the idea; the ring was removed again.  Fewer weight bytes per row need a taller row block, and 256 rows of x are
// 160 KB by themselves.
//
// Why: at C = 320 (the 64x64 level) the two GEMMs cost 332 + 194 us per layer and are bound by memory traffic, not by
// the matrix pipe: the GEGLU output H (M x 4C fp16 = 335 MB) is written to HBM and read back, and the second GEMM
// streams it as its A operand.  Here H never leaves the CU.
//
// One 512-thread block (8 waves as 2(M) x 4(N)) owns 128 rows:
//   * the 128 x 320 x-tile stays resident in LDS for the whole block (5 planes of [128 rows][128 B], 80 KB);
//   * the hidden dimension is walked in chunks of 64 units = 128 rows of the packed W1 ([16 value | 16 gate] per
//     32 rows, hipops.pack_geglu): G[128 x 128] = x · W1c^T is accumulated over five 64-deep K-tiles of W1c streamed
//     through two 16 KB LDS stages by LDS-DMA; value and gate of a hidden unit sit in the same lane, so GEGLU is applied
//     in registers and the 128 x 64 fp16 tile H goes to LDS (into the W1 stage that has just been retired);
//   * out[128 x 320] += H · W2c^T with W2c = the chunk's 64 columns of W2 (320 rows x 128 B = 40 KB, one LDS-DMA
//     burst per chunk issued while G is being computed); 80 accumulator VGPRs per wave (64 x 80 wave tile);
//   * epilogue: + b2 + residual, 16-B stores (v_permlane16_swap pairs of tiles, as in gemm2).
// LDS: 80 + 32 + 40 = 152 KB -> one block per CU.  Global->LDS traffic per block: 80 KB (x) + 20 x 120 KB (W1, W2).
#include "common.h"

// GEGLU activation of the epilogues: gelu_poly_f (FMA pipe only; default) or, with -DANIP_GELU_EXACT, gelu_fast_f (A&S erf,
// |error| <= 1.5e-7, two transcendentals)
#ifdef ANIP_GELU_EXACT
#define ANIP_GELU gelu_fast_f
#else
#define ANIP_GELU gelu_poly_f
#endif

namespace {

constexpr uint32_t FFN_OOB = 0xFFFFFFF0u;
#define FFN_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void ffn_row_swap(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

struct FfnArgs {
  const f16* x;      // [M][C]
  const f16* w1p;    // [8C][C]   packed GEGLU projection
  const float* b1p;  // [8C]      packed the same way
  const f16* w2;     // [C][4C]
  const float* b2;   // [C] or null
  const f16* res;    // [M][C] or null
  f16* out;          // [M][C]
  int M;
  const float* gamma;  // LN variant: x holds the RAW rows, LayerNorm(gamma, beta, eps) is applied while the x tile is staged
  const float* beta;
  float eps;
};

template <int C_, bool LN>
__global__ __launch_bounds__(512, 2) void ffn_geglu_kernel(const FfnArgs a) {
  constexpr int BM = 128;
  constexpr int KT = C_ / 64;            // 64-deep K-tiles of x / W1
  constexpr int H4 = 4 * C_;             // hidden units
  constexpr int NCHUNK = H4 / 64;        // chunks of 64 hidden units
  constexpr int PLANE = BM * 128;        // bytes of one [128][128 B] LDS image
  constexpr int XS = KT * PLANE;         // x planes
  constexpr int W1S = XS;                // two W1 stages
  constexpr int W2S = W1S + 2 * PLANE;   // W2 chunk: [C_][128 B]
  constexpr int FN = C_ / 4 / 16;        // output 16-col tiles per wave (wave tile 64 x C_/4)
  static_assert(C_ % 64 == 0 && (C_ / 4) % 16 == 0, "unsupported width");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * BM;

  const uint32_t x_bytes = (uint32_t)min((int64_t)0xFFFFF000ll, (int64_t)a.M * C_ * 2);
  auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
  auto rW1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1p, 0, (uint32_t)(2 * H4 * C_ * 2), 0x00020000);
  auto rW2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, 0, (uint32_t)(C_ * H4 * 2), 0x00020000);

  // LDS-DMA: one instruction = 8 rows x 128 B, lane -> (row lr, 16-B slot ls); the bank swizzle (chunk ^= row & 7) is
  // applied on the global source address, the LDS image is lane-linear
  const int lr = lane >> 3, ls = lane & 7;
  const int g16 = (ls ^ lr) * 16;                              // byte offset of this lane's source chunk inside the 128-B row
  auto issue_x = [&](int kt) {                                 // x plane kt: 16 instructions, 2 per wave
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 8 + lr;
      const int m = m0 + row;
      const uint32_t vo = m < a.M ? (uint32_t)((m * C_ + kt * 64) * 2 + g16) : FFN_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, FFN_LDS_PTR(smem + kt * PLANE + (wave * 2 + i) * 1024), 16, vo, 0, 0, 0);
    }
  };
  auto issue_w1 = [&](int chunk, int kt, int stage) {          // 128 packed rows x 64 k: 16 instructions, 2 per wave
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (wave * 2 + i) * 8 + lr;
      const uint32_t vo = (uint32_t)(((chunk * 128 + row) * C_ + kt * 64) * 2 + g16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW1, FFN_LDS_PTR(smem + W1S + stage * PLANE + (wave * 2 + i) * 1024), 16, vo, 0,
                                               0, 0);
    }
  };
  auto issue_w2 = [&](int chunk) {                             // C_ rows x 64 k of W2: C_/8 instructions, C_/64 per wave
#pragma unroll
    for (int i = 0; i < C_ / 64; ++i) {
      const int row = (wave * (C_ / 64) + i) * 8 + lr;
      const uint32_t vo = (uint32_t)((row * H4 + chunk * 64) * 2 + g16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW2, FFN_LDS_PTR(smem + W2S + (wave * (C_ / 64) + i) * 1024), 16, vo, 0, 0, 0);
    }
  };

  // fragment addressing: lane (fr = row in the 16-row tile, fq = 8-wide k group); k32 step ks -> chunk ks*4 + fq
  const int fr = lane & 15, fq = lane >> 4;
  const int ko0 = ((0 * 4 + fq) ^ (fr & 7)) << 4, ko1 = ((1 * 4 + fq) ^ (fr & 7)) << 4;
  const int a_row = (wm * 64 + fr) * 128;                      // x / H rows of this wave (+ i * 16 * 128)
  const int b1_row = (wn * 32 + fr) * 128;                     // W1 stage rows: [16 value | 16 gate] of this wave
  // output column of this wave's j-th 16-column tile: FN is odd (5 tiles = 80 columns = 160 B), so contiguous ranges
  // would leave waves 1 and 3 with every 64-B row segment of the epilogue 32 B off the access granule (see gemm2.hip):
  // four tiles from a 128-B aligned range, the fifth from the tail
  auto tile_c = [&](int j) -> int {
    if ((FN & 1) == 0) return wn * (C_ / 4) + j * 16;
    return j < FN - 1 ? wn * (FN - 1) * 16 + j * 16 : 4 * (FN - 1) * 16 + wn * 16;
  };

  f32x4 oacc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) oacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (LN) {
    // LayerNorm in the prologue (round 5): four lanes per row (16-B chunks part, part + 4, ..), statistics from registers, the
    // normalised row written as fp16 into the same swizzled planes the LDS-DMA path fills — the stand-alone layernorm launch
    // (84 MB read + 84 MB written per 64x64 layer) and the second input tensor are gone.  W1's first tile streams meanwhile.
    issue_w1(0, 0, 0);
    constexpr int NCH = C_ / 8 / 4;            // chunks per lane
    const int row = tid >> 2, part = tid & 3;
    const int m = m0 + row;
    const f16* xr = a.x + (int64_t)(m < a.M ? m : 0) * C_;
    U4H8 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i].u = *(const u32x4*)(xr + (part + 4 * i) * 8);
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) sm += (float)v[i].e[e];
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    const float mean = sm / (float)C_;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[i].e[e] - mean;
        q += d * d;
      }
    q += __shfl_xor(q, 1, 64);
    q += __shfl_xor(q, 2, 64);
    const float rstd = rsqrtf(q / (float)C_ + a.eps);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = part + 4 * i;              // 16-B chunk of the row: plane c / 8, slot (c % 8) ^ (row & 7)
      const float4 g0 = *(const float4*)(a.gamma + c * 8), g1 = *(const float4*)(a.gamma + c * 8 + 4);
      const float4 b0 = *(const float4*)(a.beta + c * 8), b1 = *(const float4*)(a.beta + c * 8 + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      U4H8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = (f16)(((float)v[i].e[e] - mean) * rstd * gg[e] + bb[e]);
      if (m >= a.M) o.u = u32x4{0u, 0u, 0u, 0u};
      *(u32x4*)(smem + (c >> 3) * PLANE + row * 128 + (((c & 7) ^ (row & 7)) << 4)) = o.u;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) issue_x(kt);
    issue_w1(0, 0, 0);
  }

  for (int c = 0; c < NCHUNK; ++c) {
    const int base = c & 1;                                    // K-tile kt of this chunk lives in stage (kt + base) & 1
    // GEGLU bias of this wave's 16 hidden units (value at packed row r, gate at r + 16)
    float bv[4], bg[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bv[r] = a.b1p[c * 128 + wn * 32 + fq * 4 + r];
      bg[r] = a.b1p[c * 128 + wn * 32 + 16 + fq * 4 + r];
    }
    f32x4 gacc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) gacc[i][0] = gacc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < KT; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // W1(c, kt) (and x, W2 pieces) landed for all waves; previous reads are complete
      if (kt == 0) issue_w2(c);       // W2 buffer: last read in phase O of chunk c-1, finished before this barrier
      if (kt + 1 < KT) issue_w1(c, kt + 1, (kt + 1 + base) & 1);
      const char* xs = smem + kt * PLANE;
      const char* ws = smem + W1S + ((kt + base) & 1) * PLANE;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ko = ks ? ko1 : ko0;
        const f16x8 b0 = *(const f16x8*)(ws + b1_row + ko);
        const f16x8 b1 = *(const f16x8*)(ws + b1_row + 16 * 128 + ko);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f16x8 af = *(const f16x8*)(xs + a_row + i * 16 * 128 + ko);
          gacc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, af, gacc[i][0], 0, 0, 0);
          gacc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, af, gacc[i][1], 0, 0, 0);
        }
      }
    }
    // ---- GEGLU in registers -> H tile (fp16) into the stage that held the last K-tile -------------------------
    //   gacc[i][j][r] = G[row wm*64 + i*16 + fr][packed col wn*32 + j*16 + fq*4 + r], j = 0 value / 1 gate
    __builtin_amdgcn_s_barrier();     // every wave is done reading that stage
    char* hs = smem + W1S + ((KT - 1 + base) & 1) * PLANE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { u32x2 u; f16 e[4]; } t;
#pragma unroll
      for (int r = 0; r < 4; ++r) t.e[r] = (f16)((gacc[i][0][r] + bv[r]) * ANIP_GELU(gacc[i][1][r] + bg[r]));
      const int row = wm * 64 + i * 16 + fr;
      // hidden unit wn*16 + fq*4 .. +3 of the chunk: 16-B slot (wn*2 + fq/2) ^ (row & 7), byte (fq & 1) * 8 in it
      *(u32x2*)(hs + row * 128 + (((wn * 2 + (fq >> 1)) ^ (row & 7)) << 4) + (fq & 1) * 8) = t.u;
    }
    if (c + 1 < NCHUNK) issue_w1(c + 1, 0, (base ^ 1) & 1);   // next chunk's first K-tile -> the other stage (read last at kt = KT-2)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();     // H complete; W2 chunk landed (every wave drained its pieces at the kt >= 1 waits)
    // ---- out += H · W2c^T (K = 64) -----------------------------------------------------------------------------
    const char* w2s = smem + W2S;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks ? ko1 : ko0;
      f16x8 bf[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *(const f16x8*)(w2s + (tile_c(j) + fr) * 128 + ko);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f16x8 af = *(const f16x8*)(hs + a_row + i * 16 * 128 + ko);
#pragma unroll
        for (int j = 0; j < FN; ++j) oacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af, oacc[i][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: oacc[i][j][r] = out[m0 + wm*64 + i*16 + fr][wn*(C_/4) + j*16 + fq*4 + r] ---------------------------
  const int tsel = fq & 1, csel = (fq >> 1) * 8;
  // residual vectors of a column pair in flight together AHEAD of its stores (a load written after a store stays after
  // it: `out` may alias anything as far as the compiler knows — see gemm2.hip)
  // Full 128-B lines (round 3, as in gemm2.hip's tight epilogue): the wave's first four tiles are 64 consecutive columns; the
  // two column pairs of a 16-row block trade halves across lanes fr <-> fr ^ 8, so that store A covers rows 0-7 and store
  // B rows 8-15 of the block with 8 lanes x 16 B per row (64-B row segments: 4.8 TB/s, 128-B ones: 6.9 TB/s for the same
  // bytes, tools/exp_store_pattern.py).  The residual is fetched in the same two shapes.
  if constexpr (FN == 5 || FN == 4) {
    const int r8 = fr & 7;
    const int n = tile_c(0) + (fr >> 3) * 32 + tsel * 16 + csel;
    float add[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) add[e] = 0.f;
    if (a.b2 != nullptr) {
      const float4 b0 = *(const float4*)(a.b2 + n), b1 = *(const float4*)(a.b2 + n + 4);
      add[0] = b0.x; add[1] = b0.y; add[2] = b0.z; add[3] = b0.w;
      add[4] = b1.x; add[5] = b1.y; add[6] = b1.z; add[7] = b1.w;
    }
    U4H8 resA[4], resB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ma = m0 + wm * 64 + i * 16 + r8;
      resA[i].u = u32x4{0u, 0u, 0u, 0u};
      resB[i].u = u32x4{0u, 0u, 0u, 0u};
      if (a.res != nullptr && ma < a.M) resA[i].u = *(const u32x4*)(a.res + (int64_t)ma * C_ + n);
      if (a.res != nullptr && ma + 8 < a.M) resB[i].u = *(const u32x4*)(a.res + (int64_t)(ma + 8) * C_ + n);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ma = m0 + wm * 64 + i * 16 + r8;
      float va[8], vb[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x0 = oacc[i][0][r], x1 = oacc[i][1][r], y0 = oacc[i][2][r], y1 = oacc[i][3][r];
        ffn_row_swap(x0, x1);
        ffn_row_swap(y0, y1);
        va[r] = half_swap_hi(x0, y0);
        va[4 + r] = half_swap_hi(x1, y1);
        vb[r] = half_swap_lo(y0, x0);
        vb[4 + r] = half_swap_lo(y1, x1);
      }
      U4H8 ta, tb;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ta.e[e] = (f16)((va[e] + add[e]) + (float)resA[i].e[e]);
        tb.e[e] = (f16)((vb[e] + add[e]) + (float)resB[i].e[e]);
      }
      if (ma < a.M) *(u32x4*)(a.out + (int64_t)ma * C_ + n) = ta.u;
      if (ma + 8 < a.M) *(u32x4*)(a.out + (int64_t)(ma + 8) * C_ + n) = tb.u;
    }
  } else {
#pragma unroll
    for (int jp = 0; jp < FN / 2; ++jp) {
      const int n = tile_c(2 * jp) + tsel * 16 + csel;
      U4H8 res[4];
      float add[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) add[e] = 0.f;
      if (a.b2 != nullptr) {
        const float4 b0 = *(const float4*)(a.b2 + n), b1 = *(const float4*)(a.b2 + n + 4);
        add[0] = b0.x; add[1] = b0.y; add[2] = b0.z; add[3] = b0.w;
        add[4] = b1.x; add[5] = b1.y; add[6] = b1.z; add[7] = b1.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        res[i].u = u32x4{0u, 0u, 0u, 0u};
        if (a.res != nullptr && m < a.M) res[i].u = *(const u32x4*)(a.res + (int64_t)m * C_ + n);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = oacc[i][2 * jp][r], y = oacc[i][2 * jp + 1][r];
          ffn_row_swap(x, y);
          v[r] = x;
          v[4 + r] = y;
        }
        if (m < a.M) {
          U4H8 t;
#pragma unroll
          for (int e = 0; e < 8; ++e) t.e[e] = (f16)((v[e] + add[e]) + (float)res[i].e[e]);
          *(u32x4*)(a.out + (int64_t)m * C_ + n) = t.u;
        }
      }
    }
  }
  if (FN & 1) {
    const int n = tile_c(FN - 1) + fq * 4;
    union H4u { u32x2 u; f16 e[4]; };
    H4u res[4];
    f32x4 add = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.b2 != nullptr) add = *(const f32x4*)(a.b2 + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      res[i].u = u32x2{0u, 0u};
      if (a.res != nullptr && m < a.M) res[i].u = *(const u32x2*)(a.res + (int64_t)m * C_ + n);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 64 + i * 16 + fr;
      if (m < a.M) {
        H4u t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t.e[e] = (f16)((oacc[i][FN - 1][e] + add[e]) + (float)res[i].e[e]);
        *(u32x2*)(a.out + (int64_t)m * C_ + n) = t.u;
      }
    }
  }
}

}  // namespace

extern "C" int anip_ffn_geglu(const void* x, const void* w1p, const float* b1p, const void* w2, const float* b2,
                              const void* residual, void* out, int64_t M, int C, void* stream) {
  ANIP_REQUIRE(x && w1p && b1p && w2 && out, "anip_ffn_geglu: null pointer");
  ANIP_REQUIRE(C == 320, "anip_ffn_geglu: only C = 320 is built (got %d); use two anip_gemm calls", C);
  ANIP_REQUIRE(M > 0 && M * (int64_t)C * 2 < 0xFFFFF000ll, "anip_ffn_geglu: bad M=%lld", (long long)M);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)w1p | (uintptr_t)w2 | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)b2) & 15) == 0,
               "anip_ffn_geglu: pointers must be 16-B aligned");
  FfnArgs a;
  a.x = (const f16*)x; a.w1p = (const f16*)w1p; a.b1p = b1p; a.w2 = (const f16*)w2; a.b2 = b2;
  a.res = (const f16*)residual; a.out = (f16*)out; a.M = (int)M;
  a.gamma = nullptr; a.beta = nullptr; a.eps = 0.f;
  constexpr int LDS = (320 / 64 + 2) * 128 * 128 + 320 * 128;
  if (anip_raise_lds_limit((const void*)ffn_geglu_kernel<320, false>, LDS) != 0) {
    anip_set_error("anip_ffn_geglu: cannot raise the dynamic LDS limit to %d bytes", LDS);
    return -2;
  }
  {
    AnipProfScope prof_(ANIP_K_GEMM, stream);
    hipLaunchKernelGGL((ffn_geglu_kernel<320, false>), dim3((unsigned)((M + 127) / 128)), dim3(512), LDS, (hipStream_t)stream, a);
  }
  ANIP_LAUNCH_CHECK("anip_ffn_geglu");
  return 0;
}

// out = residual + FeedForward(LayerNorm(x; gamma, beta, eps)) — the LayerNorm applied while the x tile is staged
extern "C" int anip_ffn_geglu_ln(const void* x, const float* gamma, const float* beta, float eps, const void* w1p,
                                 const float* b1p, const void* w2, const float* b2, const void* residual, void* out,
                                 int64_t M, int C, void* stream) {
  ANIP_REQUIRE(x && gamma && beta && w1p && b1p && w2 && out, "anip_ffn_geglu_ln: null pointer");
  ANIP_REQUIRE(C == 320, "anip_ffn_geglu_ln: only C = 320 is built (got %d); use anip_layernorm + two anip_gemm calls", C);
  ANIP_REQUIRE(M > 0 && M * (int64_t)C * 2 < 0xFFFFF000ll, "anip_ffn_geglu_ln: bad M=%lld", (long long)M);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w1p | (uintptr_t)w2 | (uintptr_t)out |
                 (uintptr_t)residual | (uintptr_t)b2) & 15) == 0, "anip_ffn_geglu_ln: pointers must be 16-B aligned");
  FfnArgs a;
  a.x = (const f16*)x; a.w1p = (const f16*)w1p; a.b1p = b1p; a.w2 = (const f16*)w2; a.b2 = b2;
  a.res = (const f16*)residual; a.out = (f16*)out; a.M = (int)M;
  a.gamma = gamma; a.beta = beta; a.eps = eps;
  constexpr int LDS = (320 / 64 + 2) * 128 * 128 + 320 * 128;
  if (anip_raise_lds_limit((const void*)ffn_geglu_kernel<320, true>, LDS) != 0) {
    anip_set_error("anip_ffn_geglu_ln: cannot raise the dynamic LDS limit to %d bytes", LDS);
    return -2;
  }
  {
    AnipProfScope prof_(ANIP_K_GEMM, stream);
    hipLaunchKernelGGL((ffn_geglu_kernel<320, true>), dim3((unsigned)((M + 127) / 128)), dim3(512), LDS, (hipStream_t)stream, a);
  }
  ANIP_LAUNCH_CHECK("anip_ffn_geglu_ln");
  return 0;
}
