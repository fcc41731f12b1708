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
//  * epilogue straight from the accumulators, no LDS staging and no barriers: the MFMA is issued with the
//    operands swapped (C^T tiles), so a lane holds 4 consecutive output columns of one row; one
//    v_permlane16_swap per dword between two neighbouring 16-column tiles widens that to 8 consecutive
//    columns = one 16-B store / residual load per lane.  Fused: bias, per-row-group bias, GEGLU (value and
//    gate tiles of one output column sit in the same lane: weights packed per 32 columns as [16 v | 16 g]),
//    residual, fp32 out; the transposed-out variant (V^T projection) keeps the un-swapped operand order and
//    pairs tiles along M instead.  Single rounding to fp16.
//  * XCD-aware bijective tile order: each XCD's private L2 sees a contiguous run of tiles sharing A panels.
//  * (round 3) wide 256 x {256,320} x 64 tiles with a quarter-phased, role-alternating main loop; launched as a
//    PERSISTENT WALK when there are more tiles than CUs (the next tile's first K-tile is staged in front of the running
//    tile's epilogue); tight-epilogue stores / residual loads in whole 128-B lines (a DPP half swap between the two
//    column pairs of a wave); wide-tile split-K for the 8x8 / 16x16 levels; 64-deep K-tiles for one-tile-per-CU launches.
#include <stdlib.h>

#include "common.h"

// GEGLU activation of the epilogues: gelu_poly_f (FMA pipe only; default) or, with -DANIP_GELU_EXACT, gelu_fast_f (A&S erf,
// |error| <= 1.5e-7, two transcendentals)
#ifdef ANIP_GELU_EXACT
#define ANIP_GELU gelu_fast_f
#else
#define ANIP_GELU gelu_poly_f
#endif

namespace {

constexpr uint32_t OOB = 0xFFFFFFF0u;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// bank swizzle of the 16-B chunk index inside an LDS row (applied on the DMA SOURCE address and on the
// ds_read_b128 side; the DMA destination is lane-linear):
//   64-B rows (BK = 32, 4 chunks): chunk ^= 2 * bit3(row);  128-B rows (BK = 64, 8 chunks): chunk ^= row & 7
template <int BKT>
__device__ __forceinline__ int swz_of(int row) {
  return BKT == 32 ? ((row >> 3) & 1) * 2 : (row & 7);
}

// rows of 16 lanes: swap the odd rows of x with the even rows of y (v_permlane16_swap).  Afterwards, for two
// tiles X, Y whose lanes (row fq) each held columns fq*4..+3:  lanes of even rows hold tile X columns
// (fq/2)*8..+7 as [x, y], lanes of odd rows hold tile Y columns (fq/2)*8..+7 as [x, y].
__device__ __forceinline__ void row_swap(float& x, float& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// Tile configurations (BM2 x BN x BKT, NW waves arranged (NW/WNW) x WNW):
//   256 x {128,160} x 32, 8 waves 4x2, 3-stage ring, 2 blocks/CU  — many short-K tiles
//   128 x {128,160} x 32, 4 waves 2x2, 3-stage ring, up to 3 blocks/CU — small problems
//   256 x {256,320} x 64, 8 waves 2x4 (wave tile 128 x {64,80}), 2-stage, 1 block/CU — the global->LDS traffic per
//     flop drops by 1/4..1/3 and every DMA row is a full 128-B line (BK = 32 rows are half lines), which is what
//     bounds the smaller tiles (measured: DMA alone = 0.93 ms of the 1.18 ms 8192^3 GEMM).
//   wide tiles (NST = 2, BK = 64, 8 waves): the quarter-phased main loop (round 3), see "quarter-phased schedule" below.
template <int BM2, int BN, int NW, int WNW, int BKT, int NST, bool CONV, bool TRANS>
__global__ __launch_bounds__(NW * 64, (WNW == 4 ? 2 : (NW == 8 ? 4 : 2))) void gemm2_kernel(const anip_gemm_params p,
                                                                                           const int skip_epilogue, const int splitk) {
  constexpr int NT2 = NW * 64;
  constexpr int RB = BKT * 2;                  // LDS row bytes
  constexpr int CPR = RB / 16;                 // 16-B chunks per row
  constexpr int RPI = 1024 / RB;               // rows per LDS-DMA instruction
  // NST = ring depth
  constexpr int KH = BKT / 32;                 // 32-deep MFMA steps per K-tile
  constexpr int WMW = NW / WNW;                // waves along M
  constexpr int WTM = BM2 / WMW, WTN = BN / WNW;  // wave tile
  constexpr int FM = WTM / 16, NB = WTN / 16;  // 16x16 MFMA tiles per wave along M / N
  constexpr int A_BYTES = BM2 * RB, B_BYTES = BN * RB, STAGE = A_BYTES + B_BYTES;
  constexpr int NA_I = BM2 / RPI / NW;         // A DMA instructions per wave per K-tile
  constexpr int NB_TOT = BN / RPI;             // B DMA instructions per K-tile
  constexpr int NB_I = (NB_TOT + NW - 1) / NW; // max per wave
  static_assert(BM2 % (RPI * NW) == 0 && WTM % 16 == 0 && WTN % 16 == 0 && (FM % 2) == 0, "bad tile configuration");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNW, wn = wave % WNW;

  const int nbm = (p.M + BM2 - 1) / BM2, nbn = (p.N + BN - 1) / BN, nblk = nbm * nbn;
  // virtual block id -> tile origin.  A launch with fewer workgroups than tiles (the PERSISTENT form of the quarter-phased
  // wide tiles, see launch_gemm2) walks vb = blockIdx.x, + gridDim.x, ...: gridDim.x is a multiple of 8 there, so vb & 7 is
  // the XCD the workgroup runs on and the XCD-aware order below holds for every tile of the walk.
  auto tile_of = [&](int vb_, int& m0_, int& n0_) {
    int swz;
    {
      const int bid = vb_, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
      swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // tile index -> (bm, bn).  Few column tiles (every layer with N <= 4 tiles): row-major — consecutive tiles of an XCD
    // share the A panel.  Many column tiles (GEGLU ff-in N = 5120 / 10240, the 16x16 qkv N = 3840): groups of GM = 8 row
    // tiles walked column-major inside the group, so that the ~32 tiles an XCD runs at a time are 8 (M) x 4 (N) — 12
    // operand panels through its L2 instead of 1 + 32 (rocprofv3: the M = 8192, N = 10240, K = 1280 GEGLU GEMM fetched
    // 884 MB per launch for 131 MB of operands, the weight matrix once per row tile).
    int bm, bn;
    {
      constexpr int GM = 8;
      // (only when the weight matrix does not fit an XCD's L2 next to the streaming A panels: with a resident W — the
      // 64x64 ff-in, 1.6 MB — row-major order reads every operand exactly once and measured 4 % faster)
      if (nbn >= 8 && (int64_t)p.N * p.K * 2 > (2 << 20)) {
        const int per = GM * nbn, grp_ = swz / per, first = grp_ * GM, rem = swz - grp_ * per;
        const int gsz = min(nbm - first, GM);
        bm = first + rem % gsz;
        bn = rem / gsz;
      } else {
        bm = swz / nbn;
        bn = swz % nbn;
      }
    }
    m0_ = bm * BM2;
    n0_ = bn * BN;
  };
  int vb = blockIdx.x, m0, n0;
  tile_of(vb, m0, n0);

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
  const int lr = lane / CPR, ls = lane % CPR;  // row within the DMA instruction's row group, 16-B slot within the row
  // Everything an A DMA instruction needs per lane is precomputed here, so that issuing one inside the main loop is a
  // scalar add + a select (round 2 recomputed row * lda with 64-bit multiplies and re-read lda / lda2 from the kernel
  // arguments for every instruction: s_load + s_waitcnt + v_mul_lo in the middle of the load segment):
  //   plain   a_v0[i] / a_v1[i] = byte offset of (row, chunk g) at k = 0 in source 1 / source 2 (OOB if row >= M);
  //           the K position travels in the instruction's scalar offset
  //   conv    a_v0[i] = byte offset of channel chunk g of the window's CENTRE pixel (oy*stride, ox*stride), a_v1[i] =
  //           bit (3 dy + dx) set iff tap (dy, dx) of this row lies inside the image; the tap / channel position is a
  //           per-K-tile scalar delta.  Fused nearest-2x upsample (3 convs per UNet call): a_v0 = image base pixel index,
  //           a_v1 = packed window origin (y0 + 1) << 16 | (x0 + 1) in the upsampled grid (round 2's computation).
  uint32_t a_v0[NA_I], a_v1[NA_I];
  // 16-B chunk (within the K-tile's row) this lane fetches: the swizzle depends on the row inside the instruction's row
  // group only (the groups start at multiples of 8 / 16 rows), so it is the same for every A instruction of the lane
  const int a_g = ls ^ swz_of<BKT>(lr);
  // RPI-row block of the A tile that this wave's i-th DMA instruction fills.  Narrow tiles: blocks wave*NA_I + i.
  // Wide tiles (BM2 = 256, 8 waves 2 x 4, NA_I = 4): instructions 0, 1 fill "A-lo" blocks — the first 64 rows of each
  // 128-row wave-row band — and 2, 3 the "A-hi" blocks, so that the two halves can be staged (and waited for) separately.
  auto a_blk = [&](int i) -> int {
    if (!(NST == 2 && BKT == 64 && NW == 8)) return wave * NA_I + i;
    const int r = 2 * wave + (i & 1);            // 0..15 within the half
    return (r >> 3) * 16 + (i >> 1) * 8 + (r & 7);
  };
  uint32_t b_off[NB_I];
  // (per tile: the persistent form calls it again, in place, once the last K-tile of the running tile has been issued)
  auto setup_lanes = [&](const int m0_, const int n0_) {
#pragma unroll
  for (int i = 0; i < NA_I; ++i) {
    const int row = a_blk(i) * RPI + lr;
    const int g = ls ^ swz_of<BKT>(row);
    const int m = m0_ + row;
    const bool okm = m < p.M;
    if (CONV) {
      const int hw = p.Hout * p.Wout;
      const int mm = okm ? m : 0;
      const int img = mm / hw, rem = mm - img * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
      uint32_t mask = 0;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const int y = y0 + t9 / 3, x = x0 + t9 % 3;
        mask |= (okm && y >= 0 && y < p.Hin && x >= 0 && x < p.Win) ? (1u << t9) : 0u;
      }
      const uint32_t pix0 = (uint32_t)img * (uint32_t)(p.Hin * p.Win);
      const uint32_t centre = ((pix0 + (uint32_t)(oy * p.stride * p.Win + ox * p.stride)) * (uint32_t)p.Cin + (uint32_t)(g * 8)) * 2u;
      const bool ups = p.upsample != 0;
      a_v0[i] = ups ? (okm ? pix0 : 0xFFFFFFFFu) : centre;
      a_v1[i] = ups ? (((uint32_t)(y0 + 1) << 16) | (uint32_t)(x0 + 1)) : mask;
    } else {
      a_v0[i] = okm ? (uint32_t)(((int64_t)m * p.lda + g * 8) * 2) : OOB;
      a_v1[i] = (okm && p.A2 != nullptr) ? (uint32_t)(((int64_t)m * p.lda2 + g * 8) * 2) : OOB;
    }
  }
#pragma unroll
  for (int i = 0; i < NB_I; ++i) {
    const int j = wave + NW * i;
    const int row = j * RPI + lr;
    const int g = ls ^ swz_of<BKT>(row);
    const int n = n0_ + row;
    b_off[i] = (j < NB_TOT && n < p.N) ? (uint32_t)(((int64_t)n * p.ldw + g * 8) * 2) : OOB;
  }
  };
  setup_lanes(m0, n0);
  const int my_b = (NB_TOT - wave + NW - 1) / NW;  // B DMA instructions this wave issues (wave-uniform)

  // one A / one B DMA instruction of K-tile kt into ring stage `stage`
  auto issue_a1 = [&](int kt, int stage, int i) {
    char* sa = smem + stage * STAGE;
    const int k0 = kt * BKT;
    const bool ktail = k0 + BKT > p.K;         // wave-uniform
    if (CONV) {
      // (tap, first channel) of the K-tile; uniform over it.  conv = 1: tap-major K (Cin % BKT == 0); conv = 2:
      // channel-block-major K, [Cin/64][9 taps][64] (Cin % 64 == 0): the nine taps of a channel block are consecutive
      int tap, c0;
      if (p.conv == 2) {
        const int cb = k0 / 576, r = k0 - cb * 576;
        tap = r >> 6;
        c0 = cb * 64 + (r & 63);
      } else {
        tap = k0 / p.Cin;
        c0 = k0 - tap * p.Cin;
      }
      const int dy = tap / 3, dx = tap - dy * 3;
      uint32_t vo;
      if (p.upsample) {
        int y = (int)(a_v1[i] >> 16) - 1 + dy, x = (int)(a_v1[i] & 0xFFFFu) - 1 + dx;
        const bool ok = a_v0[i] != 0xFFFFFFFFu && y >= 0 && y < 2 * p.Hin && x >= 0 && x < 2 * p.Win;
        y >>= 1; x >>= 1;
        vo = ok ? ((a_v0[i] + (uint32_t)(y * p.Win + x)) * (uint32_t)p.Cin + (uint32_t)(c0 + a_g * 8)) * 2u : OOB;
      } else {
        const int delta = (((dy - p.pad) * p.Win + (dx - p.pad)) * p.Cin + c0) * 2;     // scalar
        vo = ((a_v1[i] >> tap) & 1u) ? a_v0[i] + (uint32_t)delta : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sa + a_blk(i) * 1024), 16, vo, 0, 0, 0);
    } else {
      const bool second = (p.A2 != nullptr) && (k0 >= p.K1);   // wave-uniform (K1 % BKT == 0)
      const int kk = second ? k0 - p.K1 : k0;
      uint32_t vo = second ? a_v1[i] : a_v0[i];
      if (ktail) {
        const int klim = second ? p.K - p.K1 : (p.A2 ? p.K1 : p.K);
        if (kk + a_g * 8 >= klim) vo = OOB;
      }
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(sa + a_blk(i) * 1024), 16, vo, (uint32_t)kk * 2u, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(sa + a_blk(i) * 1024), 16, vo, (uint32_t)kk * 2u, 0, 0);
    }
  };
  auto issue_b1 = [&](int kt, int stage, int i) {
    char* sb = smem + stage * STAGE + A_BYTES;
    const int k0 = kt * BKT;
    uint32_t vo = b_off[i];
    if (k0 + BKT > p.K) {
      const int row = (wave + NW * i) * RPI + lr;
      if (k0 + (ls ^ swz_of<BKT>(row)) * 8 >= p.K) vo = OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(sb + (wave + NW * i) * 1024), 16, vo, (uint32_t)k0 * 2u, 0, 0);
  };
  auto issue = [&](int kt, int stage) {
#pragma unroll
    for (int i = 0; i < NA_I; ++i) issue_a1(kt, stage, i);
#pragma unroll
    for (int i = 0; i < NB_I; ++i)
      if (i < my_b) issue_b1(kt, stage, i);
  };

  // ---- fragment read offsets ------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  int koff[KH];
#pragma unroll
  for (int kh = 0; kh < KH; ++kh) koff[kh] = ((kh * 4 + fq) ^ swz_of<BKT>(fr)) << 4;
  const int a_row_off = (wm * WTM + fr) * RB;
  // Column (within the block tile) of this wave's j-th 16-column MFMA tile.  Even tile counts: the wave's WTN columns
  // are contiguous.  Odd counts (BN = 160 / 320: 5 tiles = 80 columns = 160 B per wave): contiguous ranges would put
  // every second wave at an odd multiple of 80 columns, i.e. all its 64-B row segments (the epilogue pairs two tiles:
  // 4 lanes x 16 B) 32 B off the 64-B access granule — rocprofv3 WRITE_SIZE 1.5x the output bytes on the N = 960
  // layer, the residual reads likewise.  Instead each wave takes NB - 1 tiles from a 64-B aligned contiguous range and
  // its last (single) tile from the tail of the block tile.
  // (BN = 160: every second block tile starts 64 B off a 128-B line.  Dealing the single tiles FIRST there — so that the
  // four-tile groups of the epilogue's full-line stores start on a line — measured 10 % SLOWER on the N = K = 320 layers,
  // profiles/r03: one deal for all tiles.)
  auto tile_c = [&](int j) -> int {
    if ((NB & 1) == 0) return wn * WTN + j * 16;
    return j < NB - 1 ? wn * (NB - 1) * 16 + j * 16 : WNW * (NB - 1) * 16 + wn * 16;
  };

  // K-tile range of this block: all of K, or — split-K, blockIdx.y = slice — one of `splitk` contiguous slices whose
  // fp32 partial tile goes to the workspace (the batch offset of the epilogue) and is reduced by splitk_reduce_kernel
  int kt_begin = 0, nk = (p.K + BKT - 1) / BKT;
  if (splitk > 1) {
    const int per = (nk + splitk - 1) / splitk;
    kt_begin = (int)blockIdx.y * per;
    nk = max(0, min(nk - kt_begin, per));
  }
  constexpr bool PHASED = (NST == 2 && KH == 2 && NW == 8);   // the wide tiles: quarter-phased main loop, persistent walk
  f32x4 acc[FM][NB];

  // ======== quarter-phased schedule (PHASED): state, DMA operand preparation, prologue ========
  // Quarter-phased schedule (round 3).  Round 2's role-alternating loop below issues the WHOLE next K-tile (8-9 LDS-DMA
  // instructions per wave, 36 KiB per wave group through the CU's one 64 B/clk address path) inside ONE of its four
  // load segments per K-tile and drains it (vmcnt(0)) once per K-tile: that segment is ~3x longer than the 32-MFMA
  // compute segment of the partner wave it is supposed to hide behind, and the matrix pipe waits at the barrier.
  // Here a K-tile is FOUR sub-steps of 4 x NB MFMAs (a wave's 128 x WTN tile as [A-lo | A-hi] x [k-half 0 | 1]), the
  // next tile's DMA is dealt over the four load segments — B first half, B second half, A-lo, A-hi: 2-3 instructions
  // each — and the queue is never drained: counted vmcnt at two points per tile,
  //   sub-step order   j=0: A-lo x B(kh0)   j=1: A-hi x B(kh0)   j=2: A-hi x B(kh1)   j=3: A-lo x B(kh1)
  //   LOAD(j) reads    bf, af                af                    bf (replaced), af    af         (ONE set of B registers)
  //   LOAD(j) stages   B blocks 0,1 of t+1   B blocks 2.. of t+1   A-lo of t+1          A-hi of t+1      (per wave)
  //   wait before bar  vmcnt(2): A-hi(t)     lgkm                  lgkm                 vmcnt(2): all of t+1 but A-hi
  // A-hi of tile t+1, staged last, is first read at j=1 of tile t+1 and waited for at the end of LOAD(j=0) there.
  // Waves 4-7 run one barrier behind waves 0-3 as before (a wave's compute segment coincides with its SIMD partner's
  // load segment).  Hazards, with barrier #b closing interval I(b); group 0 runs LOAD(s) in I(2s) and COMPUTE(s) in
  // I(2s+1), group 1 LOAD(s) in I(2s+1), COMPUTE(s) in I(2s+2), s = 4t + j:
  //   RAW  every wave waits for its own DMA share before a barrier that every reader passes before it reads: the
  //        j=3 wait of tile t is before #8t+6 (group 0) / #8t+7 (group 1), the first reads of tile t+1 come after
  //        #8t+7 / #8t+8; the j=0 wait of tile t+1 before #8t+8 / #8t+9, the A-hi reads after #8t+9 / #8t+10.
  //   WAR  a region of the other stage is re-staged in tile t no earlier than the sub-step that last read it in tile
  //        t-1 (B: from j=0, read last at j=2; A-lo: j=2, read last at j=3; A-hi: j=3, read last at j=2) — a whole
  //        K-tile later, and every LOAD ends with lgkmcnt(0) before its barrier.
  // (First version of this loop: both B fragment sets live, A-hi needed only at j=2.  The 256 x 320 convolution kernel
  // then sits at 253-256 VGPRs and any extra state spills INTO the loop: 3x slower, profiles/r03/h_*.)
  if constexpr (PHASED) {
    static_assert(FM == 8 && NA_I == 4 && NB_TOT % NW == 0, "quarter-phased schedule: 256-row tile, 2 x 4 waves");
  }
  // DMA operands of the NEXT K-tile are PREPARED inside a compute segment (scalar / vector address arithmetic hides
  // between the MFMAs) so that a load segment carries only the bare `buffer_load ... lds` instructions: with the
  // arithmetic inside the load segments those ran 45-126 non-memory instructions each (conv: tap decode, window test,
  // per-lane select; plain: source select, K-tail test), longer than the 16-20 MFMAs of the partner they hide behind.
  // Prepared state is SCALAR only (the per-lane part of an A offset is 2-4 VALU operations at the point of issue):
  //   conv    p_tap (window tap), p_delta (byte offset of (tap, first channel) relative to the window's centre pixel)
  //   plain   pa_second (second A source), pa_soff (byte offset of the K position inside the source row)
  //   pb_soff byte offset of the K-tile inside a W row;  p_ktail: the tile crosses K (per-lane K-tail tests, rare)
  uint32_t pa_soff = 0, pb_soff = 0;
  int p_tap = 0, p_delta = 0, p_c0 = 0;
  bool pa_second = false, p_ktail = false;
  // conv K cursor of the tile to prepare next (no division in the loop): tap, first channel
  int cv_tap = 0, cv_c0 = 0;
  // conv, K order 2, no upsample (every hot conv): the byte delta of the K-tile's (tap, channel block) relative to the
  // window's centre pixel advances by one of three constants per K-tile — next tap in the row, next row, next channel
  // block — so that `prepare` is a handful of scalar adds (round 3's first version recomputed it with divisions and
  // multiplies from the kernel arguments: 400 cycles behind the MFMAs of every fourth compute segment).
  const bool cv_fast = CONV && p.conv == 2 && !p.upsample && BKT == 64;
  int cv_delta = 0;                                                     // delta of the tile to prepare next
  // the cursor at K-tile kt_begin + 1: in front of the first output tile and of every further tile of a persistent walk
  auto reset_cursor = [&]() {
    if (!CONV) return;
    const int k1 = (kt_begin + 1) * BKT;
    if (p.conv == 2) {
      const int cb = k1 / 576, r = k1 - cb * 576;
      cv_tap = r >> 6;
      cv_c0 = cb * 64 + (r & 63);
    } else {
      cv_tap = k1 / p.Cin;
      cv_c0 = k1 - cv_tap * p.Cin;
    }
    if (cv_fast) {
      const int dy = cv_tap / 3, dx = cv_tap - dy * 3;
      cv_delta = (((dy - p.pad) * p.Win + (dx - p.pad)) * p.Cin + cv_c0) * 2;
    }
  };
  reset_cursor();
  auto prepare = [&](int kt) {
    const int k0 = kt * BKT;
    p_ktail = k0 + BKT > p.K;
    pb_soff = (uint32_t)k0 * 2u;
    if (CONV) {
      p_tap = cv_tap;
      p_c0 = cv_c0;
      if (cv_fast) {
        p_delta = cv_delta;
        const bool row_end = cv_tap == 2 || cv_tap == 5;
        const bool blk_end = cv_tap == 8;
        const int stepx = p.Cin * 2;                          // tap + 1 inside a row
        cv_delta += blk_end ? 128 - (2 * p.Win + 2) * stepx   // tap 8 -> tap 0 of the next 64-channel block
                            : (row_end ? (p.Win - 2) * stepx : stepx);   // tap 2 -> 3, 5 -> 6 / next tap
        cv_tap = blk_end ? 0 : cv_tap + 1;
      } else {
        const int dy = p_tap / 3, dx = p_tap - dy * 3;
        p_delta = (((dy - p.pad) * p.Win + (dx - p.pad)) * p.Cin + p_c0) * 2;
        // advance the cursor by one K-tile — value selects only: conditional stores to the two cursor variables get
        // merged by the compiler into one store through a selected POINTER, which puts them in scratch memory
        const int c = cv_c0 + BKT;
        const bool blk = p.conv == 2 ? ((c & 63) == 0) : (c >= p.Cin);   // left the channel block / the tap
        const int tap1 = cv_tap + (blk ? 1 : 0);
        const bool wrap9 = p.conv == 2 && tap1 == 9;                       // conv = 2: next 64-channel block
        cv_tap = wrap9 ? 0 : tap1;
        cv_c0 = p.conv == 2 ? ((blk && !wrap9) ? c - 64 : c) : (blk ? 0 : c);
      }
    } else {
      pa_second = (p.A2 != nullptr) && (k0 >= p.K1);   // wave-uniform (K1 % BKT == 0)
      pa_soff = (uint32_t)(pa_second ? k0 - p.K1 : k0) * 2u;
    }
  };
  auto fire_a = [&](int stage, int i) {
    char* dst = smem + stage * STAGE + a_blk(i) * 1024;
    if (CONV) {
      uint32_t vo;
      if (p.upsample) {                    // 3 convs per UNet call: round 2's per-issue arithmetic
        const int dy = p_tap / 3, dx = p_tap - dy * 3;
        int y = (int)(a_v1[i] >> 16) - 1 + dy, x = (int)(a_v1[i] & 0xFFFFu) - 1 + dx;
        const bool ok = a_v0[i] != 0xFFFFFFFFu && y >= 0 && y < 2 * p.Hin && x >= 0 && x < 2 * p.Win;
        y >>= 1; x >>= 1;
        vo = ok ? ((a_v0[i] + (uint32_t)(y * p.Win + x)) * (uint32_t)p.Cin + (uint32_t)(p_c0 + a_g * 8)) * 2u : OOB;
      } else {
        vo = ((a_v1[i] >> p_tap) & 1u) ? a_v0[i] + (uint32_t)p_delta : OOB;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(dst), 16, vo, 0, 0, 0);
    } else {
      uint32_t vo = pa_second ? a_v1[i] : a_v0[i];
      if (p_ktail) {                       // rare: K % 64 != 0, last tile only
        const int klim = pa_second ? p.K - p.K1 : (p.A2 ? p.K1 : p.K);
        if ((int)(pa_soff >> 1) + a_g * 8 >= klim) vo = OOB;
      }
      if (pa_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA2, LDS_PTR(dst), 16, vo, pa_soff, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(dst), 16, vo, pa_soff, 0, 0);
    }
  };
  auto fire_b = [&](int stage, int i) {
    uint32_t vo = b_off[i];
    if (p_ktail) {                         // rare: K % 64 != 0, last tile only
      const int row = (wave + NW * i) * RPI + lr;
      if ((int)(pb_soff >> 1) + (ls ^ swz_of<BKT>(row)) * 8 >= p.K) vo = OOB;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(smem + stage * STAGE + A_BYTES + (wave + NW * i) * 1024), 16, vo, pb_soff, 0, 0);
  };
  f16x8 bf[NB], af[4];
  const int a_lo = a_row_off, a_hi = a_row_off + 64 * RB;
  // (One barrier per sub-step instead of two — without the mid-step rendezvous — was measured 2-5 % slower on every wide shape,
  //  profiles/r03/t_kbench_*: the two waves of a SIMD drift into computing / loading at the same time.)
#define ANIP_G2_BAR_RAW()            \
  __builtin_amdgcn_sched_barrier(0); \
  __builtin_amdgcn_s_barrier();      \
  __builtin_amdgcn_sched_barrier(0)
#define ANIP_G2_BAR_L() ANIP_G2_BAR_RAW()
#define ANIP_G2_BAR_C() ANIP_G2_BAR_RAW()
#define ANIP_G2_MMA(I0)                                                                                           \
  __builtin_amdgcn_s_setprio(1);                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)                    \
      acc[(I0) + i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[(I0) + i][j], 0, 0, 0)  \
                               : __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[(I0) + i][j], 0, 0, 0); \
  __builtin_amdgcn_s_setprio(0)
#define ANIP_G2_DMA_B(I)    \
  if (CONV) fire_b(nst, I); \
  else issue_b1(kt_begin + t + 1, nst, I)
#define ANIP_G2_DMA_A(I)    \
  if (CONV) fire_a(nst, I); \
  else issue_a1(kt_begin + t + 1, nst, I)
  const int grp = wave >> 2;
  if constexpr (PHASED) {
    if (nk > 0) issue(kt_begin, 0);
    if (CONV && nk > 1) prepare(kt_begin + 1);
  }

  // ======== tile loop: one pass per output tile.  Every form but the persistent one has gridDim.x == nblk: one pass.
  // Persistent walk (quarter-phased wide tiles, launch_gemm2): once the main loop of a tile is through — every wave past
  // the closing rendezvous, so no LDS read of the tile is outstanding — a wave recomputes its per-lane source offsets for
  // the NEXT tile of the walk, issues that tile's first K-tile into stage 0, and only then runs the epilogue: the first
  // operands of the next tile arrive under the stores of this one, and a tile starts without the workgroup launch, the
  // kernel-argument loads and the exposed HBM round trip of a fresh workgroup (5-14 % of a K <= 1280 tile).
  for (;;) {
    const int vbn = vb + (int)gridDim.x;
    const bool has_next = (PHASED) && vbn < nblk;   // block-uniform
    int m0n = 0, n0n = 0;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // "tight" blocks — tile fully inside the output, 16-B accessible operands, alpha == 1: every hot shape of the path —
    // start the accumulators from the per-column additive terms (bias and, when the block's rows share one row group,
    // the row-group bias): NB small L2-resident loads that travel with the prologue DMA, no registers, and nothing
    // left to fetch for them in the epilogue, where loads queue behind the tile's stores (see there).
    const bool acc_has_bias = !TRANS && splitk <= 1 && p.alpha == 1.0f && m0 + BM2 <= p.M && n0 + BN <= p.N &&
                              (p.bias != nullptr || p.rowbias != nullptr) &&
                              (p.rowbias == nullptr || (((p.ld_rowbias & 3) == 0) && ((((uintptr_t)p.rowbias) & 15) == 0)));
    const bool rb_uni = acc_has_bias && p.rowbias != nullptr &&
                        (m0 / p.rows_per_group == (m0 + BM2 - 1) / p.rows_per_group);
    if (!TRANS && acc_has_bias) {
      const float* rb_row = rb_uni ? p.rowbias + (int64_t)(m0 / p.rows_per_group) * p.ld_rowbias : nullptr;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int cb = n0 + tile_c(j) + (lane >> 4) * 4;       // column of acc[.][j][0] in this lane
        f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias != nullptr) b = *(const f32x4*)(p.bias + cb);
        if (rb_uni) b += *(const f32x4*)(rb_row + cb);
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = b;
      }
    }

    if constexpr (PHASED) {
      // K-tile 0 (issued in front of the tile loop / in front of the previous tile's epilogue) has landed for every wave;
      // on a later tile of a walk the counter also covers the previous tile's stores (in order: they were issued later)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (grp == 1) __builtin_amdgcn_s_barrier();
      for (int t = 0; t < nk; ++t) {
        const char* sa = smem + (t & 1) * STAGE;
        const char* sb = sa + A_BYTES;
        const int nst = (t + 1) & 1;
        const bool more = t + 1 < nk;                 // block-uniform
        // ---- j = 0: A-lo x B(k-half 0); stage B blocks 0, 1 of tile t+1; A-hi of THIS tile must have landed
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = *(const f16x8*)(sb + (tile_c(j) + fr) * RB + koff[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(sa + a_lo + i * 16 * RB + koff[0]);
        if (more) {
          ANIP_G2_DMA_B(0);
          ANIP_G2_DMA_B(1);
          asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");   // leaves the 2 B blocks just issued: A-hi(t) landed
        } else {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        ANIP_G2_BAR_L();
        ANIP_G2_MMA(0);
        ANIP_G2_BAR_C();
        // ---- j = 1: A-hi x B(k-half 0); stage the remaining B blocks
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(sa + a_hi + i * 16 * RB + koff[0]);
        if (more) {
#pragma unroll
          for (int i = 2; i < NB_I; ++i) { ANIP_G2_DMA_B(i); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ANIP_G2_BAR_L();
        ANIP_G2_MMA(4);
        ANIP_G2_BAR_C();
        // ---- j = 2: A-hi x B(k-half 1) (the B fragments are replaced: one set of B registers); stage A-lo of tile t+1
#pragma unroll
        for (int j = 0; j < NB; ++j) bf[j] = *(const f16x8*)(sb + (tile_c(j) + fr) * RB + koff[1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(sa + a_hi + i * 16 * RB + koff[1]);
        if (more) {
          ANIP_G2_DMA_A(0);
          ANIP_G2_DMA_A(1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ANIP_G2_BAR_L();
        ANIP_G2_MMA(4);
        ANIP_G2_BAR_C();
        // ---- j = 3: A-lo x B(k-half 1); stage A-hi of tile t+1; everything of tile t+1 but that must have landed
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const f16x8*)(sa + a_lo + i * 16 * RB + koff[1]);
        if (more) {
          ANIP_G2_DMA_A(2);
          ANIP_G2_DMA_A(3);
          asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        ANIP_G2_BAR_L();
        ANIP_G2_MMA(0);
        // conv: the scalar operands of tile t+2 (its first part is fired in the next load segment) behind these MFMAs.
        // (The plain GEMM issues with its per-instruction arithmetic in place: hoisting it measured 4-6 % SLOWER —
        // the main loop waits for DMA data, not for instruction issue; profiles/r03/e_kbench_prepared_operands.jsonl.)
        if (CONV && t + 2 < nk) prepare(kt_begin + t + 2);
        ANIP_G2_BAR_C();
      }
      if (grp == 0) __builtin_amdgcn_s_barrier();   // closing rendezvous: group 1 has finished its last LOAD
      if (has_next) {
        tile_of(vbn, m0n, n0n);
        setup_lanes(m0n, n0n);
        issue(kt_begin, 0);
        if (CONV && nk > 1) {
          reset_cursor();
          prepare(kt_begin + 1);
        }
      }
    } else {
#pragma unroll
      for (int t = 0; t < NST - 1; ++t)
        if (t < nk) issue(kt_begin + t, t);
      for (int kt = 0; kt < nk; ++kt) {
        // this wave's part of tile kt has landed; later tiles (if any were issued) stay in flight
        if (NST > 2 && kt + 1 < nk) {
          if (my_b == NB_I) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (NA_I + NB_I)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (NA_I + NB_I - 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();  // tile kt is complete for all waves; the stage read at kt-1 is free
        if (kt + NST - 1 < nk) issue(kt_begin + kt + NST - 1, (kt + NST - 1) % NST);
        const char* sa = smem + (kt % NST) * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
          f16x8 bf[NB];
#pragma unroll
          for (int t = 0; t < NB; ++t) bf[t] = *(const f16x8*)(sb + (tile_c(t) + fr) * RB + koff[kh]);
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            const f16x8 af = *(const f16x8*)(sa + a_row_off + i * 16 * RB + koff[kh]);
#pragma unroll
              for (int j = 0; j < NB; ++j)
                acc[i][j] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf[j], acc[i][j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af, acc[i][j], 0, 0, 0);
          }
        }
      }
    }

    // ---- epilogue: straight from the accumulators ------------------------------------------------------
    //   !TRANS: acc[i][j][r] = C[m0 + wm*WTM + i*16 + fr][n0 + wn*WTN + j*16 + fq*4 + r]
    //    TRANS: acc[i][j][r] = C[m0 + wm*WTM + i*16 + fq*4 + r][n0 + wn*WTN + j*16 + fr]
    // An exit that uses the accumulators as they leave the main loop; `skip_epilogue` is 0 in every launch.  It is here for the
    // register allocator: without a use of `acc` at this point the 256 x 256 instantiations spill 60 dwords more and the
    // convolution one reloads values INSIDE its main loop (16 scratch loads per K-tile: the VAE's 256-channel 256x256 convolution
    // 1.40 -> 2.14 ms) — found when round 3's "no epilogue" experiment switch, which sat here, was pruned.
    if (skip_epilogue) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;               // (ends a persistent walk, too)
    }
    float alpha = p.alpha;
    // what the accumulators already carry of the per-column additive terms (see acc_has_bias above)
    bool bias_in_acc = acc_has_bias, rb_in_acc = rb_uni;
    const int64_t obatch = (p.batch > 1) ? (int64_t)blockIdx.y * p.strideO : 0;
    const int tsel = fq & 1, csel = (fq >> 1) * 8;   // after row_swap: tile of the pair / column offset in it

    // 8 consecutive output columns [n, n+8) of row m: bias / row-group bias / residual / store
    // (add_bias / add_rb: false when the accumulators were started from that term — acc_has_bias / rb_uni above)
    auto emit8 = [&](int m, int n, int ncols, float (&v)[8], bool add_bias, bool add_rb, bool add_res) {
      if (m >= p.M) return;
      const int nvalid = min(8, ncols - n);
      if (nvalid <= 0) return;
      if (add_bias && p.bias != nullptr) {
        if (nvalid == 8) {
          const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < nvalid) v[e] += p.bias[n + e];
        }
      }
      if (add_rb && p.rowbias != nullptr) {
        const float* rbp = p.rowbias + ((int64_t)m / p.rows_per_group) * p.ld_rowbias + n;
        if (nvalid == 8 && ((p.ld_rowbias & 3) == 0) && (((uintptr_t)p.rowbias & 15) == 0)) {
          const float4 b0 = *(const float4*)rbp, b1 = *(const float4*)(rbp + 4);
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < nvalid) v[e] += rbp[e];
        }
      }
      if (add_res && p.residual != nullptr) {
        const f16* rp = (const f16*)p.residual + (int64_t)m * p.ldr + n;
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
      const int64_t o = p.head_dim > 0 ? ((int64_t)(n / p.head_dim) * p.M + m) * p.head_dim + n % p.head_dim
                                       : obatch + (int64_t)m * p.ldo + n;
      const int64_t ldo_eff = p.head_dim > 0 ? p.head_dim : p.ldo;
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
        if (nvalid == 8 && ((ldo_eff & 7) == 0)) {
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
    };
    // 4 consecutive output columns (the unpaired last tile when BN/32 is odd)
    auto emit4 = [&](int m, int n, float (&v)[4], bool add_bias, bool add_rb) {
      if (m >= p.M) return;
      const int nvalid = min(4, p.N - n);
      if (nvalid <= 0) return;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (e < nvalid) {
          if (add_bias && p.bias != nullptr) v[e] += p.bias[n + e];
          if (add_rb && p.rowbias != nullptr) v[e] += p.rowbias[((int64_t)m / p.rows_per_group) * p.ld_rowbias + n + e];
        }
      }
      if (p.residual != nullptr) {
        const f16* rp = (const f16*)p.residual + (int64_t)m * p.ldr + n;
        if (nvalid == 4 && ((p.ldr & 3) == 0)) {
          union { u32x2 u; f16 e[4]; } t;
          t.u = *(const u32x2*)rp;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)t.e[e];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) v[e] += (float)rp[e];
        }
      }
      const int64_t o = p.head_dim > 0 ? ((int64_t)(n / p.head_dim) * p.M + m) * p.head_dim + n % p.head_dim
                                       : obatch + (int64_t)m * p.ldo + n;
      const int64_t ldo_eff = p.head_dim > 0 ? p.head_dim : p.ldo;
      if (p.out_f32) {
        float* op = (float*)p.out + o;
        if (nvalid == 4 && ((p.ldo & 3) == 0)) *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) op[e] = v[e];
        }
      } else {
        f16* op = (f16*)p.out + o;
        if (nvalid == 4 && ((ldo_eff & 3) == 0)) {
          union { u32x2 u; f16 e[4]; } t;
#pragma unroll
          for (int e = 0; e < 4; ++e) t.e[e] = (f16)v[e];
          *(u32x2*)op = t.u;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < nvalid) op[e] = (f16)v[e];
        }
      }
    };

    if (TRANS) {
      // out[n][m], m contiguous: pair the tiles (i, i+1) along M; bias only (checked by the launcher)
      float bt[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {       // all bias loads in front of the first store
        const int n = n0 + tile_c(j) + fr;
        bt[j] = (p.bias != nullptr && n < p.N) ? p.bias[n] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int n = n0 + tile_c(j) + fr;
        const float bn_ = bt[j];
#pragma unroll
        for (int ip = 0; ip < FM / 2; ++ip) {
          float x[4], y[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            x[r] = acc[2 * ip][j][r] * alpha + bn_;
            y[r] = acc[2 * ip + 1][j][r] * alpha + bn_;
            row_swap(x[r], y[r]);
          }
          const int m = m0 + wm * WTM + (2 * ip + tsel) * 16 + csel;
          if (n < p.N && m < p.M) {
            f16* op = (f16*)p.out + obatch + (int64_t)n * p.ldo + m;
            if (m + 8 <= p.M && ((p.ldo & 7) == 0)) {
              U4H8 t;
#pragma unroll
              for (int r = 0; r < 4; ++r) { t.e[r] = (f16)x[r]; t.e[4 + r] = (f16)y[r]; }
              *(u32x4*)op = t.u;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (m + r < p.M) op[r] = (f16)x[r];
                if (m + 4 + r < p.M) op[4 + r] = (f16)y[r];
              }
            }
          }
        }
      }
    } else if (p.act == 1) {
      // GEGLU: packed columns per 32 = [16 x value | 16 x gate] -> tiles (2t, 2t+1) of a wave are the value / gate
      // of the same 16 output columns; out column = packed column / 2
      if (NB == 4) {
        const int pn = n0 + wn * WTN + fq * 4;          // packed column of acc[i][0][0]
        float bv0[4], bg0[4], bv1[4], bg1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool hb = p.bias != nullptr && !bias_in_acc;
          bv0[r] = (hb && pn + r < p.N) ? p.bias[pn + r] : 0.f;
          bg0[r] = (hb && pn + 16 + r < p.N) ? p.bias[pn + 16 + r] : 0.f;
          bv1[r] = (hb && pn + 32 + r < p.N) ? p.bias[pn + 32 + r] : 0.f;
          bg1[r] = (hb && pn + 48 + r < p.N) ? p.bias[pn + 48 + r] : 0.f;
        }
        const int ocol = (n0 + wn * WTN) / 2 + tsel * 16 + csel;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float h0 = (acc[i][0][r] * alpha + bv0[r]) * ANIP_GELU(acc[i][1][r] * alpha + bg0[r]);
            float h1 = (acc[i][2][r] * alpha + bv1[r]) * ANIP_GELU(acc[i][3][r] * alpha + bg1[r]);
            row_swap(h0, h1);
            v[r] = h0;
            v[4 + r] = h1;
          }
          emit8(m0 + wm * WTM + i * 16 + fr, ocol, p.N / 2, v, false, false, false);
        }
      }
    } else {
      // A block whose tile lies fully inside the output and whose operands allow 16-B accesses (every hot shape of the
      // path) takes the tight epilogue below; ragged tiles take the general per-element-guarded one.
      const bool has_rb = p.rowbias != nullptr, has_res = p.residual != nullptr;
      // (32-bit per-lane byte offsets: the output / residual extents must stay below 4 GiB)
      const bool tight = m0 + BM2 <= p.M && n0 + BN <= p.N &&
                         (p.out_f32 ? (p.ldo & 3) == 0 : ((p.head_dim > 0 ? p.head_dim : p.ldo) & 7) == 0) &&
                         (!has_res || (p.ldr & 7) == 0) &&
                         (bias_in_acc || (p.bias == nullptr && !has_rb)) &&
                         (int64_t)p.M * p.ldo * (p.out_f32 ? 4 : 2) < (1ll << 32) &&
                         (!has_res || (int64_t)p.M * p.ldr * 2 < (1ll << 32));
      if (tight && alpha != 1.0f) {      // (acc_has_bias implies alpha == 1: nothing but products in the accumulators here)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] *= alpha;
      }
      if (!tight) {
        // general path.  The accumulators may ALREADY hold the bias (and a block-uniform row-group bias): a full tile with
        // 16-B accessible bias terms takes `acc_has_bias` whether or not the rest of the tight conditions hold (output of
        // 4 GiB or more, unaligned leading dimensions).  Round 2 added them a second time here — the decoded frames of a
        // 16-frame VAE batch at 512x512 / 768x768 (2^30+ output elements in the upsampler convs) carried every channel's
        // bias twice: the "46 dB at 512x512, 38 dB at 768x768" of the round-2 / round-3 parity runs.
        const bool add_bias = !bias_in_acc, add_rb = !(bias_in_acc && rb_in_acc);
#pragma unroll
        for (int jp = 0; jp < NB / 2; ++jp) {
          const int n = n0 + tile_c(2 * jp) + tsel * 16 + csel;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float x = acc[i][2 * jp][r] * alpha, y = acc[i][2 * jp + 1][r] * alpha;
              row_swap(x, y);
              v[r] = x;
              v[4 + r] = y;
            }
            emit8(m0 + wm * WTM + i * 16 + fr, n, p.N, v, add_bias, add_rb, true);
          }
        }
        if (NB & 1) {
          const int n = n0 + tile_c(NB - 1) + fq * 4;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][NB - 1][r] * alpha;
            emit4(m0 + wm * WTM + i * 16 + fr, n, v, add_bias, add_rb);
          }
        }
      } else {
        // Residual vectors in batches AHEAD of the stores.  `out` may alias anything as far as the compiler knows, so a
        // load written after a store stays after it: fetched where they are used (the general path), every (row block,
        // column pair) pays a dependent L2 / HBM round trip behind the previous stores — FM * NB/2 of them per wave with
        // one KiB in flight each, which capped the K = 320 / 640 layers at ~2.5 TB/s.  Here RBAT residual vectors of a
        // column pair are in flight together before the first of their stores; bias and block-uniform row-group bias
        // are already inside the accumulators.  (RBAT: what the 128-VGPR budget of the 4-waves-per-SIMD tiles allows.)
        constexpr int RBAT = (NW == 8 && WNW == 2 && NB == 5) ? 2 : FM;   // 256 x 160, 4 waves per SIMD: 128 VGPRs
        constexpr bool QUAD = (NB == 4 || NB == 5);
        // addressing: wave-uniform 64-bit bases (+ the uniform 16-row step) in SGPRs, one 32-bit per-lane BYTE offset —
        // 64-bit per-lane pointers for every (row block, column pair) do not fit the 128-VGPR budget next to the
        // accumulators
        const int mrow = m0 + wm * WTM + fr;                 // row of acc[0][.]; tile i is 16 rows further down
        const uint32_t esz = p.out_f32 ? 4u : 2u;
        const bool hm = p.head_dim > 0;      // head-major output: row step = head_dim, plus a per-column head offset
        const uint32_t ldr_b = (uint32_t)p.ldr * 2u, ldo_b = (uint32_t)(hm ? p.head_dim : p.ldo) * esz;
        const char* resb = (const char*)p.residual;
        char* outb = (char*)p.out + obatch * (int64_t)esz;
        const uint32_t rrow = (uint32_t)mrow * ldr_b, orow = (uint32_t)mrow * ldo_b;
        const bool rb_row = has_rb && !rb_in_acc;            // row-group bias that changes inside the block (rare)
        // Full-line stores (round 3).  A wave's first four 16-column tiles are 64 consecutive columns = one 128-B line of
        // fp16 per row, but after the pair swap a store instruction covers 16 rows x 64 B (4 lanes per row): half lines.
        // Measured with the same block tiles and nothing but the stores (tools/exp_store_pattern.py, 268 MB): 4.8 TB/s with
        // 64-B row segments, 6.9 TB/s with 128-B ones; with a residual read of the same shape 3.8 vs 5.1 TB/s.  So the two
        // pairs of a 16-row block trade halves across lanes fr <-> fr ^ 8 (one v_mov_dpp row_ror:8 per dword, the bank mask
        // doing the select): store A = rows 0-7 of the block, store B = rows 8-15, each lane 16 B, 8 lanes = 128 B per row.
        //   lane (fr < 8):  A <- own pair 0 of row fr          B <- pair 0 of row fr + 8 (from lane fr + 8)
        //   lane (fr >= 8): A <- pair 1 of row fr - 8 (lane fr - 8)   B <- own pair 1 of row fr
        // so a lane's column is the same in A and B: pair (fr >> 3), and the residual is fetched in the same two shapes.
        if constexpr (QUAD) {
          constexpr int RBQ = (NW == 8 && WNW == 2 && NB == 5) ? 1 : (NB == 5 ? 2 : FM / 2);   // 16-row blocks per residual batch (2 vectors each)
          const int r8 = fr & 7;
          const int n = n0 + tile_c(0) + (fr >> 3) * 32 + tsel * 16 + csel;
          const int mq = m0 + wm * WTM + r8;                                // row of store A of block 0; B: + 8
          const uint32_t rn = (uint32_t)mq * ldr_b + (uint32_t)n * 2u;
          const uint32_t on = (uint32_t)mq * ldo_b + (hm ? ((uint32_t)(n / p.head_dim) * (uint32_t)p.M * (uint32_t)p.head_dim +
                                                           (uint32_t)(n % p.head_dim)) : (uint32_t)n) * esz;
          // (The residual sub-tile fetched by LDS-DMA into the vacated operand ring, all of it in flight at once, measured + 5 %
          //  cache-cold on the N = K = 320 residual layers and nothing in the pipeline, profiles/r03/zk_*: not kept.)
#pragma unroll
          for (int ib = 0; ib < FM; ib += RBQ) {
            U4H8 resA[RBQ], resB[RBQ];
            if (has_res) {
#pragma unroll
              for (int i = 0; i < RBQ; ++i) {
                resA[i].u = *(const u32x4*)(resb + (size_t)((ib + i) * 16) * ldr_b + rn);
                resB[i].u = *(const u32x4*)(resb + (size_t)((ib + i) * 16 + 8) * ldr_b + rn);
              }
            } else if (PHASED) {   // (defined either way: an undefined value becomes loop-carried state of the tile loop)
#pragma unroll
              for (int i = 0; i < RBQ; ++i) {
                resA[i].u = u32x4{0u, 0u, 0u, 0u};
                resB[i].u = u32x4{0u, 0u, 0u, 0u};
              }
            }
#pragma unroll
            for (int ii = 0; ii < RBQ; ++ii) {
              const int i = ib + ii;
              float va[8], vb[8];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float x0 = acc[i][0][r], x1 = acc[i][1][r], y0 = acc[i][2][r], y1 = acc[i][3][r];
                row_swap(x0, x1);            // pair 0: 8 consecutive columns (x0 | x1) of row fr
                row_swap(y0, y1);            // pair 1
                va[r] = half_swap_hi(x0, y0);
                va[4 + r] = half_swap_hi(x1, y1);
                vb[r] = half_swap_lo(y0, x0);
                vb[4 + r] = half_swap_lo(y1, x1);
              }
              if (rb_row) {
                const float* ra = p.rowbias + ((int64_t)(mq + i * 16) / p.rows_per_group) * p.ld_rowbias + n;
                const float* rb_ = p.rowbias + ((int64_t)(mq + i * 16 + 8) / p.rows_per_group) * p.ld_rowbias + n;
                const float4 a0 = *(const float4*)ra, a1 = *(const float4*)(ra + 4);
                const float4 b0 = *(const float4*)rb_, b1 = *(const float4*)(rb_ + 4);
                va[0] += a0.x; va[1] += a0.y; va[2] += a0.z; va[3] += a0.w;
                va[4] += a1.x; va[5] += a1.y; va[6] += a1.z; va[7] += a1.w;
                vb[0] += b0.x; vb[1] += b0.y; vb[2] += b0.z; vb[3] += b0.w;
                vb[4] += b1.x; vb[5] += b1.y; vb[6] += b1.z; vb[7] += b1.w;
              }
              if (has_res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  va[e] += (float)resA[ii].e[e];
                  vb[e] += (float)resB[ii].e[e];
                }
              }
              char* opa = outb + (size_t)(i * 16) * ldo_b + on;
              char* opb = outb + (size_t)(i * 16 + 8) * ldo_b + on;
              if (p.out_f32) {
                *(float4*)opa = make_float4(va[0], va[1], va[2], va[3]);
                *(float4*)(opa + 16) = make_float4(va[4], va[5], va[6], va[7]);
                *(float4*)opb = make_float4(vb[0], vb[1], vb[2], vb[3]);
                *(float4*)(opb + 16) = make_float4(vb[4], vb[5], vb[6], vb[7]);
              } else {
                U4H8 ta, tb;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  ta.e[e] = (f16)va[e];
                  tb.e[e] = (f16)vb[e];
                }
                *(u32x4*)opa = ta.u;
                *(u32x4*)opb = tb.u;
              }
            }
          }
        } else {
#pragma unroll
          for (int jp = 0; jp < NB / 2; ++jp) {
            const int n = n0 + tile_c(2 * jp) + tsel * 16 + csel;
            const uint32_t rn = rrow + (uint32_t)n * 2u;
            const uint32_t on = orow + (hm ? ((uint32_t)(n / p.head_dim) * (uint32_t)p.M * (uint32_t)p.head_dim +
                                              (uint32_t)(n % p.head_dim)) : (uint32_t)n) * esz;
#pragma unroll
            for (int ib = 0; ib < FM; ib += RBAT) {
              U4H8 res[RBAT];
              if (has_res) {
#pragma unroll
                for (int i = 0; i < RBAT; ++i) res[i].u = *(const u32x4*)(resb + (size_t)((ib + i) * 16) * ldr_b + rn);
              } else if (PHASED) {   // (defined either way: an undefined value becomes loop-carried state of the tile loop)
#pragma unroll
                for (int i = 0; i < RBAT; ++i) res[i].u = u32x4{0u, 0u, 0u, 0u};
              }
#pragma unroll
              for (int ii = 0; ii < RBAT; ++ii) {
                const int i = ib + ii;
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float x = acc[i][2 * jp][r], y = acc[i][2 * jp + 1][r];
                  row_swap(x, y);
                  v[r] = x;
                  v[4 + r] = y;
                }
                if (rb_row) {
                  const float* rbp = p.rowbias + ((int64_t)(mrow + i * 16) / p.rows_per_group) * p.ld_rowbias + n;
                  const float4 b0 = *(const float4*)rbp, b1 = *(const float4*)(rbp + 4);
                  v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
                  v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
                }
                if (has_res) {
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] += (float)res[ii].e[e];
                }
                char* op = outb + (size_t)(i * 16) * ldo_b + on;
                if (p.out_f32) {
                  *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
                  *(float4*)(op + 16) = make_float4(v[4], v[5], v[6], v[7]);
                } else {
                  U4H8 t;
#pragma unroll
                  for (int e = 0; e < 8; ++e) t.e[e] = (f16)v[e];
                  *(u32x4*)op = t.u;
                }
              }
            }
          }
        }
        if (NB & 1) {
          const int n = n0 + tile_c(NB - 1) + fq * 4;
          const uint32_t rn = rrow + (uint32_t)n * 2u;
          const uint32_t on = orow + (hm ? ((uint32_t)(n / p.head_dim) * (uint32_t)p.M * (uint32_t)p.head_dim +
                                            (uint32_t)(n % p.head_dim)) : (uint32_t)n) * esz;
          union H4 { u32x2 u; f16 e[4]; };
#pragma unroll
          for (int ib = 0; ib < FM; ib += RBAT) {
            H4 res[RBAT];
            if (has_res) {
#pragma unroll
              for (int i = 0; i < RBAT; ++i) res[i].u = *(const u32x2*)(resb + (size_t)((ib + i) * 16) * ldr_b + rn);
            } else if (PHASED) {
#pragma unroll
              for (int i = 0; i < RBAT; ++i) res[i].u = u32x2{0u, 0u};
            }
#pragma unroll
            for (int ii = 0; ii < RBAT; ++ii) {
              const int i = ib + ii;
              float v[4] = {acc[i][NB - 1][0], acc[i][NB - 1][1], acc[i][NB - 1][2], acc[i][NB - 1][3]};
              if (rb_row) {
                const float4 b = *(const float4*)(p.rowbias + ((int64_t)(mrow + i * 16) / p.rows_per_group) * p.ld_rowbias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
              }
              if (has_res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)res[ii].e[e];
              }
              char* op = outb + (size_t)(i * 16) * ldo_b + on;
              if (p.out_f32) {
                *(float4*)op = make_float4(v[0], v[1], v[2], v[3]);
              } else {
                H4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t.e[e] = (f16)v[e];
                *(u32x2*)op = t.u;
              }
            }
          }
        }
      }
    }
    if (!has_next) break;
    vb = vbn;
    m0 = m0n;
    n0 = n0n;
  }
#undef ANIP_G2_DMA_A
#undef ANIP_G2_DMA_B
#undef ANIP_G2_BAR_RAW
#undef ANIP_G2_BAR_L
#undef ANIP_G2_BAR_C
#undef ANIP_G2_MMA
}

inline int gemm2_cu_count() { return anip_cu_count(); }   // per device ordinal (api.cpp): partitions of one node may differ

template <int BM2, int BN, int NW, int WNW, int BKT, int NST, bool CONV, bool TRANS>
int launch_gemm2(const anip_gemm_params& p, hipStream_t stream, int splitk = 1) {
  constexpr int NT2 = NW * 64;
  constexpr int LDS = NST * (BM2 + BN) * BKT * 2;
  {
    auto kfn = gemm2_kernel<BM2, BN, NW, WNW, BKT, NST, CONV, TRANS>;
    if (anip_raise_lds_limit((const void*)kfn, LDS) != 0) {
      anip_set_error("anip_gemm: cannot raise the dynamic LDS limit to %d bytes", LDS);
      return -2;
    }
  }
  const int nbm = (p.M + BM2 - 1) / BM2, nbn = (p.N + BN - 1) / BN;
  unsigned grid = (unsigned)(nbm * nbn);
  // Persistent form of the quarter-phased wide tiles (one workgroup per CU: 147 KB of LDS): a launch with more tiles than
  // CUs starts one workgroup per CU, and each walks tiles blockIdx.x, + gridDim.x, ... with the first K-tile of the next
  // tile staged under the epilogue of the running one (see the tile loop in the kernel).
  if (NST == 2 && BKT == 64 && NW == 8 && p.batch <= 1 && splitk <= 1 && (p.K + BKT - 1) / BKT >= 2) {
    const unsigned ncu = (unsigned)gemm2_cu_count();
    if (ncu >= 8 && (ncu & 7) == 0 && grid > ncu) grid = ncu;
  }
  hipLaunchKernelGGL((gemm2_kernel<BM2, BN, NW, WNW, BKT, NST, CONV, TRANS>), dim3(grid, (unsigned)p.batch, 1),
                     dim3(NT2), LDS, stream, p, 0, splitk);
  return 1;
}

template <int BM2, int BN, int NW, int WNW, int BKT, int NST>
int dispatch_gemm2(const anip_gemm_params& p, hipStream_t stream, int splitk = 1) {
  if (p.conv) return launch_gemm2<BM2, BN, NW, WNW, BKT, NST, true, false>(p, stream, splitk);
  if (p.trans_out) return launch_gemm2<BM2, BN, NW, WNW, BKT, NST, false, true>(p, stream, splitk);
  return launch_gemm2<BM2, BN, NW, WNW, BKT, NST, false, false>(p, stream, splitk);
}

// ---------------------------------------------------------------------------------------------------------
// split-K second pass: out = epilogue(alpha * sum_s ws[s]) with bias / row-group bias / residual, 4 columns per thread
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, const anip_gemm_params p) {
  const int64_t MN = (int64_t)p.M * p.N;
  const int64_t nq = MN >> 2;
  const int nq_row = p.N >> 2;
  for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < nq; q += (int64_t)gridDim.x * 256) {
    const int m = (int)(q / nq_row), n = (int)(q - (int64_t)m * nq_row) * 4;
    float4 a = *(const float4*)(ws + q * 4);
    for (int s_ = 1; s_ < S; ++s_) {
      const float4 b = *(const float4*)(ws + (int64_t)s_ * MN + q * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
    if (p.bias != nullptr) {
      const float4 b = *(const float4*)(p.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (p.rowbias != nullptr) {
      const float* rb = p.rowbias + ((int64_t)m / p.rows_per_group) * p.ld_rowbias + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += rb[e];
    }
    if (p.residual != nullptr) {
      const f16* rp = (const f16*)p.residual + (int64_t)m * p.ldr + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += (float)rp[e];
    }
    if (p.out_f32) {
      float* op = (float*)p.out + (int64_t)m * p.ldo + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) op[e] = v[e];
    } else {
      f16* op = (f16*)p.out + (p.head_dim > 0 ? ((int64_t)(n / p.head_dim) * p.M + m) * p.head_dim + n % p.head_dim
                                              : (int64_t)m * p.ldo + n);
      if (((p.head_dim > 0 ? p.head_dim : p.ldo) & 3) == 0 && (((uintptr_t)p.out) & 7) == 0) {
        union { u32x2 u; f16 e[4]; } t;
#pragma unroll
        for (int e = 0; e < 4; ++e) t.e[e] = (f16)v[e];
        *(u32x2*)op = t.u;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) op[e] = (f16)v[e];
      }
    }
  }
}

}  // namespace

// split factor for problems with too few tiles to fill the chip (1 = no split); *cfg = tile configuration:
// 128 / 160: 128-row 4-wave tiles of that width; 256 / 320: the wide 256-row tiles of that width
static int gemm2_split(const anip_gemm_params& p, int* cfg) {
  if (p.batch > 1 || p.act == 1 || p.trans_out || p.M < 64 || p.M > 16384 || (p.N & 3) != 0) return 1;
  if (p.conv ? (p.Cin % 32 != 0) : (p.A2 != nullptr && (p.K1 % 32) != 0)) return 1;
  if ((((uintptr_t)p.bias | (uintptr_t)p.rowbias) & 15) != 0) return 1;
  if (p.M < 1024) {
    // A handful of 128-row tiles under a long K: the once-per-clip ReferenceNet at its 8x8 / 16x16 levels (M = 128 / 512 for
    // the CFG pair of one reference frame; its 3x3 convolutions stream 29-59 MB of weights through 10 workgroups of the
    // small-problem kernel: 270 / 530 us -> 30 / 36 us split, 16x16: 273 / 538 -> 48 / 80, profiles/r04/u_*).  Up to 32 slices, at least 256 deep, until the launch has one block per CU.
    if (p.K < 1024) return 1;
    const int64_t pad128 = (int64_t)((p.N + 127) / 128) * 128, pad160 = (int64_t)((p.N + 159) / 160) * 160;
    const int bn = pad160 < pad128 ? 160 : 128;
    const int64_t tiles = (int64_t)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
    if (tiles > 96) return 1;
    const int nk = (p.K + 31) / 32;
    int S = (int)min((int64_t)32, (256 + tiles - 1) / tiles);
    S = min(S, nk / 8);
    if (S < 2) return 1;
    *cfg = bn;
    const int per = (nk + S - 1) / S;
    return (nk + per - 1) / per;
  }
  // experiment knobs: smallest M that takes the wide-tile split (default 2048: the 8x8 level), most slices (default 8)
  constexpr int wsplit_min_m = 2048;
  constexpr int wsplit_max_s = 8;
  if (p.M >= wsplit_min_m) {
    // M >= 2048 (the 8x8 and 16x16 levels): 2..8 slices of the WIDE tiles when K is long (3x3 convs, ff-out: K >= 4096) and
    // the tiles alone leave half the CUs idle; shorter K: no split at all.  (Round 3, one call, same box: the 8x8 convs on
    // 4 slices of 128 x 160 tiles 105 / 179 us, on 8 slices of 256 x 320 tiles 88 / 131 us — half the weight re-reads per
    // row tile; M = 2048, N = K = 1280 on 2 slices + reduce 30.1 us, unsplit 25.6 us.)
    const bool k64 = p.conv ? (p.Cin % 64 == 0) : (p.A2 == nullptr || p.K1 % 64 == 0);
    if (!k64 || p.K < 4096) return 1;
    const int64_t pad320 = (int64_t)((p.N + 319) / 320) * 320, pad256 = (int64_t)((p.N + 255) / 256) * 256;
    int wbn = 0;
    if (pad320 <= pad256 && pad320 * 100 <= (int64_t)p.N * 115) wbn = 320;
    else if (pad256 * 100 <= (int64_t)p.N * 115) wbn = 256;
    if (wbn == 0) return 1;
    const int64_t tiles = (int64_t)((p.M + 255) / 256) * ((p.N + wbn - 1) / wbn);
    if (tiles >= 192) return 1;
    const int nk = (p.K + 63) / 64;
    int S = (int)min((int64_t)wsplit_max_s, (256 + tiles - 1) / tiles);
    S = min(S, nk / 16);                     // slices at least 1024 deep
    if (S < 2) return 1;
    *cfg = wbn;
    const int per = (nk + S - 1) / S;
    return (nk + per - 1) / per;
  }
  const int64_t pad128 = (int64_t)((p.N + 127) / 128) * 128, pad160 = (int64_t)((p.N + 159) / 160) * 160;
  const int bn = pad160 < pad128 ? 160 : 128;
  *cfg = bn;
  const int64_t tiles = (int64_t)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
  if (tiles * 2 < 128 || tiles >= 320) return 1;
  const int nk = (p.K + 31) / 32;
  constexpr int mink = 16;
  int S = (int)min((int64_t)8, (512 + tiles - 1) / tiles);
  S = min(S, nk / mink);                     // slices at least 512 deep
  if (S < 2) return 1;
  const int per = (nk + S - 1) / S;
  return (nk + per - 1) / per;               // every slice non-empty
}

int64_t anip_gemm2_workspace_bytes(const anip_gemm_params& p) {
  int bn;
  const int S = gemm2_split(p, &bn);
  return S > 1 ? (int64_t)S * p.M * p.N * 4 : 0;
}

// split-K path: 1 if launched (partials + reduce), 0 if the problem is not split, < 0 on error
int anip_gemm2_try(const anip_gemm_params& p, hipStream_t stream);
int anip_gemm2_try_splitk(const anip_gemm_params& p, hipStream_t stream) {
  int bn = 128;
  const int S = gemm2_split(p, &bn);
  if (S <= 1) return 0;
  const int64_t need = (int64_t)S * p.M * p.N * 4;
  if (p.workspace == nullptr || p.workspace_bytes < need || (((uintptr_t)p.workspace) & 15) != 0) {
    anip_set_error("anip_gemm: this problem is split over K and needs %lld bytes of 16-B aligned workspace "
                   "(anip_gemm_workspace_bytes); got %lld", (long long)need, (long long)p.workspace_bytes);
    return -1;
  }
  anip_gemm_params q = p;
  q.out = p.workspace; q.ldo = p.N; q.out_f32 = 1;
  q.alpha = 1.0f; q.bias = nullptr; q.rowbias = nullptr; q.residual = nullptr;
  q.head_dim = 0;            // partial tiles are plain [split][M][N]; the reduce kernel applies the output mapping
  q.batch = S; q.strideA = 0; q.strideW = 0; q.strideO = (int64_t)p.M * p.N;
  int rc;
  if (bn == 128) rc = dispatch_gemm2<128, 128, 4, 2, 32, 3>(q, stream, S);
  else if (bn == 160) rc = dispatch_gemm2<128, 160, 4, 2, 32, 3>(q, stream, S);
  else if (bn == 256) rc = dispatch_gemm2<256, 256, 8, 4, 64, 2>(q, stream, S);
  else rc = dispatch_gemm2<256, 320, 8, 4, 64, 2>(q, stream, S);
  if (rc < 0) return rc;
  const int64_t nq = ((int64_t)p.M * p.N) >> 2;
  const unsigned blocks = (unsigned)min((int64_t)4096, (nq + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)p.workspace, S, p);
  return 1;
}

// returns 1 if the problem was launched on gemm2, 0 if it is not eligible (caller falls back to the
// small-problem kernel), < 0 on error
int anip_gemm2_try(const anip_gemm_params& p, hipStream_t stream) {
  if (p.M < 1024) return 0;
  if (p.conv) {
    if (p.Cin % 32 != 0 || p.trans_out) return 0;
  } else {
    if (p.A2 != nullptr && (p.K1 % 32) != 0) return 0;
  }
  if (p.trans_out && (p.act == 1 || p.out_f32 || p.rowbias || p.residual)) return 0;
  // the vector paths of the epilogue assume 16-B aligned bases
  if ((((uintptr_t)p.out | (uintptr_t)p.bias | (uintptr_t)p.residual) & 15) != 0) return 0;
  if (p.batch > 1 && (p.strideO & 7) != 0) return 0;
  const int64_t nb = p.batch > 1 ? p.batch : 1;
  const int64_t mt256 = (p.M + 255) / 256;

  // wide tiles (256 x 320 / 256 x 256, BK = 64): every K-tile inside one conv tap / one A source, N padded < 15 %,
  // and K long enough that the main loop (not the per-tile prologue / epilogue, where two resident blocks per CU
  // overlap better) dominates — measured crossover between K = 320 and K = 640
  const bool k64 = p.conv ? (p.Cin % 64 == 0) : (p.A2 == nullptr || p.K1 % 64 == 0);
  // (round 2 microbench, 32 frames: K = 640 layers of the 32x32 level 10-12 % faster on the wide tiles — ff-in GEGLU
  // 307 -> 272 us, temporal qkv 122 -> 107, out-proj 59.9 -> 54.0; K = 320 layers no better, some worse)
  // (with the 64-B aligned tile deal the wide tiles also win on the K = 320 layers whose output is at least two of
  // their tiles wide — temporal qkv N = 960: 181 -> 155 us; round 4, with the persistent walk and the whole-line epilogue in:
  // the N = K = 320 layers of the 64x64 level, too — warm the same, cache-cold 97-107 -> 91 us (+ residual), 68 -> 60, 65 -> 58;
  // whole clip 1302.5 -> 1298.4 ms in one call, profiles/r04/w_*)
  if (k64 && (p.K >= 640 || (p.K >= 256 && p.N >= 320))) {
    const int64_t pad320 = (int64_t)((p.N + 319) / 320) * 320, pad256 = (int64_t)((p.N + 255) / 256) * 256;
    int wbn = 0;
    if (p.act == 1) wbn = (pad256 * 100 <= (int64_t)p.N * 115) ? 256 : 0;   // GEGLU pairs tiles: even count per wave
    else if (pad320 <= pad256 && pad320 * 100 <= (int64_t)p.N * 115) wbn = 320;
    else if (pad256 * 100 <= (int64_t)p.N * 115) wbn = 256;
    constexpr int wide_min = 192;
    if (wbn != 0 && mt256 * ((p.N + wbn - 1) / wbn) * nb >= wide_min)      // >= 3/4 of the CUs busy
      return wbn == 320 ? dispatch_gemm2<256, 320, 8, 4, 64, 2>(p, stream) : dispatch_gemm2<256, 256, 8, 4, 64, 2>(p, stream);
  }
  int bn = 128;
  if (p.act != 1) {
    const int64_t pad128 = (int64_t)((p.N + 127) / 128) * 128, pad160 = (int64_t)((p.N + 159) / 160) * 160;
    if (pad160 < pad128) bn = 160;
  }
  const int64_t tiles256 = mt256 * ((p.N + bn - 1) / bn) * nb;
  if (tiles256 * 2 < 128) return 0;
  constexpr int big_min = 1024;
  const bool big = tiles256 >= big_min;   // >= 2 full rounds of 2 x 256 resident 256-row blocks
  if (big) return bn == 128 ? dispatch_gemm2<256, 128, 8, 2, 32, 3>(p, stream) : dispatch_gemm2<256, 160, 8, 2, 32, 3>(p, stream);
  // At most one 128-row tile per CU (the 8x8 level, M = 2048): 64-deep K-tiles and EIGHT waves per tile (wave tile 32 x 64 /
  // 32 x 80).  These launches are bound by what one wave per SIMD can do in order — issue its share of the LDS-DMA, read its
  // fragments, run its MFMAs — not by LDS space (96 / 108 KB: one block per CU) and not by the ring depth (a 4-stage ring:
  // no change).  Round 3, A/B inside one call each: M = 2048, N = K = 1280 25.6 us (4 waves, 32-deep) -> 22.1 (64-deep) ->
  // 19.1 (8 waves); the 2560 -> 1280 shortcut 42.2 -> 35.0 -> 30.9 us.  With 512 tiles (M = 8192: two to three blocks per
  // CU on the 32-deep tiles) the 64-deep tiles lose: 50.6 -> 61.8 us.
  const int64_t tiles128 = (int64_t)((p.M + 127) / 128) * ((p.N + bn - 1) / bn) * nb;
  if (k64 && p.K >= 512 && tiles128 <= 256)
    return bn == 128 ? dispatch_gemm2<128, 128, 8, 2, 64, 3>(p, stream) : dispatch_gemm2<128, 160, 8, 2, 64, 3>(p, stream);
  // N a multiple of both widths (1280 at the 16x16 level): the 160-wide tiles — 512 of them instead of 640, wave tile 64 x 80
  // (round 4, one call: M = 8192, N = K = 1280 + residual 52 / 66 us warm / cold -> 39 / 57; whole clip 1338.0 -> 1302.5 ms)
  if (bn == 128 && p.act != 1 && p.N % 160 == 0) bn = 160;
  return bn == 128 ? dispatch_gemm2<128, 128, 4, 2, 32, 3>(p, stream) : dispatch_gemm2<128, 160, 4, 2, 32, 3>(p, stream);
}
