// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4), fp16 in, fp32 accumulate.
//
//   out[M,N] = epilogue(alpha * A[M,K] @ W[N,K]^T)
//
// Tile 128(M) x 128(N) x 64(K), 256 threads = 4 waves (2x2), each wave a 64x64 sub-tile computed as
// 4x4 v_mfma_f32_16x16x32_f16 tiles (64 fp32 accumulators per lane).  A and W tiles are staged
// global -> registers -> LDS (double buffered, one barrier per K-tile, loads for tile t+1 in flight
// under the MFMAs of tile t).  LDS rows are 128 B (64 fp16); the 16-B k-group index is XOR-swizzled
// with (row>>1)&7 so that the per-fragment ds_read_b128 is bank-conflict free.
// The A loader is a gather: plain row-major (optionally two sources split along K = fused channel
// concat), or an NHWC 3x3 window (stride 1/2, zero padding, optional fused nearest-2x upsample).
// Epilogue: accumulators -> LDS (fp32) -> coalesced 16-B rows with bias / row-group bias / GEGLU /
// residual fused, single rounding to fp16 (or fp32 out).
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous run of tiles that share
// the same A row-panel.
#include "common.h"

// GEGLU activation of the epilogues: gelu_poly_f (FMA pipe only; default) or, with -DANIP_GELU_EXACT, gelu_fast_f (A&S erf,
// |error| <= 1.5e-7, two transcendentals)
#ifdef ANIP_GELU_EXACT
#define ANIP_GELU gelu_fast_f
#else
#define ANIP_GELU gelu_poly_f
#endif

int anip_gemm2_try(const anip_gemm_params& p, hipStream_t stream);  // gemm2.hip
int anip_gemm2_try_splitk(const anip_gemm_params& p, hipStream_t stream);
int64_t anip_gemm2_workspace_bytes(const anip_gemm_params& p);

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;
constexpr int A_TILE_BYTES = BM * BK * 2;  // 16384
constexpr int B_TILE_BYTES = BN * BK * 2;  // 16384
constexpr int SMEM_BYTES = 2 * (A_TILE_BYTES + B_TILE_BYTES);  // 65536
constexpr int CS = BN + 4;                 // fp32 epilogue staging row stride (floats)
static_assert(64 * CS * 4 <= SMEM_BYTES, "epilogue staging must fit");

template <bool CONV>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_small_kernel(const anip_gemm_params p) {
  __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- tile assignment (XCD-aware, bijective) -------------------------------------------------
  const int Ncols = p.N;
  const int nbm = (p.M + BM - 1) / BM, nbn = (Ncols + BN - 1) / BN;
  const int nblk = nbm * nbn;
  int swz;
  {
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bm = swz / nbn, bn = swz % nbn;
  const int m0 = bm * BM, n0 = bn * BN;

  const f16* __restrict__ A = (const f16*)p.A;
  const f16* __restrict__ A2 = (const f16*)p.A2;
  const f16* __restrict__ Wt = (const f16*)p.W;
  if (p.batch > 1) {
    A += (int64_t)blockIdx.y * p.strideA;
    Wt += (int64_t)blockIdx.y * p.strideW;
  }

  // ---- per-thread staging assignment -----------------------------------------------------------
  const int g = tid & 7;            // 16-B k-group inside the 64-wide K tile
  const int srow = tid >> 3;        // staged rows: srow + 32*i
  const int sw_w = (tid >> 4) & 7;  // ((srow + 32 i) >> 1) & 7, independent of i
  int st_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) st_off[i] = (srow + 32 * i) * 128 + ((g ^ sw_w) << 4);

  // per staged A row (fixed over the K loop): plain: row index; conv: image index * Hin * Win (pixels)
  // and the top-left input coordinate of the 3x3 window (after stride/pad)
  int64_t ra_base[4];
  int ra_y0[4], ra_x0[4];
  bool ra_valid[4];
  int64_t rb_base[4];
  bool rb_valid[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + srow + 32 * i;
    ra_valid[i] = m < p.M;
    if (CONV) {
      const int hw = p.Hout * p.Wout;
      const int mm = ra_valid[i] ? m : 0;
      const int img = mm / hw;
      const int rem = mm - img * hw;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      ra_base[i] = (int64_t)img * p.Hin * p.Win;
      ra_y0[i] = oy * p.stride - p.pad;
      ra_x0[i] = ox * p.stride - p.pad;
    } else {
      ra_base[i] = (int64_t)m;
      ra_y0[i] = 0;
      ra_x0[i] = 0;
    }
    const int n = n0 + srow + 32 * i;
    rb_valid[i] = n < Ncols;
    rb_base[i] = (int64_t)n * p.ldw;
  }

  u32x4 ra[4], rb[4];
  const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

  auto load_tiles = [&](int kt) {
    const int kk = kt * BK + g * 8;  // first k of this thread's 8-wide chunk
    const bool kval = kk < p.K;
    if (CONV) {
      const int k0 = kt * BK;
      // conv = 1: tap-major K; conv = 2: channel-block-major K [Cin/64][9 taps][64] (see hipops.pack_conv3x3)
      const int tap = p.conv == 2 ? (k0 % 576) >> 6 : k0 / p.Cin;    // uniform over the tile (Cin % 64 == 0, BK = 64)
      const int c = p.conv == 2 ? (k0 / 576) * 64 + g * 8 : kk - tap * p.Cin;
      const int dy = tap / 3, dx = tap - dy * 3;
      const int He = p.upsample ? 2 * p.Hin : p.Hin, We = p.upsample ? 2 * p.Win : p.Win;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int y = ra_y0[i] + dy, x = ra_x0[i] + dx;
        const bool ok = ra_valid[i] && kval && y >= 0 && y < He && x >= 0 && x < We;
        if (p.upsample) { y >>= 1; x >>= 1; }
        const int64_t off = ok ? (ra_base[i] + (int64_t)y * p.Win + x) * p.Cin + c : 0;
        const u32x4 v = *(const u32x4*)(A + off);
        ra[i] = ok ? v : zero4;
      }
    } else {
      const bool second = (A2 != nullptr) && (kk >= p.K1);
      const f16* src = second ? A2 : A;
      const int64_t ld = second ? p.lda2 : p.lda;
      const int kc = second ? kk - p.K1 : kk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = ra_valid[i] && kval;
        const u32x4 v = *(const u32x4*)(ok ? src + ra_base[i] * ld + kc : A);
        ra[i] = ok ? v : zero4;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = rb_valid[i] && kval;
      const u32x4 v = *(const u32x4*)(ok ? Wt + rb_base[i] + kk : Wt);
      rb[i] = ok ? v : zero4;
    }
  };
  auto store_tiles = [&](int buf) {
    char* sa = smem + buf * A_TILE_BYTES;
    char* sb = smem + 2 * A_TILE_BYTES + buf * B_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(u32x4*)(sa + st_off[i]) = ra[i];
      *(u32x4*)(sb + st_off[i]) = rb[i];
    }
  };

  // ---- fragment read offsets ---------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int sw_r = fr >> 1;  // ((tile_row0 + fr) >> 1) & 7 with tile_row0 % 16 == 0
  const int a_row_off = (wm * 64 + fr) * 128;
  const int b_row_off = (wn * 64 + fr) * 128;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = (kt + 1) < nk;
    if (more) load_tiles(kt + 1);
    const char* sa = smem + cur * A_TILE_BYTES;
    const char* sb = smem + 2 * A_TILE_BYTES + cur * B_TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int koff = (((ks * 4 + fq) ^ sw_r) << 4);
      f16x8 af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *(const f16x8*)(sa + a_row_off + t * 16 * 128 + koff);
        bf[t] = *(const f16x8*)(sb + b_row_off + t * 16 * 128 + koff);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: two passes of 64 rows through an fp32 LDS staging tile ---------------------------
  float* cs = (float*)smem;
  const float alpha = p.alpha;
  const bool geglu = p.act == 1;
  const int64_t obatch = (p.batch > 1) ? (int64_t)blockIdx.y * p.strideO : 0;
  for (int half = 0; half < 2; ++half) {
    if (wm == half) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            cs[(i * 16 + fq * 4 + r) * CS + wn * 64 + j * 16 + fr] = acc[i][j][r] * alpha;
    }
    __syncthreads();
    // 64 rows x 16 column-chunks of 8 (GEGLU: 8 output chunks per row)
    const int nchunk = geglu ? 8 : 16;
    for (int c = tid; c < 64 * nchunk; c += NTHREADS) {
      const int row = c / nchunk, cc = c - row * nchunk;
      const int m = m0 + half * 64 + row;
      if (m >= p.M) continue;
      float v[8];
      int ncol;        // first output column of this chunk
      int nvalid;      // number of valid output columns in the chunk
      if (geglu) {
        // packed columns per 32: [16 x value | 16 x gate]; output chunk cc = output columns cc*8..+7 of the tile
        const int pc = (cc >> 1) * 32 + (cc & 1) * 8;
        const float* hrow = cs + row * CS + pc;
        const float* grow = hrow + 16;
        const int pn = n0 + pc;      // packed column of the value; gate at +16
        ncol = bn * 64 + cc * 8;
        nvalid = min(8, p.N / 2 - ncol);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float hh = hrow[e], gg = grow[e];
          if (p.bias != nullptr && e < nvalid) { hh += p.bias[pn + e]; gg += p.bias[pn + 16 + e]; }
          v[e] = hh * ANIP_GELU(gg);
        }
      } else {
        const float* crow = cs + row * CS + cc * 8;
        ncol = n0 + cc * 8;
        nvalid = min(8, Ncols - ncol);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = crow[e];
        if (nvalid <= 0) continue;
        if (p.bias != nullptr) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < nvalid) v[e] += p.bias[ncol + e];
        }
        if (p.rowbias != nullptr) {
          const float* rbp = p.rowbias + ((int64_t)m / p.rows_per_group) * p.ld_rowbias + ncol;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e < nvalid) v[e] += rbp[e];
        }
        if (p.act == 2) {   // quick-GELU x * sigmoid(1.702 x) (CLIP's MLP, transformers `QuickGELUActivation`), ahead of the residual
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] / (1.0f + __expf(-1.702f * v[e]));
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
      }
      if (nvalid <= 0) continue;
      if (p.trans_out) {  // small problems only: element-wise transposed store
        f16* op = (f16*)p.out + obatch + (int64_t)ncol * p.ldo + m;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (e < nvalid) op[(int64_t)e * p.ldo] = (f16)v[e];
        continue;
      }
      const int64_t o = p.head_dim > 0 ? ((int64_t)(ncol / p.head_dim) * p.M + m) * p.head_dim + ncol % p.head_dim
                                       : obatch + (int64_t)m * p.ldo + ncol;
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
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int64_t anip_gemm_workspace_bytes(const anip_gemm_params* pp) {
  if (pp == nullptr || pp->M <= 0 || pp->N <= 0 || pp->K <= 0) return 0;
  anip_gemm_params p = *pp;
  if (p.batch < 1) p.batch = 1;
  const int64_t extA = p.conv ? (int64_t)p.Nimg * p.Hin * p.Win * p.Cin : (int64_t)p.M * p.lda;
  if (!(extA * 2 < 0xFFFF0000ll && (int64_t)p.N * p.ldw * 2 < 0xFFFF0000ll && (!p.A2 || (int64_t)p.M * p.lda2 * 2 < 0xFFFF0000ll)))
    return 0;
  if (p.act == 2) return 0;
  return anip_gemm2_workspace_bytes(p);
}

extern "C" int anip_gemm(const anip_gemm_params* pp, void* stream) {
  anip_gemm_params p = *pp;
  ANIP_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "anip_gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
  ANIP_REQUIRE(p.A && p.W && p.out, "anip_gemm: null pointer");
  ANIP_REQUIRE((p.K & 7) == 0, "anip_gemm: K=%d must be a multiple of 8", p.K);
  ANIP_REQUIRE((p.ldw & 7) == 0, "anip_gemm: ldw=%lld must be a multiple of 8", (long long)p.ldw);
  ANIP_REQUIRE(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.W & 15) == 0, "anip_gemm: A/W must be 16-B aligned");
  if (p.batch < 1) p.batch = 1;
  if (p.alpha == 0.0f) p.alpha = 1.0f;
  if (p.rowbias != nullptr) ANIP_REQUIRE(p.rows_per_group > 0, "anip_gemm: rows_per_group must be > 0");
  if (p.act == 1) {
    ANIP_REQUIRE((p.N % 128) == 0, "anip_gemm: GEGLU needs N %% 128 == 0 (N=%d)", p.N);  // whole 32-column [v|g] groups per tile
    ANIP_REQUIRE(p.residual == nullptr && p.rowbias == nullptr, "anip_gemm: GEGLU excludes residual/rowbias");
  }
  if (p.conv) {
    ANIP_REQUIRE(p.Cin % 64 == 0, "anip_gemm(conv): Cin=%d must be a multiple of 64", p.Cin);
    ANIP_REQUIRE(p.K == 9 * p.Cin, "anip_gemm(conv): K must equal 9*Cin");
    ANIP_REQUIRE(p.conv == 1 || p.conv == 2, "anip_gemm(conv): conv must be 1 (tap-major K) or 2 (channel-block-major K)");
    ANIP_REQUIRE((int64_t)p.Nimg * p.Hout * p.Wout == (int64_t)p.M, "anip_gemm(conv): M != Nimg*Hout*Wout");
    ANIP_REQUIRE(p.A2 == nullptr, "anip_gemm(conv): two-source A not supported");
    ANIP_REQUIRE(!p.upsample || (p.stride == 1 && p.pad == 1 && p.Hout == 2 * p.Hin && p.Wout == 2 * p.Win),
                 "anip_gemm(conv): upsample needs stride 1, pad 1, Hout=2Hin");
  } else {
    ANIP_REQUIRE((p.lda & 7) == 0, "anip_gemm: lda must be a multiple of 8");
    if (p.A2) {
      ANIP_REQUIRE((p.lda2 & 7) == 0 && (p.K1 & 7) == 0 && p.K1 > 0 && p.K1 < p.K &&
                       ((uintptr_t)p.A2 & 15) == 0,
                   "anip_gemm: bad two-source split K1=%d", p.K1);
    }
  }
  ANIP_REQUIRE(p.act >= 0 && p.act <= 2, "anip_gemm: act=%d (0 none, 1 GEGLU, 2 quick-GELU)", p.act);
  if (p.trans_out)
    ANIP_REQUIRE(p.act == 0 && !p.out_f32 && !p.rowbias && !p.residual && p.batch == 1,
                 "anip_gemm: trans_out supports bias only");
  if (p.head_dim != 0)
    ANIP_REQUIRE(p.head_dim > 0 && (p.head_dim & 7) == 0 && p.N % p.head_dim == 0 && !p.trans_out && !p.out_f32 && p.act == 0 &&
                     p.batch == 1 && !p.conv,
                 "anip_gemm: head-major output needs head_dim %% 8 == 0, N %% head_dim == 0, fp16, plain epilogue (head_dim=%d N=%d)",
                 p.head_dim, p.N);
  const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(nbm * nbn), (unsigned)p.batch, 1);
  {
    AnipProfScope prof_(p.conv ? ANIP_K_CONV3X3 : ANIP_K_GEMM, stream);
    // the main kernel (gemm2.hip) takes every problem large enough to fill the chip with 256-row tiles;
    // small / odd-shaped problems run on the 128x128 register-staged kernel below
    const int64_t extA = p.conv ? (int64_t)p.Nimg * p.Hin * p.Win * p.Cin : (int64_t)p.M * p.lda;
    int used = 0;
    // (act == 2, quick-GELU: the small-problem kernel's epilogue only — its one user is the CLIP tower's fc1 at M = 257 rows per image)
    if (p.act != 2 && extA * 2 < 0xFFFF0000ll && (int64_t)p.N * p.ldw * 2 < 0xFFFF0000ll &&
        (!p.A2 || (int64_t)p.M * p.lda2 * 2 < 0xFFFF0000ll)) {
      used = anip_gemm2_try_splitk(p, (hipStream_t)stream);
      if (used == 0) used = anip_gemm2_try(p, (hipStream_t)stream);
    }
    if (used < 0) return used;
    if (used == 1) {
    } else if (p.conv)
      hipLaunchKernelGGL(gemm_small_kernel<true>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, p);
    else
      hipLaunchKernelGGL(gemm_small_kernel<false>, grid, dim3(NTHREADS), 0, (hipStream_t)stream, p);
  }
  ANIP_LAUNCH_CHECK("anip_gemm");
  return 0;
}
