// HBM-bound normalisation kernels for gfx950: per-frame GroupNorm (+SiLU, + fused channel concat),
// LayerNorm (+ fused temporal positional encoding), fp32 row softmax.
// Channels-last fp16 activations, fp32 statistics, 16-B vector loads/stores, deterministic
// two-level reductions (no atomics).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------
// GroupNorm pass 1: partial (sum, sumsq) per (image n, pixel chunk, group)
// grid (nchunks, N); ws[((n * nchunks + chunk) * G + g) * 2 + {0,1}]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gn_stats_kernel(const f16* __restrict__ x1, int C1, const f16* __restrict__ x2,
                                                     int C2, int64_t HW, int G, int64_t ppc, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [rows_par][C][2]
  const int C = C1 + C2;
  const int CV = C >> 3;
  const int tid = threadIdx.x;
  const int n = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
  const int64_t p0 = (int64_t)chunk * ppc;
  const int64_t p1 = min(HW, p0 + ppc);
  const int rows_par = CV <= NT ? NT / CV : 1;

  for (int cv0 = 0; cv0 < CV; cv0 += NT) {  // executes once unless C > 2048
    int cv, r;
    bool active;
    if (CV <= NT) {
      cv = tid % CV;
      r = tid / CV;
      active = r < rows_par;
    } else {
      cv = cv0 + tid;
      r = 0;
      active = cv < CV;
    }
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (active) {
      const int c = cv << 3;
      const bool second = c >= C1;
      const f16* base = second ? x2 : x1;
      const int Cs = second ? C2 : C1;
      const int cc = second ? c - C1 : c;
      // four independent 16-B loads in flight per thread (one load per trip left ~38 KB per CU in flight: 3.7 TB/s)
      const f16* src = base + (int64_t)n * HW * Cs + cc;
      int64_t p = p0 + r;
      for (; p + 3 * (int64_t)rows_par < p1; p += 4 * (int64_t)rows_par) {
        U4H8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u].u = *(const u32x4*)(src + (p + u * (int64_t)rows_par) * Cs);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[u].e[e];
            s[e] += f;
            q[e] += f * f;
          }
      }
      for (; p < p1; p += rows_par) {
        U4H8 v;
        v.u = *(const u32x4*)(src + p * Cs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v.e[e];
          s[e] += f;
          q[e] += f * f;
        }
      }
      float* dst = sm + ((int64_t)r * C + c) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dst[2 * e] = s[e];
        dst[2 * e + 1] = q[e];
      }
    }
  }
  __syncthreads();
  if (tid < G) {
    const int cpg = C / G;
    float S = 0.f, Q = 0.f;
    for (int r = 0; r < rows_par; ++r)
      for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
        S += sm[((int64_t)r * C + c) * 2];
        Q += sm[((int64_t)r * C + c) * 2 + 1];
      }
    float* o = ws + (((int64_t)n * nchunks + chunk) * G + tid) * 2;
    o[0] = S;
    o[1] = Q;
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm pass 2: finalize stats, y = silu?((x - mean) * rstd * gamma + beta)
// grid (apply_chunks, N)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gn_apply_kernel(const f16* __restrict__ x1, int C1, const f16* __restrict__ x2,
                                                     int C2, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, f16* __restrict__ y, int64_t HW,
                                                     int G, float eps, int silu, const float* __restrict__ ws,
                                                     int nchunks, int64_t ppc_apply, int fps) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // scale[C], shift[C], then mean[G], rstd[G]
  const int C = C1 + C2;
  const int CV = C >> 3;
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  float* scale = sm;
  float* shift = sm + C;
  float* gmean = sm + 2 * C;
  float* grstd = gmean + G;
  const int cpg = C / G;
  {
    // fps = frames per statistic: 1 = per-frame GroupNorm (InflatedGroupNorm); f = plain nn.GroupNorm on the 5-D tensor
    // (src/models/resnet.py:161-164 with use_inflated_groupnorm=False, configs/inference/inference_v1.yaml): the
    // statistics run over (C/G, f, H, W) of a sample — the per-(frame, chunk) partial sums of its f frames are added here.
    // NT / G threads share a group's partials (fixed strided order, then a fixed-order sum of the NT / G parts: still
    // deterministic) — one or two frames come with up to 256 chunks, and 32 threads adding them one after the other cost
    // every apply block of the ReferenceNet / VAE-encoder launches 60 us (round 4)
    float* red = grstd + G;                     // [parts][G][2]
    const int parts = NT / G;
    const int part = tid / G, g = tid - part * G;
    const int n_first = (n / fps) * fps;
    const int total = fps * nchunks;
    if (part < parts) {
      float S = 0.f, Q = 0.f;
      for (int idx = part; idx < total; idx += parts) {
        const float* o = ws + (((int64_t)n_first * nchunks + idx) * G + g) * 2;   // (frame, chunk) pairs are consecutive
        S += o[0];
        Q += o[1];
      }
      red[(part * G + g) * 2] = S;
      red[(part * G + g) * 2 + 1] = Q;
    }
    __syncthreads();
    if (tid < G) {
      float S = 0.f, Q = 0.f;
      for (int pt = 0; pt < parts; ++pt) {
        S += red[(pt * G + tid) * 2];
        Q += red[(pt * G + tid) * 2 + 1];
      }
      const float cnt = (float)((double)HW * cpg * fps);
      const float mean = S / cnt;
      const float var = fmaxf(Q / cnt - mean * mean, 0.f);
      gmean[tid] = mean;
      grstd[tid] = rsqrtf(var + eps);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const int g = c / cpg;
    const float sc = grstd[g] * gamma[c];
    scale[c] = sc;
    shift[c] = beta[c] - gmean[g] * sc;
  }
  __syncthreads();
  const int64_t p0 = (int64_t)blockIdx.x * ppc_apply;
  const int64_t p1 = min(HW, p0 + ppc_apply);
  // thread = (pixel row r of rows_par, channel vector cv): its 8 scale / shift values stay in registers and it walks the
  // pixels r, r + rows_par, ... with four independent 16-B loads in flight (round 2 walked (pixel, vector) with stride
  // NT: one dependent load per trip and four LDS reads per vector — 3.4 TB/s)
  const int rows_par = CV <= NT ? NT / CV : 1;
  for (int cv0 = 0; cv0 < CV; cv0 += NT) {  // executes once unless C > 2048
    int cv, r;
    bool active;
    if (CV <= NT) {
      cv = tid % CV;
      r = tid / CV;
      active = r < rows_par;
    } else {
      cv = cv0 + tid;
      r = 0;
      active = cv < CV;
    }
    if (!active) continue;
    const int c = cv << 3;
    const bool second = c >= C1;
    const f16* src = second ? x2 + (int64_t)n * HW * C2 + (c - C1) : x1 + (int64_t)n * HW * C1 + c;
    const int Cs = second ? C2 : C1;
    f16* dst = y + (int64_t)n * HW * C + c;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = scale[c + e];
      sh[e] = shift[c + e];
    }
    auto norm8 = [&](const U4H8& v, f16* out) {
      U4H8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v.e[e] * sc[e] + sh[e];
        if (silu) f = silu_f(f);
        o.e[e] = (f16)f;
      }
      *(u32x4*)out = o.u;
    };
    int64_t p = p0 + r;
    for (; p + 3 * (int64_t)rows_par < p1; p += 4 * (int64_t)rows_par) {
      U4H8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u].u = *(const u32x4*)(src + (p + u * (int64_t)rows_par) * Cs);
#pragma unroll
      for (int u = 0; u < 4; ++u) norm8(v[u], dst + (p + u * (int64_t)rows_par) * C);
    }
    for (; p < p1; p += rows_par) {
      U4H8 v;
      v.u = *(const u32x4*)(src + p * Cs);
      norm8(v, dst + p * C);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm in ONE launch: one block per (image n, group g) walks the group's slab — HW rows of cpg = C/G channels —
// twice: sum / sum of squares, then normalise (+SiLU) and store.  The second walk hits L2 (a slab is 5 .. 160 KB), so
// HBM sees the tensor once in and once out, and the separate statistics pass with its (image, chunk, group) workspace
// and second launch disappears.  V = halfs per vector (2, 4 or 8: what cpg and the group's byte offset allow).
// Block -> (n, g): the four groups 4x .. 4x+3 of an image run on XCD x (block id % 8), so the 64 / 128-B lines that
// adjacent groups share are fetched into one L2, not eight.
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(NT) void gn_slab_kernel(const f16* __restrict__ x1, int C1, const f16* __restrict__ x2, int C2,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    f16* __restrict__ y, int64_t HW, int G, float eps, int silu) {
  typedef unsigned int uvec __attribute__((ext_vector_type(V / 2)));
  union VH { uvec u; f16 e[V]; };
  __shared__ float red[2][NT / 64];
  __shared__ float sc[256], sh[256];     // cpg <= 256
  const int tid = threadIdx.x;
  const int C = C1 + C2, cpg = C / G, vpr = cpg / V;
  int g, n;
  if ((G & 31) == 0) {                   // XCD-aware: block b -> xcd = b % 8 owns groups (G/8) * xcd .. of every image
    const int b = blockIdx.x, xcd = b & 7, k = b >> 3, gpx = G >> 3;
    g = xcd * gpx + (k % gpx);
    n = k / gpx;
  } else {
    g = blockIdx.x % G;
    n = blockIdx.x / G;
  }
  const int c0 = g * cpg;
  const int64_t nvec = HW * vpr;
  const f16* b1 = x1 + (int64_t)n * HW * C1;
  const f16* b2 = x2 ? x2 + (int64_t)n * HW * C2 : nullptr;
  float s = 0.f, q = 0.f;
  for (int64_t v = tid; v < nvec; v += NT) {
    const int64_t row = v / vpr;
    const int c = c0 + (int)(v - row * vpr) * V;
    const f16* src = c < C1 ? b1 + row * C1 + c : b2 + row * C2 + (c - C1);
    VH t;
    t.u = *(const uvec*)src;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      const float f = (float)t.e[e];
      s += f;
      q += f * f;
    }
  }
  s = wave_sum(s);
  q = wave_sum(q);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s;
    red[1][tid >> 6] = q;
  }
  __syncthreads();
  float S = 0.f, Q = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    S += red[0][w];
    Q += red[1][w];
  }
  const float cnt = (float)((double)HW * cpg);
  const float mean = S / cnt;
  const float rstd = rsqrtf(fmaxf(Q / cnt - mean * mean, 0.f) + eps);
  for (int c = tid; c < cpg; c += NT) {
    const float a = rstd * gamma[c0 + c];
    sc[c] = a;
    sh[c] = beta[c0 + c] - mean * a;
  }
  __syncthreads();
  f16* yb = y + (int64_t)n * HW * C;
  for (int64_t v = tid; v < nvec; v += NT) {
    const int64_t row = v / vpr;
    const int cl = (int)(v - row * vpr) * V;
    const int c = c0 + cl;
    const f16* src = c < C1 ? b1 + row * C1 + c : b2 + row * C2 + (c - C1);
    VH t, o;
    t.u = *(const uvec*)src;
#pragma unroll
    for (int e = 0; e < V; ++e) {
      float f = (float)t.e[e] * sc[cl + e] + sh[cl + e];
      if (silu) f = silu_f(f);
      o.e[e] = (f16)f;
    }
    *(uvec*)(yb + row * C + c) = o.u;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: G lanes (power of two) cooperate on one row, 64/G rows per wave, so that every lane is
// busy for any channel count: lane l of a group owns the 16-B chunks l, l+G, l+2G, ... (NCH of them,
// kept in registers between the mean pass and the variance pass).  C % 8 == 0, C <= 2560.
// G is chosen on the host as the smallest power of two with ceil(C/8 / G) <= 5.
// ---------------------------------------------------------------------------------------------
constexpr int LN_MAXCH = 5;  // chunks of 8 channels per lane

template <int G, int NCH>
__global__ __launch_bounds__(NT) void layernorm_kernel(const f16* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, f16* __restrict__ y, int64_t M,
                                                      int C, float eps, const float* __restrict__ pe,
                                                      int64_t rows_per_frame, int F) {
  constexpr int RPB = NT / G;  // rows per block
  const int gl = threadIdx.x % G;
  const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / G;
  const bool rvalid = row < M;
  const int CV = C >> 3;
  const f16* xr = x + (rvalid ? row : 0) * C;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int cv = gl + G * k;
    U4H8 t;
    t.u = u32x4{0u, 0u, 0u, 0u};
    if (rvalid && cv < CV) t.u = *(const u32x4*)(xr + cv * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[k][e] = (float)t.e[e];
      s += v[k][e];
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int cv = gl + G * k;
    if (cv < CV) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[k][e] - mean;
        q += d * d;
      }
    }
  }
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)C + eps);
  if (!rvalid) return;
  const float* per = nullptr;
  if (pe != nullptr) per = pe + ((row / rows_per_frame) % F) * C;
  f16* yr = y + row * C;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int cv = gl + G * k;
    if (cv < CV) {
      const int c = cv * 8;
      const float4 g0 = *(const float4*)(gamma + c), g1 = *(const float4*)(gamma + c + 4);
      const float4 b0 = *(const float4*)(beta + c), b1 = *(const float4*)(beta + c + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      U4H8 o;
      if (per != nullptr) {
        const float4 p0 = *(const float4*)(per + c), p1 = *(const float4*)(per + c + 4);
        const float pp[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (f16)((v[k][e] - mean) * rstd * gg[e] + bb[e] + pp[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o.e[e] = (f16)((v[k][e] - mean) * rstd * gg[e] + bb[e]);
      }
      *(u32x4*)(yr + c) = o.u;
    }
  }
}

template <int G>
void launch_layernorm(int nch, dim3 grid, hipStream_t st, const f16* x, const float* gamma, const float* beta, f16* y,
                      int64_t M, int C, float eps, const float* pe, int64_t rpf, int F) {
  switch (nch) {
    case 1: hipLaunchKernelGGL((layernorm_kernel<G, 1>), grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps, pe, rpf, F); break;
    case 2: hipLaunchKernelGGL((layernorm_kernel<G, 2>), grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps, pe, rpf, F); break;
    case 3: hipLaunchKernelGGL((layernorm_kernel<G, 3>), grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps, pe, rpf, F); break;
    case 4: hipLaunchKernelGGL((layernorm_kernel<G, 4>), grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps, pe, rpf, F); break;
    default: hipLaunchKernelGGL((layernorm_kernel<G, 5>), grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps, pe, rpf, F); break;
  }
}

// ---------------------------------------------------------------------------------------------
// row softmax fp32 -> fp16, one block per row
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void softmax_rows_kernel(const float* __restrict__ s, f16* __restrict__ p, int cols) {
  __shared__ float red[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sr = s + (int64_t)blockIdx.x * cols;
  f16* pr = p + (int64_t)blockIdx.x * cols;
  float mx = -INFINITY;
  for (int c = tid; c < cols; c += NT) mx = fmaxf(mx, sr[c]);
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int c = tid; c < cols; c += NT) sum += __expf(sr[c] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = red[0] + red[1] + red[2] + red[3];
  const float inv = 1.0f / sum;
  for (int c = tid; c < cols; c += NT) pr[c] = (f16)(__expf(sr[c] - mx) * inv);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm pass 2 for a consumer that applies the affine form itself (anip_affine_linear320): finalise the statistics of frame
// n exactly as gn_apply_kernel does and write (scale, shift)[n][c] = (rstd gamma, beta - mean rstd gamma).  grid (N)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gn_scale_shift_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ out, int C, int64_t HW, int G, float eps,
                                                           const float* __restrict__ ws, int nchunks) {
  __shared__ float gmean[NT], grstd[NT], red[2 * NT];
  const int tid = threadIdx.x, n = blockIdx.x;
  const int cpg = C / G;
  const int parts = NT / G;
  const int part = tid / G, g = tid - part * G;
  if (part < parts) {
    float S = 0.f, Q = 0.f;
    for (int idx = part; idx < nchunks; idx += parts) {
      const float* o = ws + (((int64_t)n * nchunks + idx) * G + g) * 2;
      S += o[0];
      Q += o[1];
    }
    red[(part * G + g) * 2] = S;
    red[(part * G + g) * 2 + 1] = Q;
  }
  __syncthreads();
  if (tid < G) {
    float S = 0.f, Q = 0.f;
    for (int pt = 0; pt < parts; ++pt) {
      S += red[(pt * G + tid) * 2];
      Q += red[(pt * G + tid) * 2 + 1];
    }
    const float cnt = (float)((double)HW * cpg);
    const float mean = S / cnt;
    const float var = fmaxf(Q / cnt - mean * mean, 0.f);
    gmean[tid] = mean;
    grstd[tid] = rsqrtf(var + eps);
  }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const int gg = c / cpg;
    const float sc = grstd[gg] * gamma[c];
    out[((int64_t)n * C + c) * 2] = sc;
    out[((int64_t)n * C + c) * 2 + 1] = beta[c] - gmean[gg] * sc;
  }
}

static void gn_chunks(int N, int64_t HW, int* nchunks, int64_t* ppc) {
  int64_t want = (1024 + N - 1) / N;          // aim at >= ~1024 workgroups
  int64_t maxc = (HW + 15) / 16;              // at least 16 pixels per chunk (small maps still get >= 128 workgroups)
  int64_t nc = want < maxc ? want : maxc;
  if (nc < 1) nc = 1;
  if (nc > 256) nc = 256;
  *ppc = (HW + nc - 1) / nc;
  *nchunks = (int)((HW + *ppc - 1) / *ppc);
}

}  // namespace

extern "C" int64_t anip_groupnorm_ws_floats(int N, int64_t HW, int C, int G) {
  int nchunks;
  int64_t ppc;
  gn_chunks(N, HW, &nchunks, &ppc);
  (void)C;
  return (int64_t)N * nchunks * G * 2;
}

// 1 if anip_groupnorm runs this problem as ONE kernel (gn_slab_kernel), 0 if as statistics + apply.
// Single-launch form for the 8x8 / 16x16 levels (measured on MI355X, 32 frames, stats + apply vs one slab kernel:
// 16x16 C1280 27.8 -> 18.7 us, C2560 46.7 -> 31.5, 8x8 C1280 19.3 -> 14.6, C2560 29.9 -> 15.6; from 32x32 up the slab's
// narrow row segments (cpg * 2 = 20 .. 120 B) lose to the two coalesced passes: 64x64 C320 57 -> 147 us).
extern "C" int anip_groupnorm_single_launch(int N, int64_t HW, int C, int G) {
  constexpr int slab_hw = 256;
  if (G <= 0 || C % G != 0) return 0;
  const int cpg = C / G;
  return (slab_hw > 0 && HW <= slab_hw && cpg <= 256 && (cpg & 1) == 0 && (int64_t)N * G >= 256) ? 1 : 0;
}

extern "C" int anip_groupnorm(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                              void* y, int N, int64_t HW, int G, float eps, int silu, float* ws, void* stream) {
  return anip_groupnorm_frames(x1, C1, x2, C2, gamma, beta, y, N, HW, G, eps, silu, 1, ws, stream);
}

extern "C" int anip_groupnorm_frames(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                                     void* y, int N, int64_t HW, int G, float eps, int silu, int frames_per_stat, float* ws,
                                     void* stream) {
  const int C = C1 + C2;
  ANIP_REQUIRE(frames_per_stat >= 1 && N % frames_per_stat == 0,
               "anip_groupnorm_frames: N=%d is not a multiple of frames_per_stat=%d", N, frames_per_stat);
  ANIP_REQUIRE(x1 && y && gamma && beta && ws, "anip_groupnorm: null pointer");
  ANIP_REQUIRE(N > 0 && HW > 0 && G > 0 && G <= NT, "anip_groupnorm: bad sizes N=%d HW=%lld G=%d", N, (long long)HW, G);
  ANIP_REQUIRE((C1 & 7) == 0 && (C2 & 7) == 0 && C % G == 0, "anip_groupnorm: C1=%d C2=%d must be multiples of 8, C %% G == 0", C1, C2);
  ANIP_REQUIRE((C2 == 0) == (x2 == nullptr), "anip_groupnorm: x2/C2 mismatch");
  ANIP_REQUIRE(C <= 8192, "anip_groupnorm: C=%d too large", C);
  {
    if (frames_per_stat == 1 && anip_groupnorm_single_launch(N, HW, C, G)) {
      const int cpg = C / G;
      const int V = ((cpg & 7) == 0) ? 8 : ((cpg & 3) == 0) ? 4 : 2;
      AnipProfScope prof_(ANIP_K_GN_APPLY, (void*)stream);
      const dim3 grid((unsigned)(N * G));
      hipStream_t st = (hipStream_t)stream;
      if (V == 8)
        hipLaunchKernelGGL(gn_slab_kernel<8>, grid, dim3(NT), 0, st, (const f16*)x1, C1, (const f16*)x2, C2, gamma, beta,
                           (f16*)y, HW, G, eps, silu);
      else if (V == 4)
        hipLaunchKernelGGL(gn_slab_kernel<4>, grid, dim3(NT), 0, st, (const f16*)x1, C1, (const f16*)x2, C2, gamma, beta,
                           (f16*)y, HW, G, eps, silu);
      else
        hipLaunchKernelGGL(gn_slab_kernel<2>, grid, dim3(NT), 0, st, (const f16*)x1, C1, (const f16*)x2, C2, gamma, beta,
                           (f16*)y, HW, G, eps, silu);
      ANIP_LAUNCH_CHECK("anip_groupnorm(slab)");
      return 0;
    }
  }
  int nchunks;
  int64_t ppc;
  gn_chunks(N, HW, &nchunks, &ppc);
  const int CV = C / 8;
  const int rows_par = CV <= NT ? NT / CV : 1;
  const size_t sm1 = (size_t)rows_par * C * 2 * sizeof(float);
  ANIP_REQUIRE(sm1 <= 65536, "anip_groupnorm: stats LDS %zu too large", sm1);
  {
    AnipProfScope prof_(ANIP_K_GN_STATS, (void*)stream);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, N), dim3(NT), sm1, (hipStream_t)stream, (const f16*)x1, C1,
                       (const f16*)x2, C2, HW, G, ppc, ws);
  }
  ANIP_LAUNCH_CHECK("anip_groupnorm(stats)");
  // apply: ~2048 pixels*C/8 vectors per block at least, >= 1024 blocks when possible
  constexpr int apply_blocks = 1024;
  int64_t want = (apply_blocks + N - 1) / N;
  int64_t maxc = (HW + 15) / 16;
  int64_t ac = want < maxc ? want : maxc;
  if (ac < 1) ac = 1;
  const int64_t ppa = (HW + ac - 1) / ac;
  const int achunks = (int)((HW + ppa - 1) / ppa);
  const size_t sm2 = (size_t)(2 * C + 2 * G + 2 * NT) * sizeof(float);   // scale, shift, mean, rstd, finalize partials
  {
    AnipProfScope prof_(ANIP_K_GN_APPLY, (void*)stream);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(achunks, N), dim3(NT), sm2, (hipStream_t)stream, (const f16*)x1, C1,
                       (const f16*)x2, C2, gamma, beta, (f16*)y, HW, G, eps, silu, (const float*)ws, nchunks, ppa,
                       frames_per_stat);
  }
  ANIP_LAUNCH_CHECK("anip_groupnorm(apply)");
  return 0;
}

// GroupNorm statistics of x [N][HW][C] -> scale_shift [N][C][2] fp32 = (rstd gamma, beta - mean rstd gamma): the per-frame affine
// form of InflatedGroupNorm for a consumer that applies it on the fly (anip_affine_linear320).  ws as for anip_groupnorm.
extern "C" int anip_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float* scale_shift, int N,
                                          int64_t HW, int C, int G, float eps, float* ws, void* stream) {
  ANIP_REQUIRE(x && gamma && beta && scale_shift && ws, "anip_groupnorm_scale_shift: null pointer");
  ANIP_REQUIRE(N > 0 && HW > 0 && G > 0 && G <= NT && (C & 7) == 0 && C % G == 0 && C <= 8192,
               "anip_groupnorm_scale_shift: bad sizes N=%d HW=%lld C=%d G=%d", N, (long long)HW, C, G);
  int nchunks;
  int64_t ppc;
  gn_chunks(N, HW, &nchunks, &ppc);
  const int CV = C / 8;
  const int rows_par = CV <= NT ? NT / CV : 1;
  const size_t sm1 = (size_t)rows_par * C * 2 * sizeof(float);
  ANIP_REQUIRE(sm1 <= 65536, "anip_groupnorm_scale_shift: stats LDS %zu too large", sm1);
  {
    AnipProfScope prof_(ANIP_K_GN_STATS, (void*)stream);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, N), dim3(NT), sm1, (hipStream_t)stream, (const f16*)x, C, (const f16*)nullptr,
                       0, HW, G, ppc, ws);
  }
  ANIP_LAUNCH_CHECK("anip_groupnorm_scale_shift(stats)");
  {
    AnipProfScope prof_(ANIP_K_GN_APPLY, (void*)stream);
    hipLaunchKernelGGL(gn_scale_shift_kernel, dim3(N), dim3(NT), 0, (hipStream_t)stream, gamma, beta, scale_shift, C, HW, G, eps,
                       (const float*)ws, nchunks);
  }
  ANIP_LAUNCH_CHECK("anip_groupnorm_scale_shift(finalise)");
  return 0;
}

extern "C" int anip_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t M, int C,
                              float eps, const float* pe, int64_t rows_per_frame, int F, void* stream) {
  ANIP_REQUIRE(x && y && gamma && beta, "anip_layernorm: null pointer");
  ANIP_REQUIRE(M > 0 && C > 0 && (C & 7) == 0 && C <= LN_MAXCH * 512, "anip_layernorm: bad C=%d (multiple of 8, <= %d)", C, LN_MAXCH * 512);
  if (pe != nullptr) ANIP_REQUIRE(rows_per_frame > 0 && F > 0, "anip_layernorm: pe needs rows_per_frame, F");
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)pe) & 15) == 0,
               "anip_layernorm: pointers must be 16-B aligned");
  const int CV = C / 8;
  int G = 8;
  while (G < 64 && (CV + G - 1) / G > LN_MAXCH) G <<= 1;
  const int nch = (CV + G - 1) / G;
  const dim3 grid((unsigned)cdiv64(M, NT / G));
  {
    AnipProfScope prof_(ANIP_K_LAYERNORM, (void*)stream);
    hipStream_t st = (hipStream_t)stream;
    const f16* xx = (const f16*)x;
    f16* yy = (f16*)y;
    switch (G) {
      case 8: launch_layernorm<8>(nch, grid, st, xx, gamma, beta, yy, M, C, eps, pe, rows_per_frame, F); break;
      case 16: launch_layernorm<16>(nch, grid, st, xx, gamma, beta, yy, M, C, eps, pe, rows_per_frame, F); break;
      case 32: launch_layernorm<32>(nch, grid, st, xx, gamma, beta, yy, M, C, eps, pe, rows_per_frame, F); break;
      default: launch_layernorm<64>(nch, grid, st, xx, gamma, beta, yy, M, C, eps, pe, rows_per_frame, F); break;
    }
  }
  ANIP_LAUNCH_CHECK("anip_layernorm");
  return 0;
}

extern "C" int anip_softmax_rows(const float* s, void* p, int64_t rows, int cols, void* stream) {
  ANIP_REQUIRE(s && p && rows > 0 && cols > 0, "anip_softmax_rows: bad arguments");
  {
    AnipProfScope prof_(ANIP_K_SOFTMAX, (void*)stream);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(NT), 0, (hipStream_t)stream, s, (f16*)p, cols);
  }
  ANIP_LAUNCH_CHECK("anip_softmax_rows");
  return 0;
}
