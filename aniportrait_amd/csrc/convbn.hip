// PoseGuider front-end kernels for gfx950 (src/models/pose_guider.py:19-85): direct convolution for the
// small-channel stem (3/16/32-channel 3x3 and 4x4-stride-2 convs, where an implicit GEMM would waste the
// MFMA tile) and train-/eval-mode BatchNorm2d + ReLU over channels-last activations.  All of it is
// HBM-bound: 16-B vector accesses where the channel count allows, fp32 statistics, deterministic two-level
// reductions (per-block partials, then one fp64 finalize per channel; no atomics).
#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------------
// direct convolution: x [N][H][W][Cin] fp16, wp [ks*ks*Cin][Cout8] fp16 (Cout padded to a multiple of 8,
// output channel fastest), y [N][Ho][Wo][Cout].  thread -> (output pixel, group of 8 output channels);
// the weights sit in LDS and the 8 weights of one (tap, ci) are ONE 16-B broadcast read.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void conv_direct_kernel(const f16* __restrict__ x, const f16* __restrict__ wp,
                                                        const float* __restrict__ bias, const f16* __restrict__ res,
                                                        f16* __restrict__ y, int N, int H, int W, int Cin, int Cout,
                                                        int Cout8, int ks, int stride, int pad, int Ho, int Wo,
                                                        int pix_per_block, int relu) {
  extern __shared__ __attribute__((aligned(16))) f16 sw[];
  const int tid = threadIdx.x;
  const int nw8 = ks * ks * Cin * (Cout8 >> 3);
  for (int i = tid; i < nw8; i += NT) ((u32x4*)sw)[i] = ((const u32x4*)wp)[i];
  __syncthreads();
  const int cg_n = Cout8 >> 3;
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int64_t pix0 = (int64_t)blockIdx.x * pix_per_block;
  const bool vec_in = (Cin & 7) == 0;
  const bool vec4_in = Cin == 4;
  for (int o = tid; o < pix_per_block * cg_n; o += NT) {
    const int pl = o / cg_n, cg = o - pl * cg_n;
    const int64_t pix = pix0 + pl;
    if (pix >= npix) break;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int64_t img = pix / ((int64_t)Wo * Ho);
    const int co0 = cg * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = (bias != nullptr && co0 + e < Cout) ? bias[co0 + e] : 0.f;
    for (int ky = 0; ky < ks; ++ky) {
      const int iy = oy * stride + ky - pad;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < ks; ++kx) {
        const int ix = ox * stride + kx - pad;
        if (ix < 0 || ix >= W) continue;
        const f16* xp = x + ((img * H + iy) * (int64_t)W + ix) * Cin;
        const f16* wt = sw + (int64_t)((ky * ks + kx) * Cin) * Cout8 + co0;
        if (vec_in) {
          for (int c8 = 0; c8 < Cin; c8 += 8) {
            U4H8 xv;
            xv.u = *(const u32x4*)(xp + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              U4H8 wv;
              wv.u = *(const u32x4*)(wt + (c8 + j) * Cout8);
              const float xf = (float)xv.e[j];
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[e] += xf * (float)wv.e[e];
            }
          }
        } else if (vec4_in) {
          union { u32x2 u; f16 e[4]; } xv;
          xv.u = *(const u32x2*)xp;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            U4H8 wv;
            wv.u = *(const u32x4*)(wt + j * Cout8);
            const float xf = (float)xv.e[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += xf * (float)wv.e[e];
          }
        } else {
          for (int ci = 0; ci < Cin; ++ci) {
            U4H8 wv;
            wv.u = *(const u32x4*)(wt + ci * Cout8);
            const float xf = (float)xp[ci];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += xf * (float)wv.e[e];
          }
        }
      }
    }
    f16* yp = y + pix * Cout + co0;
    if (res != nullptr) {
      const f16* rp = res + pix * Cout + co0;
      if ((Cout & 7) == 0) {
        U4H8 rv;
        rv.u = *(const u32x4*)rp;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)rv.e[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (co0 + e < Cout) acc[e] += (float)rp[e];
      }
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    if ((Cout & 7) == 0) {
      U4H8 ov;
#pragma unroll
      for (int e = 0; e < 8; ++e) ov.e[e] = (f16)acc[e];
      *(u32x4*)yp = ov.u;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (co0 + e < Cout) yp[e] = (f16)acc[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// conv_in of the UNets (4 latent channels -> 320, 3x3, stride 1, pad 1, + fused pose-feature add): the generic kernel above
// walks the taps with a data-dependent `continue` per tap, so its nine 8-B input loads are issued and waited for one
// after the other, and decodes the pixel with 64-bit divisions — 210 us for 84 MB of output + 84 MB of pose features
// (0.8 TB/s); with the loads batched and 32-bit indices it was still VALU-bound at 137 us: 288 f16 -> f32 conversions + 288
// FMAs per item.  Here the nine loads of an item are in flight together (out-of-image taps read as zero), the index
// arithmetic is 32-bit, a block keeps its 23 KB of weights for 16 items per thread, and the 36-deep contraction runs as 18
// v_dot2_f32_f16 per output channel (exact fp16 products, fp32 accumulation; rounding differs from the generic kernel's
// sequential fp32 FMAs in the last bit of the fp32 sum).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void conv3x3_c4_kernel(const f16* __restrict__ x, const f16* __restrict__ wp,
                                                       const float* __restrict__ bias, const f16* __restrict__ res,
                                                       f16* __restrict__ y, int N, int H, int W, int Cout, int pix_per_block,
                                                       int relu) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  // LDS image: [9 taps][2 channel pairs][Cout] half2 = {w[tap][2 pr][co], w[tap][2 pr + 1][co]}: one v_dot2_f32_f16 per
  // (tap, pair, output channel) — fp16 products, fp32 accumulate, no conversion instructions
  extern __shared__ __attribute__((aligned(16))) f16 sw[];
  const int tid = threadIdx.x;
  const int cg_n = Cout >> 3;
  for (int i = tid; i < 36 * cg_n; i += NT) {
    const int k = i / cg_n, cgi = i - k * cg_n;      // k = tap * 4 + ci
    U4H8 v;
    v.u = ((const u32x4*)wp)[i];
    f16* dst = sw + ((((k >> 2) * 2 + ((k >> 1) & 1)) * Cout + cgi * 8) * 2 + (k & 1));
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[e * 2] = v.e[e];
  }
  __syncthreads();
  const int npix = N * H * W;
  const int pix0 = blockIdx.x * pix_per_block;
  const int nitems = min(pix_per_block, npix - pix0) * cg_n;
  for (int o = tid; o < nitems; o += NT) {
    const int pl = o / cg_n, cg = o - pl * cg_n;
    const int pix = pix0 + pl;
    const int ox = pix % W;
    const int t = pix / W;
    const int oy = t % H;
    const int img = t / H;
    const int co0 = cg * 8;
    union X4 { u32x2 u; h2 p[2]; };
    X4 xv[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
      const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
      xv[tap].u = u32x2{0u, 0u};                      // out-of-image taps contribute exact zeros
      if (in) xv[tap].u = *(const u32x2*)(x + ((int64_t)(img * H + iy) * W + ix) * 4);
    }
    float acc[8];
    if (bias != nullptr) {
      const float4 b0 = *(const float4*)(bias + co0), b1 = *(const float4*)(bias + co0 + 4);
      acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w;
      acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    }
    U4H8 rv;
    rv.u = u32x4{0u, 0u, 0u, 0u};
    if (res != nullptr) rv.u = *(const u32x4*)(res + (int64_t)pix * Cout + co0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        union { u32x4 u[2]; h2 p[8]; } wv;
        const u32x4* wsrc = (const u32x4*)(sw + ((tap * 2 + pr) * Cout + co0) * 2);
        wv.u[0] = wsrc[0];
        wv.u[1] = wsrc[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = __builtin_amdgcn_fdot2(xv[tap].p[pr], wv.p[e], acc[e], false);
      }
    }
    if (res != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)rv.e[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
    }
    U4H8 ov;
#pragma unroll
    for (int e = 0; e < 8; ++e) ov.e[e] = (f16)acc[e];
    *(u32x4*)(y + (int64_t)pix * Cout + co0) = ov.u;
  }
}

// ---------------------------------------------------------------------------------------------------
// BatchNorm pass 1: per-block partial (sum, sumsq) per channel over a chunk of rows of x [M][C].
// part[(blk * C + c) * 2 + {0,1}]
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void bn_stats_kernel(const f16* __restrict__ x, int64_t M, int C, int64_t rows_per_block,
                                                     float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(M, r0 + rows_per_block);
  if ((C & 7) == 0 && (C >> 3) <= NT) {
    // vector path: thread -> (channel vector cv, row lane r); sm[r][C][2]
    const int CV = C >> 3;
    const int rows_par = NT / CV;
    const int cv = tid % CV, r = tid / CV;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (r < rows_par) {
      for (int64_t m = r0 + r; m < r1; m += rows_par) {
        U4H8 v;
        v.u = *(const u32x4*)(x + m * C + cv * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)v.e[e];
          s[e] += f;
          q[e] += f * f;
        }
      }
      float* dst = sm + ((int64_t)r * C + cv * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        dst[2 * e] = s[e];
        dst[2 * e + 1] = q[e];
      }
    }
    __syncthreads();
    for (int c = tid; c < C; c += NT) {
      float S = 0.f, Q = 0.f;
      for (int rr = 0; rr < rows_par; ++rr) {
        S += sm[((int64_t)rr * C + c) * 2];
        Q += sm[((int64_t)rr * C + c) * 2 + 1];
      }
      part[((int64_t)blockIdx.x * C + c) * 2] = S;
      part[((int64_t)blockIdx.x * C + c) * 2 + 1] = Q;
    }
  } else {
    // scalar path (C < NT, e.g. 3): the first NTa = (NT / C) * C threads walk the chunk's elements with
    // stride NTa, so a thread's channel (tid % C) is fixed; sm[tid][2]
    const int NTa = (NT / C) * C;
    float s = 0.f, q = 0.f;
    if (tid < NTa) {
      const int64_t e0 = r0 * C, e1 = r1 * C;
      for (int64_t i = e0 + tid; i < e1; i += NTa) {
        const float f = (float)x[i];
        s += f;
        q += f * f;
      }
    }
    sm[2 * tid] = s;
    sm[2 * tid + 1] = q;
    __syncthreads();
    if (tid < C) {
      float S = 0.f, Q = 0.f;
      for (int t = tid; t < NTa; t += C) {
        S += sm[2 * t];
        Q += sm[2 * t + 1];
      }
      part[((int64_t)blockIdx.x * C + tid) * 2] = S;
      part[((int64_t)blockIdx.x * C + tid) * 2 + 1] = Q;
    }
  }
}

// BatchNorm pass 2: one thread per channel; ss[c] = scale, ss[C + c] = shift
//   train: mean / biased variance of the batch (nn.BatchNorm2d in training mode, F.batch_norm(training=True))
//   eval : running statistics
__global__ __launch_bounds__(NT) void bn_finalize_kernel(const float* __restrict__ part, int nblk, int64_t M, int C,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ rmean, const float* __restrict__ rvar,
                                                        float eps, float* __restrict__ ss) {
  const int c = blockIdx.x * NT + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (rmean != nullptr) {
    mean = (double)rmean[c];
    var = (double)rvar[c];
  } else {
    double S = 0.0, Q = 0.0;
    for (int b = 0; b < nblk; ++b) {
      S += (double)part[((int64_t)b * C + c) * 2];
      Q += (double)part[((int64_t)b * C + c) * 2 + 1];
    }
    mean = S / (double)M;
    var = Q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
  }
  const double sc = (double)gamma[c] / sqrt(var + (double)eps);
  ss[c] = (float)sc;
  ss[C + c] = (float)((double)beta[c] - mean * sc);
}

// BatchNorm pass 3: y = relu?(x * scale[c] + shift[c])
__global__ __launch_bounds__(NT) void bn_apply_kernel(const f16* __restrict__ x, f16* __restrict__ y, int64_t M, int C,
                                                     const float* __restrict__ ss, int relu) {
  const int64_t stride = (int64_t)gridDim.x * NT;
  if ((C & 7) == 0) {
    const int CV = C >> 3;
    const int64_t nvec = M * CV;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += stride) {
      const int c = (int)(i % CV) * 8;
      U4H8 v, o;
      v.u = ((const u32x4*)x)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v.e[e] * ss[c + e] + ss[C + c + e];
        if (relu) f = fmaxf(f, 0.f);
        o.e[e] = (f16)f;
      }
      ((u32x4*)y)[i] = o.u;
    }
  } else {
    const int64_t n = M * C;
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
      const int c = (int)(i % C);
      float f = (float)x[i] * ss[c] + ss[C + c];
      if (relu) f = fmaxf(f, 0.f);
      y[i] = (f16)f;
    }
  }
}

inline int bn_blocks(int64_t M, int C) {
  // ~64K elements per block, at most 2048 blocks
  int64_t rows = (65536 + C - 1) / C;
  if (rows < 1) rows = 1;
  int64_t nb = cdiv64(M, rows);
  if (nb > 2048) nb = 2048;
  if (nb < 1) nb = 1;
  return (int)nb;
}

}  // namespace

extern "C" int anip_conv_direct(const void* x, const void* wp, const float* bias, const void* residual, void* y, int N,
                                int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int relu, void* stream) {
  ANIP_REQUIRE(x && wp && y, "anip_conv_direct: null pointer");
  ANIP_REQUIRE(ksize >= 1 && ksize <= 5 && (stride == 1 || stride == 2) && pad >= 0 && pad < ksize,
               "anip_conv_direct: ksize=%d stride=%d pad=%d unsupported", ksize, stride, pad);
  ANIP_REQUIRE(N > 0 && H > 0 && W > 0 && Cin >= 1 && Cout >= 1, "anip_conv_direct: bad sizes");
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)wp | (uintptr_t)y | (uintptr_t)residual) & 15) == 0,
               "anip_conv_direct: pointers must be 16-B aligned");
  const int Cout8 = (Cout + 7) & ~7;
  const size_t lds = (size_t)ksize * ksize * Cin * Cout8 * sizeof(f16);
  ANIP_REQUIRE(lds <= 65536, "anip_conv_direct: weights (%zu B) do not fit in 64 KB of LDS; use anip_gemm (conv)", lds);
  const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
  ANIP_REQUIRE(Ho > 0 && Wo > 0, "anip_conv_direct: empty output");
  const int cg_n = Cout8 >> 3;
  if (ksize == 3 && Cin == 4 && stride == 1 && pad == 1 && (Cout & 7) == 0 && (((uintptr_t)bias) & 15) == 0 &&
      (int64_t)N * H * W * (int64_t)(Cout >> 3) < (1ll << 31)) {
    int ppb4 = (NT * 16) / cg_n;   // ~16 (pixel, channel-group) items per thread
    if (ppb4 < 1) ppb4 = 1;
    const int64_t blocks4 = cdiv64((int64_t)N * H * W, ppb4);
    {
      AnipProfScope prof_(ANIP_K_CONV_SMALL, (void*)stream);
      hipLaunchKernelGGL(conv3x3_c4_kernel, dim3((unsigned)blocks4), dim3(NT), lds, (hipStream_t)stream, (const f16*)x,
                         (const f16*)wp, bias, (const f16*)residual, (f16*)y, N, H, W, Cout, ppb4, relu);
    }
    ANIP_LAUNCH_CHECK("anip_conv_direct(c4)");
    return 0;
  }
  int ppb = (NT * 4) / cg_n;  // ~4 (pixel, channel-group) items per thread
  if (ppb < 1) ppb = 1;
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int64_t blocks = cdiv64(npix, ppb);
  ANIP_REQUIRE(blocks < (1ll << 31), "anip_conv_direct: grid too large");
  {
    AnipProfScope prof_(ANIP_K_CONV_SMALL, (void*)stream);
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)blocks), dim3(NT), lds, (hipStream_t)stream, (const f16*)x,
                       (const f16*)wp, bias, (const f16*)residual, (f16*)y, N, H, W, Cin, Cout, Cout8, ksize, stride, pad, Ho,
                       Wo, ppb, relu);
  }
  ANIP_LAUNCH_CHECK("anip_conv_direct");
  return 0;
}

extern "C" int64_t anip_batchnorm_ws_floats(int64_t M, int C) { return (int64_t)bn_blocks(M, C) * C * 2 + 2 * (int64_t)C; }

extern "C" int anip_batchnorm(const void* x, const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, void* y, int64_t M, int C, float eps, int relu, float* ws,
                              void* stream) {
  ANIP_REQUIRE(x && gamma && beta && y && ws, "anip_batchnorm: null pointer");
  ANIP_REQUIRE(M > 0 && C > 0, "anip_batchnorm: bad sizes");
  ANIP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "anip_batchnorm: running_mean / running_var must come together");
  ANIP_REQUIRE((C & 7) == 0 ? (C >> 3) <= NT : C <= NT, "anip_batchnorm: C=%d unsupported (need C %% 8 == 0 and C <= 2048, or C <= 256)", C);
  ANIP_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "anip_batchnorm: pointers must be 16-B aligned");
  const int nblk = bn_blocks(M, C);
  float* part = ws;
  float* ss = ws + (int64_t)nblk * C * 2;
  hipStream_t s = (hipStream_t)stream;
  AnipProfScope prof_(ANIP_K_BATCHNORM, stream);
  if (running_mean == nullptr) {
    const int64_t rpb = cdiv64(M, nblk);
    size_t lds;
    if ((C & 7) == 0) lds = (size_t)(NT / (C >> 3)) * C * 2 * sizeof(float);
    else lds = (size_t)NT * 2 * sizeof(float);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(NT), lds, s, (const f16*)x, M, C, rpb, part);
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + NT - 1) / NT), dim3(NT), 0, s, part, nblk, M, C, gamma, beta,
                     running_mean, running_var, eps, ss);
  const int64_t work = (C & 7) == 0 ? M * (C >> 3) : M * C;
  int64_t blocks = cdiv64(work, NT * 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(NT), 0, s, (const f16*)x, (f16*)y, M, C, ss, relu);
  ANIP_LAUNCH_CHECK("anip_batchnorm");
  return 0;
}
