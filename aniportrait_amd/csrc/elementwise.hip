// Small / HBM-bound kernels for gfx950: direct convolution for tiny channel counts, M<=16 dense
// layers, residual add, sliding-window accumulation, fused CFG + DDIM(v-prediction) update, and the
// boundary layout conversions between (B,C,F,H,W) and channels-last frames.
#include "common.h"

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------
// y[m][n] = sum_k f(x[m][k]) W[n][k] + b[n], M <= 16 (time-embedding MLP, the stacked time_emb_proj of all resnets,
// collapsed attn2): a weight-streaming GEMV.  f(x) (SiLU or identity) is evaluated ONCE per block into LDS — round 4:
// the first version applied it per output column, 51 M exp + rcp for the stacked projection (N = 20160, K = 1280, M = 2:
// 102 us = 0.5 TB/s of weight traffic) — and a wave walks LS_COLS columns with all their 16-B weight loads of a K-step in
// flight.  One block = 4 waves = 4 * LS_COLS columns.
// ---------------------------------------------------------------------------------------------
constexpr int LS_COLS = 4;
template <int MT>
__global__ __launch_bounds__(NT) void linear_small_kernel(const float* __restrict__ x, const f16* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ y, int M,
                                                         int N, int K, int silu_in) {
  extern __shared__ __attribute__((aligned(16))) float sx[];   // [M][K]
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < M * K; i += NT) {
    float xv = x[i];
    if (silu_in) xv = silu_f(xv);
    sx[i] = xv;
  }
  __syncthreads();
  const int n0 = (blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * LS_COLS;
  if (n0 >= N) return;
  float acc[LS_COLS][MT];
#pragma unroll
  for (int j = 0; j < LS_COLS; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[j][m] = 0.f;
  for (int k = lane * 8; k < K; k += 64 * 8) {
    U4H8 wv[LS_COLS];
#pragma unroll
    for (int j = 0; j < LS_COLS; ++j) {
      const int n = min(n0 + j, N - 1);
      wv[j].u = *(const u32x4*)(W + (int64_t)n * K + k);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {
        const float4 xa = *(const float4*)(sx + m * K + k), xb = *(const float4*)(sx + m * K + k + 4);
        const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
        for (int j = 0; j < LS_COLS; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][m] += xs[e] * (float)wv[j].e[e];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < LS_COLS; ++j) {
    const int n = n0 + j;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M) {
        const float s_ = wave_sum(acc[j][m]);
        if (lane == 0 && n < N) y[(int64_t)m * N + n] = s_ + (bias ? bias[n] : 0.f);
      }
    }
  }
}

__global__ __launch_bounds__(NT) void add_kernel(const f16* __restrict__ a, const f16* __restrict__ b,
                                                f16* __restrict__ o, int64_t nvec, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += stride) {
    U4H8 x, y2, r;
    x.u = ((const u32x4*)a)[i];
    y2.u = ((const u32x4*)b)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) r.e[e] = (f16)((float)x.e[e] + (float)y2.e[e]);
    ((u32x4*)o)[i] = r.u;
  }
  // tail
  if (blockIdx.x == 0) {
    for (int64_t i = nvec * 8 + threadIdx.x; i < n; i += NT) o[i] = (f16)((float)a[i] + (float)b[i]);
  }
}

// acc[s][frames[j]][e] += pred[s][j][e]; counter[frames[j]] += 1 (one thread block column for counters).
// frames[j] < 0: window slot j is skipped (the host masks all but the last occurrence of a frame a dilated window
// visits twice, so no two threads ever update the same element)
__global__ __launch_bounds__(NT) void window_accumulate_kernel(const f16* __restrict__ pred, float* __restrict__ acc,
                                                              float* __restrict__ counter,
                                                              const int* __restrict__ frames, int S, int Fw, int L,
                                                              int64_t HWC) {
  const int64_t total = (int64_t)S * Fw * HWC;
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
    const int64_t e = i % HWC;
    const int64_t sj = i / HWC;
    const int j = (int)(sj % Fw);
    const int s = (int)(sj / Fw);
    const int f = frames[j];
    if (f >= 0) acc[((int64_t)s * L + f) * HWC + e] += (float)pred[i];
  }
  if (blockIdx.x == 0 && threadIdx.x < Fw && frames[threadIdx.x] >= 0) counter[frames[threadIdx.x]] += 1.0f;
}

__global__ __launch_bounds__(NT) void cfg_ddim_kernel(const float* __restrict__ acc, const float* __restrict__ counter,
                                                     float* __restrict__ lat, f16* __restrict__ lat16, int S, int L,
                                                     int64_t HWC, float g, float sa, float sb, float sap, float sbp) {
  const int64_t total = (int64_t)L * HWC;
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
    const int f = (int)(i / HWC);
    float v;
    if (S == 2) {
      const float c = counter[f];
      const float u = acc[i] / c, cd = acc[total + i] / c;
      v = u + g * (cd - u);
    } else {
      v = acc[i];  // reference quirk: no division by the counter without CFG
    }
    const float x = lat[i];
    const float x0 = sa * x - sb * v;
    const float ep = sa * v + sb * x;
    const float xn = sap * x0 + sbp * ep;
    lat[i] = xn;
    if (lat16) lat16[i] = (f16)xn;
  }
}

template <typename TS>
__global__ __launch_bounds__(NT) void ncfhw_to_nhwc_kernel(const TS* __restrict__ src, f16* __restrict__ dst, int B,
                                                          int C, int F, int64_t HW) {
  const int64_t total = (int64_t)B * F * HW * C;
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int64_t p = r % HW;
    r /= HW;
    const int f = (int)(r % F);
    const int b = (int)(r / F);
    dst[i] = (f16)(float)src[(((int64_t)b * C + c) * F + f) * HW + p];
  }
}

template <typename TD>
__global__ __launch_bounds__(NT) void nhwc_to_ncfhw_kernel(const f16* __restrict__ src, TD* __restrict__ dst, int B,
                                                          int C, int F, int64_t HW, float scale, float shift,
                                                          int clamp01) {
  const int64_t total = (int64_t)B * F * HW * C;
  const int64_t stride = (int64_t)gridDim.x * NT;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < total; i += stride) {
    // i indexes dst (b, c, f, p)
    const int64_t p = i % HW;
    int64_t r = i / HW;
    const int f = (int)(r % F);
    r /= F;
    const int c = (int)(r % C);
    const int b = (int)(r / C);
    float v = (float)src[(((int64_t)b * F + f) * HW + p) * C + c] * scale + shift;
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    dst[i] = (TD)v;
  }
}

// dst[i] = (f16)(scale * src[i] + shift): uint8 image bytes -> fp16 activations (pose renderings, already
// channels-last: (L,H,W,3) uint8 IS the NHWC frame batch)
__global__ __launch_bounds__(NT) void u8_to_f16_kernel(const uint8_t* __restrict__ src, f16* __restrict__ dst, int64_t n,
                                                      float scale, float shift) {
  const int64_t stride = (int64_t)gridDim.x * NT;
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nv; i += stride) {
    const u32x2 raw = ((const u32x2*)src)[i];
    U4H8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned int b = ((e < 4 ? raw.x : raw.y) >> (8 * (e & 3))) & 0xFFu;
      o.e[e] = (f16)(scale * (float)b + shift);
    }
    ((u32x4*)dst)[i] = o.u;
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += NT) dst[i] = (f16)(scale * (float)src[i] + shift);
}

// dst[i] = (uint8) trunc(255 * fp16(clamp(scale * src[i] + shift, 0, 1))): decoded VAE frames (channels-last, already
// the (L,H,W,3) image layout) -> display bytes.  The value is rounded to fp16 first because that is what the
// reference's fp16 pipeline hands to save_videos_grid, which then does (x * 255).astype(uint8) in fp32
// (src/pipelines/pipeline_pose2vid_long.py:123-125, src/utils/util.py:97-98).
__global__ __launch_bounds__(NT) void f16_to_u8_kernel(const f16* __restrict__ src, uint8_t* __restrict__ dst, int64_t n,
                                                      float scale, float shift) {
  const int64_t stride = (int64_t)gridDim.x * NT;
  const int64_t nv = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < nv; i += stride) {
    U4H8 x;
    x.u = ((const u32x4*)src)[i];
    unsigned int lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float c = fminf(fmaxf((float)x.e[e] * scale + shift, 0.f), 1.f);
      const unsigned int b = (unsigned int)((float)(f16)c * 255.0f);
      if (e < 4) lo |= b << (8 * e);
      else hi |= b << (8 * (e - 4));
    }
    ((u32x2*)dst)[i] = u32x2{lo, hi};
  }
  if (blockIdx.x == 0)
    for (int64_t i = (nv << 3) + threadIdx.x; i < n; i += NT) {
      const float c = fminf(fmaxf((float)src[i] * scale + shift, 0.f), 1.f);
      dst[i] = (uint8_t)((float)(f16)c * 255.0f);
    }
}

inline unsigned grid_for(int64_t work) {
  int64_t b = (work + NT - 1) / NT;
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

}  // namespace

extern "C" int anip_linear_small(const float* x, const void* W, const float* bias, float* y, int M, int N, int K,
                                 int silu_in, void* stream) {
  ANIP_REQUIRE(x && W && y, "anip_linear_small: null pointer");
  ANIP_REQUIRE(M >= 1 && M <= 16, "anip_linear_small: M=%d must be in [1,16]", M);
  ANIP_REQUIRE((K & 7) == 0, "anip_linear_small: K=%d must be a multiple of 8", K);
  const int cols_per_block = (NT / 64) * LS_COLS;
  const int blocks = (N + cols_per_block - 1) / cols_per_block;
  // the block stages f(x) of its rows in LDS (64 KB): M * K floats, or — M = 13 .. 16 at K = 1280, a direct UNet forward with a
  // batch of that size — row chunks of what fits, one launch each
  ANIP_REQUIRE((size_t)K * sizeof(float) <= 65536, "anip_linear_small: K = %d floats do not fit in 64 KB of LDS", K);
  const int rows_fit = (int)(65536 / ((size_t)K * sizeof(float)));
  AnipProfScope prof_(ANIP_K_LINEAR_SMALL, (void*)stream);      // ONE bracket per call (hipops.profile pairs brackets with calls)
  for (int mb = 0; mb < M; mb += rows_fit) {
    const int mc = min(rows_fit, M - mb);
    const size_t lds = (size_t)mc * K * sizeof(float);
    const float* xc = x + (int64_t)mb * K;
    float* yc = y + (int64_t)mb * N;
    if (mc <= 2)
      hipLaunchKernelGGL(linear_small_kernel<2>, dim3(blocks), dim3(NT), lds, (hipStream_t)stream, xc, (const f16*)W, bias, yc,
                         mc, N, K, silu_in);
    else if (mc <= 4)
      hipLaunchKernelGGL(linear_small_kernel<4>, dim3(blocks), dim3(NT), lds, (hipStream_t)stream, xc, (const f16*)W, bias, yc,
                         mc, N, K, silu_in);
    else
      hipLaunchKernelGGL(linear_small_kernel<16>, dim3(blocks), dim3(NT), lds, (hipStream_t)stream, xc, (const f16*)W, bias, yc,
                         mc, N, K, silu_in);
  }
  ANIP_LAUNCH_CHECK("anip_linear_small");
  return 0;
}

extern "C" int anip_add(const void* a, const void* b, void* out, int64_t n, void* stream) {
  ANIP_REQUIRE(a && b && out && n > 0, "anip_add: bad arguments");
  ANIP_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "anip_add: pointers must be 16-B aligned");
  const int64_t nvec = n / 8;
  {
    AnipProfScope prof_(ANIP_K_ELEMENTWISE, (void*)stream);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(nvec)), dim3(NT), 0, (hipStream_t)stream, (const f16*)a, (const f16*)b,
                       (f16*)out, nvec, n);
  }
  ANIP_LAUNCH_CHECK("anip_add");
  return 0;
}

extern "C" int anip_window_accumulate(const void* pred, float* acc, float* counter, const int* frames, int S, int Fw,
                                      int L, int64_t HWC, void* stream) {
  ANIP_REQUIRE(pred && acc && counter && frames, "anip_window_accumulate: null pointer");
  ANIP_REQUIRE(S >= 1 && Fw >= 1 && Fw <= NT && L >= Fw, "anip_window_accumulate: bad sizes S=%d Fw=%d L=%d", S, Fw, L);
  {
    AnipProfScope prof_(ANIP_K_ELEMENTWISE, (void*)stream);
    hipLaunchKernelGGL(window_accumulate_kernel, dim3(grid_for((int64_t)S * Fw * HWC)), dim3(NT), 0,
                       (hipStream_t)stream, (const f16*)pred, acc, counter, frames, S, Fw, L, HWC);
  }
  ANIP_LAUNCH_CHECK("anip_window_accumulate");
  return 0;
}

extern "C" int anip_cfg_ddim_step(const float* acc, const float* counter, float* latents, void* latents_f16, int S,
                                  int L, int64_t HWC, float guidance, float sqrt_a, float sqrt_b, float sqrt_a_prev,
                                  float sqrt_b_prev, void* stream) {
  ANIP_REQUIRE(acc && counter && latents, "anip_cfg_ddim_step: null pointer");
  ANIP_REQUIRE(S == 1 || S == 2, "anip_cfg_ddim_step: S must be 1 or 2");
  {
    AnipProfScope prof_(ANIP_K_ELEMENTWISE, (void*)stream);
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((int64_t)L * HWC)), dim3(NT), 0, (hipStream_t)stream, acc, counter,
                       latents, (f16*)latents_f16, S, L, HWC, guidance, sqrt_a, sqrt_b, sqrt_a_prev, sqrt_b_prev);
  }
  ANIP_LAUNCH_CHECK("anip_cfg_ddim_step");
  return 0;
}

extern "C" int anip_ncfhw_to_nhwc(const void* src, int src_f32, void* dst, int B, int C, int F, int64_t HW,
                                  void* stream) {
  ANIP_REQUIRE(src && dst && B > 0 && C > 0 && F > 0 && HW > 0, "anip_ncfhw_to_nhwc: bad arguments");
  const int64_t total = (int64_t)B * C * F * HW;
  AnipProfScope prof_(ANIP_K_ELEMENTWISE, stream);
  if (src_f32)
    hipLaunchKernelGGL(ncfhw_to_nhwc_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const float*)src, (f16*)dst, B, C, F, HW);
  else
    hipLaunchKernelGGL(ncfhw_to_nhwc_kernel<f16>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const f16*)src, (f16*)dst, B, C, F, HW);
  ANIP_LAUNCH_CHECK("anip_ncfhw_to_nhwc");
  return 0;
}

extern "C" int anip_nhwc_to_ncfhw(const void* src, void* dst, int dst_f32, int B, int C, int F, int64_t HW,
                                  float scale, float shift, int clamp01, void* stream) {
  ANIP_REQUIRE(src && dst && B > 0 && C > 0 && F > 0 && HW > 0, "anip_nhwc_to_ncfhw: bad arguments");
  const int64_t total = (int64_t)B * C * F * HW;
  AnipProfScope prof_(ANIP_K_ELEMENTWISE, stream);
  if (dst_f32)
    hipLaunchKernelGGL(nhwc_to_ncfhw_kernel<float>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const f16*)src, (float*)dst, B, C, F, HW, scale, shift, clamp01);
  else
    hipLaunchKernelGGL(nhwc_to_ncfhw_kernel<f16>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream,
                       (const f16*)src, (f16*)dst, B, C, F, HW, scale, shift, clamp01);
  ANIP_LAUNCH_CHECK("anip_nhwc_to_ncfhw");
  return 0;
}

extern "C" int anip_u8_to_f16(const void* src, void* dst, int64_t n, float scale, float shift, void* stream) {
  ANIP_REQUIRE(src && dst && n > 0, "anip_u8_to_f16: bad arguments");
  ANIP_REQUIRE(((uintptr_t)src & 7) == 0 && ((uintptr_t)dst & 15) == 0, "anip_u8_to_f16: src must be 8-B, dst 16-B aligned");
  {
    AnipProfScope prof_(ANIP_K_ELEMENTWISE, stream);
    hipLaunchKernelGGL(u8_to_f16_kernel, dim3(grid_for(n / 8 + 1)), dim3(NT), 0, (hipStream_t)stream, (const uint8_t*)src,
                       (f16*)dst, n, scale, shift);
  }
  ANIP_LAUNCH_CHECK("anip_u8_to_f16");
  return 0;
}

extern "C" int anip_f16_to_u8(const void* src, void* dst, int64_t n, float scale, float shift, void* stream) {
  ANIP_REQUIRE(src && dst && n > 0, "anip_f16_to_u8: bad arguments");
  ANIP_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "anip_f16_to_u8: src must be 16-B, dst 8-B aligned");
  {
    AnipProfScope prof_(ANIP_K_ELEMENTWISE, stream);
    hipLaunchKernelGGL(f16_to_u8_kernel, dim3(grid_for(n / 8 + 1)), dim3(NT), 0, (hipStream_t)stream, (const f16*)src,
                       (uint8_t*)dst, n, scale, shift);
  }
  ANIP_LAUNCH_CHECK("anip_f16_to_u8");
  return 0;
}
