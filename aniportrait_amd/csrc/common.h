// Shared device/host helpers for libaniportrait_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/aniportrait_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void anip_set_error(const char* fmt, ...);
// hipFuncAttributeMaxDynamicSharedMemorySize >= bytes for `kernel` on the CURRENT device, cached per (kernel, device ordinal)
// for any ordinal (api.cpp); 0 on success
int anip_raise_lds_limit(const void* kernel, int bytes);
// CU count of the current device, cached per ordinal (-1 if the query fails)
int anip_cu_count();

#define ANIP_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      anip_set_error(__VA_ARGS__);     \
      return -1;                       \
    }                                  \
  } while (0)

#define ANIP_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      anip_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return -2;                                                             \
    }                                                                        \
  } while (0)

// per-launch HIP-event bracket (no-op unless anip_profile_enable(1)); see api.cpp
void anip_prof_begin(int kid, hipStream_t s);
void anip_prof_end(int kid, hipStream_t s);
struct AnipProfScope {
  int kid;
  hipStream_t s;
  AnipProfScope(int k, void* st) : kid(k), s((hipStream_t)st) { anip_prof_begin(kid, s); }
  ~AnipProfScope() { anip_prof_end(kid, s); }
};

// NOTE: native ext_vector types only — selects on HIP's struct uint4 are lowered through scratch.
union U4H8 {
  u32x4 u;
  f16x8 h;
  f16 e[8];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf-GELU for GEMM epilogues (VALU-bound there): erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below
// the fp16 rounding of the result), branch-free, one v_rcp + one v_exp; for x < 0 the factor 1 + erf is formed
// as poly * exp directly, so the tail keeps its relative accuracy.
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float pe = poly * __builtin_amdgcn_exp2f(z * z * -1.4426950408889634f);
  const float one_plus_erf = x >= 0.f ? 2.0f - pe : pe;
  return 0.5f * x * one_plus_erf;
}

// erf-GELU on the FMA pipe only (round 3): erf(z) = z P(z^2) on |z| <= 3, P of degree 8 fitted (Lawson-weighted least
// squares, constrained to erf(3) = 1) to |error| <= 2.9e-5, clamped outside — 14 full-rate VALU operations instead of
// gelu_fast_f's 15 + two transcendentals (v_rcp, v_exp: a quarter of the VALU rate each).  GEGLU epilogues are VALU-bound:
// 64 activations per lane per 256 x 256 tile (a fifth of the K = 640 ff-in GEMM's time), 8 per lane per chunk in the fused
// FFN.  The absolute error of the activation is <= 0.5 |x| 2.9e-5, an order below the fp16 rounding of the stored product.
__device__ __forceinline__ float gelu_poly_f(float x) {
  const float z = __builtin_amdgcn_fmed3f(x * 0.70710678118654752440f, -3.0f, 3.0f);
  const float u = z * z;
  float p = 4.4811992553e-08f;
  p = fmaf(p, u, -2.1063673519e-06f);
  p = fmaf(p, u, 4.3721626158e-05f);
  p = fmaf(p, u, -5.3454680828e-04f);
  p = fmaf(p, u, 4.3555246836e-03f);
  p = fmaf(p, u, -2.5458836285e-02f);
  p = fmaf(p, u, 1.1165953196e-01f);
  p = fmaf(p, u, -3.7576797702e-01f);
  p = fmaf(p, u, 1.1283874015e+00f);
  const float hx = 0.5f * x;
  return fmaf(hx, p * z, hx);                // 0.5 x (1 + erf(x / sqrt 2))
}

// lanes fr and fr ^ 8 of a 16-lane row trade values (v_mov_dpp row_ror:8; the bank mask selects which half is written):
//   half_swap_hi(keep, give): lanes 0-7 of each row keep `keep`, lanes 8-15 receive `give` of the lane 8 below
//   half_swap_lo(keep, give): lanes 8-15 keep `keep`, lanes 0-7 receive `give` of the lane 8 above
__device__ __forceinline__ float half_swap_hi(float keep, float give) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(keep), __float_as_uint(give), 0x128, 0xF, 0xC, false));
}
__device__ __forceinline__ float half_swap_lo(float keep, float give) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(keep), __float_as_uint(give), 0x128, 0xF, 0x3, false));
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
