// Argument block shared by the two generations of the reference-attention kernel (attention.hip, attn_dma.hip).
#pragma once
#include "common.h"

struct RefAttnArgs {
  const f16* q; int64_t ldq;
  const f16* k; int64_t ldk;
  const f16* vt; int64_t ldvt;
  const f16* kref; int64_t ldkr;
  int64_t k_hs, kr_hs;   // elements between two heads' K data (token-major: d; head-major: tokens * d)
  const f16* vtref; int64_t ldvtr;
  const int* ref_index;
  f16* out; int64_t ldo;
  int T, heads;
  float scale_log2e;
  int vt_vec_ok, vtref_vec_ok;
  int frame_mod;   // > 0: frame n reads q / k / v^T of frame n % frame_mod (the two CFG halves share their self tokens); 0: of frame n
};

// attn_dma.hip: 1 if (a, d) is a shape the LDS-DMA kernel takes (the caller has checked flags), launches it; 0 = not taken
int anip_ref_attn_dma_try(const RefAttnArgs& a, int Nf, int d, hipStream_t stream);
