/*
 * aniportrait_hip.h — C ABI of libaniportrait_hip.so (gfx950 / MI355X only).
 *
 * The reference (Zejun-Yang/AniPortrait @ 2024_08_07) has NO native/FFI layer: its hot path reaches
 * the device through stock torch/diffusers ops.  This header is therefore the boundary a maintainer
 * would bind (ctypes) to replace exactly those ops; every entry point names the
 * reference call site(s) whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers owned by the caller
 *    (PyTorch-ROCm allocations); no allocation, no ownership transfer, no exceptions.
 *  - activations are channels-last: an image batch (N,H,W,C) == a token matrix (N*H*W, C), fp16.
 *    1-D parameters (norm gamma/beta, biases) are fp32; weight matrices are fp16, row-major [out][in]
 *    (conv weights [Cout][ky][kx][Cin]).
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream).
 *  - return 0 on success, <0 on error; anip_last_error() returns a thread-local message.
 *  - all accumulation is fp32; outputs are rounded to fp16 once.
 */
#ifndef ANIPORTRAIT_HIP_H
#define ANIPORTRAIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ANIP_ABI_VERSION 15   /* 15 (round 6): anip_gemm act = 2 (quick-GELU); anip_conv_small removed */

int anip_version(void);
const char* anip_last_error(void);
/* device name / gfx arch / CU count of the current device into caller buffers; 0 on success */
int anip_device_info(char* arch, int arch_len, int* num_cu);

/* ---- GroupNorm (+SiLU), per frame ---------------------------------------------------------------
 * replaces InflatedGroupNorm / nn.GroupNorm (+ F.silu):  src/models/resnet.py:21-29,221-222,232-238,
 * src/models/transformer_3d.py:124, src/models/motion_module.py:156, src/models/unet_3d.py:573-574,
 * diffusers ResnetBlock2D / AutoencoderKL norms.
 * x = concat_C(x1 [N,HW,C1], x2 [N,HW,C2]) (x2 may be NULL, C2 = 0: fuses the skip-concat of
 * src/models/unet_3d_blocks.py:697,826 into the norm).  y [N,HW,C1+C2] fp16.
 * ws: fp32 workspace of anip_groupnorm_ws_floats(N,HW,C,G) elements. */
int64_t anip_groupnorm_ws_floats(int N, int64_t HW, int C, int G);
/* 1 if anip_groupnorm runs (N, HW, C, G) as a single kernel launch (one block per (image, group) slab: the 8x8 / 16x16
 * levels), 0 if as a statistics pass + an apply pass; only of interest to per-kernel profilers (bench.py). */
int anip_groupnorm_single_launch(int N, int64_t HW, int C, int G);
int anip_groupnorm(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                   void* y, int N, int64_t HW, int G, float eps, int silu, float* ws, void* stream);
/* the same with statistics over `frames_per_stat` consecutive images (N % frames_per_stat == 0): plain nn.GroupNorm on the
 * (b, c, f, h, w) tensor — ResnetBlock3D norm1 / norm2 and conv_norm_out when use_inflated_groupnorm is False
 * (src/models/resnet.py:161-164,186-193, src/models/unet_3d.py:237-246; configs/inference/inference_v1.yaml).
 * frames_per_stat = 1 is anip_groupnorm. */
int anip_groupnorm_frames(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta,
                          void* y, int N, int64_t HW, int G, float eps, int silu, int frames_per_stat, float* ws,
                          void* stream);

/* ---- LayerNorm over the last dim --------------------------------------------------------------
 * replaces nn.LayerNorm: src/models/attention.py:331-335,352-362, src/models/motion_module.py:228,234.
 * Optional fused temporal positional-encoding add (src/models/motion_module.py:262-277,365-366):
 * y[m] = LN(x[m]) + pe[(m / rows_per_frame) % F]  (pe fp32 [F][C], NULL to disable). */
int anip_layernorm(const void* x, const float* gamma, const float* beta, void* y, int64_t M, int C, float eps,
                   const float* pe, int64_t rows_per_frame, int F, void* stream);

/* ---- GEMM / implicit-GEMM 3x3 convolution on MFMA -----------------------------------------------
 * out[M,N] = epilogue( alpha * A[M,K] @ W[N,K]^T ), fp16 in, fp32 accumulate.
 * replaces every nn.Linear / 1x1 / 3x3 Conv2d on the path: diffusers Attention to_q/k/v/out and
 * FeedForward (src/models/attention.py:323-361, src/models/motion_module.py:122,144,233),
 * InflatedConv3d (src/models/resnet.py:10-18,166-168,195-197,214-216), Upsample3D/Downsample3D convs
 * (src/models/resnet.py:52,72-74,107-109), Transformer3D proj_in/out (src/models/transformer_3d.py:64,93),
 * AutoencoderKL convs. */
typedef struct anip_gemm_params {
  /* A operand: plain mode = row-major [M][K] possibly split over two sources along K
   * (k < K1 -> A, else A2; fuses torch.cat([h, skip], dim=1) into the 1x1 shortcut conv). */
  const void* A;   int64_t lda;
  const void* A2;  int64_t lda2;  int K1;
  const void* W;   int64_t ldw;          /* [N][K] fp16 */
  void* out;       int64_t ldo;          /* [M][N] (GEGLU: [M][N/2]) */
  int out_f32;                           /* 0: fp16 out, 1: fp32 out */
  int M, N, K;
  float alpha;
  const float* bias;                     /* [N] or NULL */
  const float* rowbias; int64_t rows_per_group; int64_t ld_rowbias; /* + rowbias[m / rows_per_group][n]:
                                            time-embedding add (resnet.py:226-230) and the collapsed
                                            length-1 CLIP cross-attention (mutual_self_attention.py:191-205) */
  const void* residual; int64_t ldr;     /* fp16 [M][N] added last, or NULL */
  int act;                               /* 0 none; 1 GEGLU: W/bias rows packed per 32 as [16 x h | 16 x gate]
                                            (N % 128 == 0), out[M][N/2] = h * gelu_erf(gate);
                                            2 quick-GELU x * sigmoid(1.702 x) on (alpha acc + bias + rowbias), ahead of the
                                            residual: fc1 of the CLIP vision tower's MLP (transformers QuickGELUActivation,
                                            reached from pipeline_pose2vid_long.py:379-385) */
  int batch; int64_t strideA, strideW, strideO; /* batched GEMM over blockIdx.y (elements) */
  /* implicit 3x3 convolution (conv != 0): A is an NHWC image batch, M = Nimg*Hout*Wout, K = 9*Cin.
   * conv = 1: tap-major K, W = [Cout][3][3][Cin].  conv = 2 (Cin % 64 == 0): channel-block-major K,
   * W = [Cout][Cin/64][3][3][64] — the nine taps of a 64-channel block are consecutive K-tiles, so the nine shifted
   * re-reads of an input line stay inside the XCD's L2 (the packing the engine uses; hipops.pack_conv3x3).
   * upsample=1 fuses nearest-2x (resnet.py:72-74) into the gather. */
  int conv; int Nimg, Hin, Win, Cin, Hout, Wout, stride, pad, upsample;
  int trans_out;                         /* 1: store the result transposed, out[n*ldo + m] (fp16; bias only):
                                            V^T = (x W_v^T)^T for anip_ref_attention */
  /* split-K for problems with too few output tiles to fill 256 CUs (the 8x8 / 16x16 levels: M = 2048, K up to
   * 23040; since round 4 also 64 <= M < 1024 under K >= 1024 — the ReferenceNet's M = 128 / 512 — with up to 32
   * slices): anip_gemm_workspace_bytes(p) > 0 means anip_gemm wants that many bytes of device scratch in
   * `workspace` (fp32 partial tiles [split][M][N], reduced with the whole epilogue by a second kernel);
   * 0 means no workspace is needed.  The library never allocates. */
  void* workspace; int64_t workspace_bytes;
  /* head_dim > 0: head-major output — column n of row m goes to out[(n / head_dim) * M * head_dim + m * head_dim +
   * n % head_dim] (fp16, non-transposed, batch 1, head_dim % 8 == 0): each head's [M][head_dim] block is contiguous.
   * The K projection of anip_ref_attention (to_k of src/models/mutual_self_attention.py:147-165): a 64-key tile of one
   * head is then one contiguous run instead of 2 d-byte pieces at a C-byte stride. */
  int head_dim;
} anip_gemm_params;
int64_t anip_gemm_workspace_bytes(const anip_gemm_params* p);
int anip_gemm(const anip_gemm_params* p, void* stream);

/* ---- fused GEGLU feed-forward (the engine's default at C = 320; ANIP_FUSED_FFN=0 selects two anip_gemm calls) -------
 * out[M][C] = residual + b2 + W2 · ((x W1v^T + b1v) * gelu_erf(x W1g^T + b1g)):  diffusers FeedForward("geglu") of
 * src/models/attention.py:361 / src/models/motion_module.py:233 plus the block's residual add, without the M x 4C
 * intermediate ever leaving the CU.  x [M][C] fp16; w1p [8C][C] fp16 and b1p [8C] fp32 packed per 32 rows as
 * [16 value | 16 gate] (the GEGLU packing of anip_gemm); w2 [C][4C] fp16; b2 [C] fp32 or NULL; residual [M][C]
 * fp16 or NULL.  Only C = 320 is built. */
int anip_ffn_geglu(const void* x, const void* w1p, const float* b1p, const void* w2, const float* b2,
                   const void* residual, void* out, int64_t M, int C, void* stream);
/* same with the block's LayerNorm inside: out = residual + FeedForward(LayerNorm(x; gamma, beta, eps)) — norm3 / ff_norm of
 * src/models/attention.py:436-445 / src/models/motion_module.py:256-257 applied while the x tile is staged (x holds the RAW
 * rows; the engine passes residual = x).  gamma, beta [C] fp32.  Only C = 320 is built. */
int anip_ffn_geglu_ln(const void* x, const float* gamma, const float* beta, float eps, const void* w1p, const float* b1p,
                      const void* w2, const float* b2, const void* residual, void* out, int64_t M, int C, void* stream);

/* ---- PoseGuider stem: direct convolution + BatchNorm2d(+ReLU) --------------------------------------
 * replaces the nn.Conv2d / nn.BatchNorm2d / nn.ReLU stacks of src/models/pose_guider.py:19-85 whose channel
 * counts (3, 16, 32) or 4x4-stride-2 windows do not fit the implicit-GEMM kernel.
 * anip_conv_direct: x [N,H,W,Cin] fp16; wp [ksize*ksize*Cin][Cout8] fp16 with Cout8 = Cout rounded up to 8
 * (tap-major, output channel fastest, zero padded); bias fp32 [Cout] or NULL; residual [N,Ho,Wo,Cout] fp16 or
 * NULL (added before the optional ReLU: the pose-feature add of src/models/unet_3d.py:485-486 on conv_in, which —
 * like AutoencoderKL post_quant_conv / decoder.conv_in / encoder.conv_in — also runs here); y [N,Ho,Wo,Cout] fp16,
 * Ho = (H + 2 pad - ksize) / stride + 1.  ksize 1..5, stride 1 or 2; ksize*ksize*Cin*Cout8*2 <= 64 KB.
 * anip_batchnorm: x,y [M][C] fp16 (channels-last rows = N*H*W); running_mean/var NULL -> batch statistics
 * with biased variance (module in training mode: scripts/pose2vid.py:77 never calls .eval()), else the
 * running statistics.  ws: fp32 workspace of anip_batchnorm_ws_floats(M, C) elements. */
int anip_conv_direct(const void* x, const void* wp, const float* bias, const void* residual, void* y, int N, int H,
                     int W, int Cin, int Cout, int ksize, int stride, int pad, int relu, void* stream);
int64_t anip_batchnorm_ws_floats(int64_t M, int C);
int anip_batchnorm(const void* x, const float* gamma, const float* beta, const float* running_mean,
                   const float* running_var, void* y, int64_t M, int C, float eps, int relu, float* ws,
                   void* stream);

/* ---- reference attention (spatial) ---------------------------------------------------------------
 * replaces attn1 of the hacked TemporalBasicTransformerBlock in read mode
 * (src/models/mutual_self_attention.py:147-186 -> diffusers AttnProcessor2_0 -> SDPA) and plain
 * self-attention of the ReferenceNet (write mode, :137-146).
 * For frame n, head h: O = softmax(Q K^T / sqrt(d)) V with keys = [self tokens of frame n]
 * ++ [reference tokens of sample ref_index[n]] (ref_index[n] < 0: self only, the CFG-unconditional
 * frames).  q,k: [Nf*T][ld] fp16 with head h at column h*d; vt: V transposed [heads*d][ldvt] with
 * token (n*T + t) at column n*T+t; kref [Nref*T][ldkr], vtref [heads*d][ldvtr]; out [Nf*T][ldo].  * k_head_stride / kref_head_stride (elements; 0 = d): distance between two heads' K data.  Token-major K (a column
 * slice of a [tokens][heads*d] matrix): d, with ldk = the row stride.  Head-major K (anip_gemm head_dim = d output):
 * tokens_total * d, with ldk = d — a 64-key tile of a head is then one contiguous 128 d-byte run. */
int anip_ref_attention(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                       const void* kref, int64_t ldkr, const void* vtref, int64_t ldvtr,
                       const int* ref_index, void* out, int64_t ldo,
                       int Nf, int T, int heads, int d, float scale, int64_t k_head_stride, int64_t kref_head_stride, void* stream);
/* The same with `flags`.  ANIP_ATTN_Q_LOG2_SCALED: q has been multiplied by scale * log2(e) by its producer (the `alpha` of
 * the to_q projection anip_gemm: still one fp16 rounding of the fp32 accumulator), so q.k is the base-2 exponent of the
 * softmax and `scale` is ignored.  It is what the engine passes: for T % 256 == 0 and d in {40, 80, 160} it selects the
 * round-4 kernel (csrc/attn_dma.hip: LDS-DMA K / V^T ring, 256 queries per workgroup, running max folded into the score
 * MFMA at d = 40); every other shape runs the first kernel with the flag honoured.  The base-2 exponents must stay
 * below 6e4 in magnitude (the folded running max is carried as an fp16 hi/lo pair).
 * ANIP_ATTN_FRAME_MOD(m), m in [1, 32767]: q / k / vt hold m frames and frame n of the Nf (a multiple of m) attends with the
 * tokens of frame n % m — the two classifier-free-guidance halves of a denoising call enter the first reference attention with
 * identical self tokens (same latents, pose features and timestep; they differ in their reference index only), so the
 * engine computes everything in front of it once (engine.unet_forward, cfg_shared_input).  Output and ref_index stay per
 * frame n. */
#define ANIP_ATTN_Q_LOG2_SCALED 1
#define ANIP_ATTN_FRAME_MOD(m) ((int)(m) << 16)
int anip_ref_attention_ex(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* vt, int64_t ldvt,
                          const void* kref, int64_t ldkr, const void* vtref, int64_t ldvtr,
                          const int* ref_index, void* out, int64_t ldo,
                          int Nf, int T, int heads, int d, float scale, int64_t k_head_stride, int64_t kref_head_stride,
                          int flags, void* stream);

/* ---- temporal self-attention ------------------------------------------------------------------
 * replaces VersatileAttention (src/models/motion_module.py:351-388): for every (b, pixel t, head):
 * attention over the F frames.  qkv [(b F) T][3C] fp16 (q|k|v), out [(b F) T][C]. */
int anip_temporal_attention(const void* qkv, void* out, int B, int F, int T, int heads, int d, float scale,
                            void* stream);

/* ---- fused front half of a motion-module attention block ---------------------------------------------
 * replaces norms[i] -> PositionalEncoding -> to_q / to_k / to_v -> VersatileAttention's softmax(q k^T / sqrt d) v
 * (src/models/motion_module.py:236-259 TemporalTransformerBlock.forward, :283-296, :351-388) in one launch; the block's
 * to_out + residual stays an anip_gemm call.
 *   x        [(b F) T][C] fp16   hidden states h of the block
 *   gamma    [C] fp32            norms[i].weight
 *   beta_pe  [F][C] fp32         norms[i].bias + pos_encoder.pe[frame]   (bias alone, repeated, when there is no encoder)
 *   w_packed [3C][C] fp16        per head PAIR p = 0..3: rows 240 p + {0..79: to_q rows 80 p.., 80..159: to_k, 160..239: to_v}
 *   out      [(b F) T][C] fp16   attention output (token-major, heads concatenated) = input of to_out
 * Built for F = 16, C = 320, heads = 8 (d = 40), T % 8 == 0 — anip_temporal_qkv_attention_supported tells (1 / 0);
 * every other shape stays on anip_layernorm + anip_gemm + anip_temporal_attention.  scale = d^-1/2 of the reference. */
/* ---- row-stationary projections at C = 320 (csrc/tblock.hip) ---------------------------------------------------
 * The 64x64 level's K = 320 Linears are bound by memory traffic; these keep a wave's 32 rows in registers through their
 * normalisation AND their projections.  anip_rowgemm320_supported: C == 320, M % 128 == 0, rows_per_frame % 32 == 0 (or 0).
 *
 * anip_ln_qkv_projection — norm1 -> attn1.to_q / to_k / to_v of a spatial (Temporal)BasicTransformerBlock
 *   (src/models/attention.py:383-401, consumed by src/models/mutual_self_attention.py:147-186) in the three layouts
 *   anip_ref_attention reads: q [M][C] token-major multiplied by q_alpha, k head-major [heads][M][d], v^T [C][ldvt].
 *   w_qkv [3C][C] fp16 = rows of to_q, then to_k, then to_v.  Replaces anip_layernorm + three anip_gemm calls.
 *
 * anip_groupnorm_scale_shift + anip_affine_linear320 — Transformer3DModel / temporal transformer  norm -> proj_in
 *   (src/models/transformer_3d.py:128-139, src/models/motion_module.py:185-204): the per-frame GroupNorm statistics are
 *   finalised into scale_shift [N][C][2] fp32 = (rstd gamma, beta - mean rstd gamma) and applied to the rows on their way
 *   into the 1x1 convolution: out = (x * scale[frame] + shift[frame]) W^T + bias.  Replaces anip_groupnorm + anip_gemm. */
int anip_rowgemm320_supported(int64_t M, int C, int64_t rows_per_frame);
int anip_ln_qkv_projection(const void* x, const float* gamma, const float* beta, float eps, const void* w_qkv, void* q,
                           float q_alpha, void* k_head_major, void* vt, int64_t ldvt, int64_t M, int C, int heads,
                           void* stream);
int anip_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float* scale_shift, int N, int64_t HW,
                               int C, int G, float eps, float* ws, void* stream);
int anip_affine_linear320(const void* x, const float* scale_shift, int64_t rows_per_frame, const void* w, const float* bias,
                          void* out, int64_t M, int C, void* stream);

int anip_temporal_qkv_attention_supported(int F, int T, int C, int heads);
int anip_temporal_qkv_attention(const void* x, const float* gamma, const float* beta_pe, const void* w_packed, void* out,
                                int B, int F, int T, int C, int heads, float eps, float scale, void* stream);

/* ---- row softmax fp32 -> fp16 (VAE mid-block attention, diffusers Attention upcast_softmax) ---- */
int anip_softmax_rows(const float* s, void* p, int64_t rows, int cols, void* stream);

/* ---- tiny dense layers (M <= 16 rows): time embedding MLP, time_emb_proj, collapsed attn2 -------
 * y[m][n] = sum_k f(x[m][k]) * W[n][k] + bias[n], f = SiLU if silu_in.  x,y fp32; W fp16.
 * 1 <= M <= 16, K % 8 == 0, and f(x) is staged in LDS: M * K * 4 bytes <= 64 KB (error otherwise).
 * (src/models/unet_3d.py:463-469, src/models/resnet.py:226-227) */
int anip_linear_small(const float* x, const void* W, const float* bias, float* y, int M, int N, int K,
                      int silu_in, void* stream);

/* ---- elementwise -------------------------------------------------------------------------------- */
/* out = a + b (fp16), pose-feature adds (src/models/unet_3d.py:508-510) */
int anip_add(const void* a, const void* b, void* out, int64_t n, void* stream);
/* acc[s][frames[j]] += pred[s][j], counter[frames[j]] += 1 for one context window
 * (src/pipelines/pipeline_pose2vid_long.py:546-548).  pred fp16 [S][Fw][HWC], acc fp32 [S][L][HWC]. */
int anip_window_accumulate(const void* pred, float* acc, float* counter, const int* frames, int S, int Fw, int L,
                           int64_t HWC, void* stream);
/* CFG combine + DDIM v-prediction update (pipeline_pose2vid_long.py:551-559 + DDIMScheduler.step):
 * eps = acc/counter ; v = u + g (c - u) (S == 2) or acc (S == 1, not divided: reference quirk) ;
 * x0 = sa*x - sb*v ; e = sa*v + sb*x ; x <- sap*x0 + sbp*e.  latents fp32 [L][HWC] in place;
 * also writes fp16 copy for the next UNet call. */
int anip_cfg_ddim_step(const float* acc, const float* counter, float* latents, void* latents_f16, int S, int L,
                       int64_t HWC, float guidance, float sqrt_a, float sqrt_b, float sqrt_a_prev,
                       float sqrt_b_prev, void* stream);
/* layout/dtype conversion between (B,C,F,H,W) [fp32 or fp16] and channels-last frames (B*F,H,W,C) fp16 */
int anip_ncfhw_to_nhwc(const void* src, int src_f32, void* dst, int B, int C, int F, int64_t HW, void* stream);
int anip_nhwc_to_ncfhw(const void* src, void* dst, int dst_f32, int B, int C, int F, int64_t HW, float scale,
                       float shift, int clamp01, void* stream);

/* dst[i] = (fp16)(scale * src[i] + shift), src uint8: the numpy path of VaeImageProcessor.preprocess on the pose
 * renderings (src/pipelines/pipeline_pose2vid_long.py:445-452: values 2 v - 1 in [-1, 509], no /255), done on
 * the device on the uploaded bytes; (L,H,W,3) uint8 is already the channels-last frame batch. */
int anip_u8_to_f16(const void* src, void* dst, int64_t n, float scale, float shift, void* stream);
/* dst[i] = (uint8) trunc(255 * fp16(clamp(scale * src[i] + shift, 0, 1))), src fp16: decoded frames (channels-last =
 * the (L,H,W,3) image layout) -> display bytes on the device; replaces the host-side `(x / 2 + 0.5).clamp(0, 1)` ...
 * `(x * 255).numpy().astype(np.uint8)` of src/pipelines/pipeline_pose2vid_long.py:123-125 + src/utils/util.py:97-98
 * (same values: the intermediate is rounded to fp16 as in the reference's fp16 run) */
int anip_f16_to_u8(const void* src, void* dst, int64_t n, float scale, float shift, void* stream);

/* ---- per-kernel timing with HIP events (bench.py's roofline leg) -------------------------------------
 * When enabled, every entry point brackets each kernel launch with hipEventRecord on the launch stream.
 * anip_profile_collect synchronises, sums elapsed time and launch count per kernel family
 * (ANIP_K_* index), writes up to max_ids entries, then clears the records.  Off by default: the timed
 * region of bench.py runs without events. */
enum {
  ANIP_K_GEMM = 0,        /* gemm_kernel<false>: Linear / 1x1 conv / batched GEMM */
  ANIP_K_CONV3X3 = 1,     /* gemm_kernel<true>: implicit-GEMM 3x3 convolution */
  ANIP_K_GN_STATS = 2,
  ANIP_K_GN_APPLY = 3,
  ANIP_K_LAYERNORM = 4,
  ANIP_K_REF_ATTN = 5,
  ANIP_K_TEMPORAL_ATTN = 6,
  ANIP_K_SOFTMAX = 7,
  ANIP_K_CONV_SMALL = 8,
  ANIP_K_LINEAR_SMALL = 9,
  ANIP_K_ELEMENTWISE = 10, /* add / window accumulate / cfg+ddim / layout conversions */
  ANIP_K_BATCHNORM = 11,  /* bn_stats + bn_finalize + bn_apply (PoseGuider) */
  ANIP_K_COUNT = 12
};
int anip_profile_enable(int on);
int anip_profile_collect(int max_ids, int64_t* launches, double* total_ms);
/* same, and additionally the individual records in launch order: rec_kid[i] / rec_ms[i] for the first
 * max_records brackets, *n_records = number of brackets seen (per-shape tables of bench.py) */
int anip_profile_collect_records(int max_ids, int64_t* launches, double* total_ms, int64_t max_records,
                                 int* rec_kid, float* rec_ms, int64_t* n_records);
const char* anip_profile_kernel_name(int kid);

#ifdef __cplusplus
}
#endif
#endif
