/* mvd_hip.h -- C ABI of the MI355X-native (gfx950) MVD-Fusion denoising hot path.
 *
 * The reference (zhizdev/mvdfusion) is pure Python/PyTorch and has no FFI of its own; this ABI is the new
 * boundary *below* the reference's config-driven classes.  Each entry point replaces a
 * cluster of torch ops on the per-DDIM-step path; the reference site is cited next to it (paths relative to the
 * reference root).  The Python mirror classes in mvdfusion_amd/ (GridAttn, UNetModel, ViewFusion, DDIMSampler)
 * bind these symbols with ctypes -- see INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr()); the library never
 *     allocates, frees or retains memory beyond a call;
 *   - every call only ENQUEUES work on `stream` (a hipStream_t); no host synchronisation => calls are capturable
 *     in a hipGraph (mvd_graph_*);
 *   - return 0 on success, <0 on error; the message is available from mvd_last_error(); nothing throws;
 *   - activations are fp32, channels-last: an image tensor is (B, H, W, C) == a row-major (B*H*W, C) matrix;
 *   - GEMM-shaped math runs on 16-bit MFMA (fp16 by default, bf16 in the -DMVD_OPERAND_BF16 build; same rate).
 *     `prec` selects MVD_PREC_X1 (= one product per operand pair) or MVD_PREC_X3 (= operands split
 *     x = hi + lo, three products hi*hi + hi*lo + lo*hi, fp32 accumulate: ~2^-22 (fp16) / ~2^-17 (bf16) relative
 *     operand error; the 50-step stochastic trajectory amplifies operand error by ~10^3, so the split is what keeps it
 *     within the 1e-3 latent-RMSE budget).
 *   - Operand range (fp16 flavour): every GEMM / attention operand is stored as x ~= hi + lo in fp16.  For
 *     6e-5 <= |x| < 65504 the relative operand error is <= 2^-22; below that the ABSOLUTE resolution is 2^-25 (fp16
 *     subnormals), i.e. a tensor whose values are all << 1e-3 loses relative precision; |x| >= 65504 overflows to inf and
 *     the outputs become non-finite (no silent saturation).  Weights are pre-scaled by a power of two at pack time
 *     (mvd_pack_*, undone exactly through acc_scale) so they always sit in the normal range.  The host mirrors check the
 *     final latents / images once per sample (mvdfusion_amd/hip.py: check_finite) and raise FloatingPointError; the bf16
 *     flavour (libmvd_hip_bf16.so, precision "bf16x3") has fp32's exponent range at ~2^-16 relative operand error.
 *   - one host thread per process / GPU (matches the reference's mp.spawn model, demo.py:208).
 */
#ifndef MVD_HIP_H
#define MVD_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* mvd_stream_t; /* hipStream_t */

#define MVD_VERSION 100
#define MVD_PREC_X1 1   /* one product per operand pair (hi only) */
#define MVD_PREC_X3 3 /* hi*hi + hi*lo + lo*hi */
#define MVD_PREC_X4 4     /* all four partial products of the (hi+lo)(hi+lo) split: fp32-class products */

int mvd_version(void);
const char* mvd_last_error(void);
/* MFMA operand element type this library was built for: 0xf16 (fp16, default) or 0xbf16 (-DMVD_OPERAND_BF16).
 * Every "split planes" / packed-weight buffer holds that type; the two flavours are separate .so files. */
int mvd_operand_format(void);

/* ------------------------------------------------------------------------------------------------
 * Weight packing (once per model load).  Packed image: [K/32][N/16][hi image | lo image] (2 KiB micro-tiles of 16 n x 32 k); an
 * image (1 KiB) is the 16x16x32 MFMA B fragment as a wave holds it: lane l = (n & 15) + 16 * ((k & 31) >> 3) owns the 8 elements
 * k & 7 = 0..7 at byte 16 l -- a granule of the LDS-DMA is one contiguous KiB and the fragment read from LDS is lane-contiguous.
 * K padded to 32, N padded to 16 with zeros.  Bytes = mvd_packed_weight_bytes(N, K).
 * Replaces nothing in the reference (its weights stay fp32 nn.Parameters); the Python mirrors keep the
 * fp32 parameters under the reference's state_dict keys and pack on first use. */
size_t mvd_packed_weight_bytes(int N, int K);
/* w: (N, K) row-major fp32 with leading dimension ldw.  geglu != 0 interleaves value/gate row blocks of 16
 * (rows [0,N/2) = value, [N/2,N) = gate, attention.py:43-44) so a GEMM tile holds matching value/gate columns. */
/* scale: power of two applied to the weights before the split (keeps the low part out of fp16's subnormal range);
 * the GEMM undoes it exactly through mvd_gemm_desc.acc_scale = 1/scale. */
int mvd_pack_linear_weight(const float* w, int N, int K, int ldw, int geglu, float scale, void* packed,
                           mvd_stream_t stream);
/* The same image from the TRANSPOSED source: wt is (K, N) row-major with leading dimension ldw, the packed weight is its transpose
 * (N, K) -- the dgrad weight W^T of a Linear packed straight from the parameter, without a transposed copy. */
int mvd_pack_linear_weight_t(const float* wt, int N, int K, int ldw, float scale, void* packed, mvd_stream_t stream);
/* w: (Cout, Cin, 3, 3) fp32 (nn.Conv2d layout).  Packed K index = ((ci/32)*9 + ky*3+kx)*32 + ci%32: the nine taps of one
 * 32-channel block are consecutive k-tiles, so the implicit-GEMM kernel re-reads a pixel's 128-byte line back to back. */
int mvd_pack_conv3x3_weight(const float* w, int Cout, int Cin, int cin_pad, float scale, void* packed,
                            mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution with fused epilogue.
 *   nn.Linear                      external/sd1/ldm/modules/attention.py:161-168, 41, 60; mvdfusion/attention.py:100,114;
 *                                  view_attn_efficient2.py:52-61,83,158,167 (timm Attention/Mlp linears)
 *   nn.Conv2d 1x1                  attention.py:245,259; openaimodel.py:241
 *   nn.Conv2d 3x3 (s1, s2, after nearest-2x upsample)   openaimodel.py:107,116,151,204,229-231; unet.py:323,499 */
#define MVD_A_DENSE 0   /* A: (M, K) row-major planes, leading dim lda (elements) */
#define MVD_A_CONV3X3 1 /* A: NHWC (B, Hin, Win, Cin) planes, pad 1; M = B*Hout*Wout, K = 9*Cin */

#define MVD_EPI_STORE 0 /* out[m,n] = res[m,n] + colscale[n] * act(acc + bias[n] + bias_b[m / rows_per_batch, n]) */
#define MVD_EPI_GEGLU 1 /* out[m,j] = (acc_v + bias[j]) * gelu(acc_g + bias[N/2 + j]);  out has N/2 columns */
#define MVD_EPI_QKV 2   /* route N = 3*heads*dhead columns to the attention operand planes (see mvd_attention) */

#define MVD_ACT_NONE 0
#define MVD_ACT_GELU 1 /* exact erf GELU (nn.GELU(), F.gelu) */
#define MVD_ACT_SILU 2
#define MVD_ACT_QUICKGELU 3 /* x * sigmoid(1.702 x): the MLP activation of OpenAI CLIP's vision transformer */

#define MVD_GEMM_TILES 5 /* tile shapes of mvd_gemm_desc.cfg */
#define MVD_GEMM_LOOPS 8 /* k-loop variants of mvd_gemm_desc.cfg (0 ... 7; 3 removed) */
#define MVD_GEMM_CFG_STRIDE 32 /* cfg = 1 + MVD_GEMM_CFG_STRIDE * tile + 2 * loop + order */
#define MVD_B_PACKED 0   /* B: weight image of mvd_pack_linear_weight / mvd_pack_conv3x3_weight */
#define MVD_B_PLANES 1   /* B: (N, ldb) row-major split planes (an activation), N % 16 == 0 */

/* one weight (or any read-only operand) a later launch of the step will read: mvd_gemm_desc.pf_items */
typedef struct mvd_prefetch_item_s {
  const void* ptr;
  unsigned long long bytes;
  int start_after;      /* host bookkeeping (launch index of the host kernel); not read by the device */
  int consumer;         /* host bookkeeping (launch index of the consumer); not read by the device */
} mvd_prefetch_item;

typedef struct mvd_gemm_desc {
  int M, N, K;      /* logical sizes; N % 16 == 0 after padding of the packed weight, K as packed (multiple of 32) */
  /* A operand: activations in the SPLIT-PLANES format, produced by the previous kernel (mvd_groupnorm_nhwc,
   * mvd_layernorm, mvd_attention, a GEMM epilogue, mvd_split_planes ...):  x ~= hi + lo as bf16; a (rows, lda) matrix
   * is stored per row as lda/32 blocks of [32 x hi | 32 x lo] (128 contiguous bytes per row and 32-element k-block,
   * the unit the kernel's LDS-DMA moves).  Same bytes as fp32.  128-byte aligned. */
  const void* A;
  int lda;          /* elements per row; multiple of 32 */
  int a_mode;       /* MVD_A_* */
  /* conv geometry (a_mode == MVD_A_CONV3X3) */
  int B, Hin, Win, Cin, Hout, Wout, stride, upsample; /* upsample: input is nearest-2x upsampled before the conv */
  int no_pad_tl;         /* 1: no zero padding on the top / left edge, i.e. F.pad(x, (0,1,0,1)) + conv(stride 2, padding 0) of
                            the VAE Downsample (diffusionmodules/model.py:72-76); taps past the bottom / right edge read zeros */
  const void* Wp;   /* B operand: packed weight (mvd_pack_*), or -- b_mode == MVD_B_PLANES -- an (N, ldb) activation matrix in split
                       planes, i.e. out = A B^T of two activation tensors (the VAE mid-block attention: Q K^T and P V,
                       external/sd1/ldm/modules/diffusionmodules/model.py:184-199) */
  int b_mode;       /* MVD_B_* */
  int ldb;          /* elements per row of a planes B operand; multiple of 32 */
  float acc_scale;  /* accumulator scale = 1 / (pack scale); 0 is treated as 1 */
  int prec;         /* MVD_PREC_* */
  /* epilogue */
  int epi;          /* MVD_EPI_* */
  int act;          /* MVD_ACT_* */
  float* out;       /* fp32 output or NULL */
  int ldo;
  void* out_sp;     /* optional split-planes output (feeds the next GEMM's A operand) */
  int ldp;          /* elements per row of out_sp; multiple of 32 */
  int n_store;      /* columns actually stored (<= N; lets N be padded to 16, e.g. the 5-channel UNet head) */
  const float* bias;     /* [N] or NULL */
  const float* bias_b;   /* [M / rows_per_batch][ldbb] or NULL (per-view vector, e.g. the kv_len==1 cross-attention) */
  int rows_per_batch;
  int ldbb;              /* row stride of bias_b in floats (multiple of 4); 0 = N */
  const float* colscale; /* [N] or NULL (adaLN gate) */
  const float* res;      /* [M][ldr] or NULL */
  int ldr;
  /* MVD_EPI_QKV */
  void *q_hi, *q_lo, *k_hi, *k_lo, *vt_hi, *vt_lo;
  int heads, dhead, L, Lpad; /* rows m = b*L + token */
  float qscale;              /* dhead^-0.5 * log2(e), applied to q in fp32 before the split: mvd_attention computes its
                                softmax with exp2 on the raw q.k products */
  /* split-K: 1 = none; >1 = that many K slices; 0 = choose automatically (fills the 256 CUs when the tile grid
   * is small or divides badly over them -- a small time model, gemm.hip: choose_splits).  Partial fp32 slabs
   * (splitk*M*N) go to `workspace`; a second kernel sums them in slice order 0..splitk-1 and applies the
   * epilogue; without a workspace (or when it is too small: workspace_elems) the GEMM runs unsplit. */
  int splitk;
  float* workspace;
  size_t workspace_elems;
  /* kernel configuration: 0 = built-in heuristic; otherwise cfg = 1 + MVD_GEMM_CFG_STRIDE * tile + 2 * loop + order with
   *   tile : 0 = 64x64 (4 waves)  1 = 128x128 (8 waves)  2 = 128x80 (4 waves)  3 = 64x80 (4 waves)  4 = 128x160 (8 waves);
   *          tiles >= 2 (the 80-column family for N = 320 * k: no N padding, 256 workgroups at M = 8192, N = 320) serve
   *          MVD_EPI_STORE only (the GEGLU / QKV epilogues walk a wave tile in 32-column blocks)
   *   loop : 0 = plain two-buffer loop, 1 = register-pipelined loop (two k-tiles in LDS, MFMA fragments double-buffered in
   *          registers), 2 = staggered (8-wave tiles only: three k-tiles in LDS, the two wavefronts of a SIMD half an
   *          iteration apart, so one issues LDS-DMA / fragment reads while the other runs MFMAs), 4 = the register-pipelined loop over a ring of up to 4 k-tiles in LDS, 5 = over a
   *          ring of up to 8 (4-wave tiles): one workgroup per CU keeps 3 / 7 k-tiles of operands in flight, for the small grids
   *          of the low-resolution levels whose k-loop is otherwise one DMA round trip per k-tile, 6 = the input-patch kernel for
   *          stride-1 padded 3x3 convolutions (tiles 1, 2, 4): the tile's pixels + halo are staged once per 32-channel block and the
   *          nine taps read shifted slots of that patch (4-6x less A traffic into LDS), 7 = the wave-specialised kernel (tiles 1, 2,
   *          4; tile 1 with every epilogue, 2 and 4 MVD_EPI_STORE): four consumer wavefronts (fragment reads + MFMAs) and four loader wavefronts (all LDS-DMAs) per
   *          workgroup;
   *          3 = removed (a four-buffer staggered loop, round 3: never the fastest on any shape; the register-staged deliveries 8 / 9 and
   *          the persistent role-split kernel 10 of rounds 4 / 5 went the same way -- tools/probes/gemm_pt.hip keeps the latter);
   *          mvd_gemm rejects it, and mvd_gemm_cfg_supported() tells whether a cfg serves a problem
   *   order: 0 = n-fastest, 1 = m-fastest order of the output tiles over the 8 XCDs.
   * The host mirror times the candidates once per distinct problem shape during the eager warm-up step and passes the
   * winner from then on (mvdfusion_amd/hip.py: autotune). */
  int cfg;
  /* GroupNorm statistics of the OUTPUT, emitted by the producer instead of a separate statistics kernel (MVD_EPI_STORE, n_store == N
   * <= 2560, M % 16 == 0): gn_stats = [M / gn_hw images][gn_groups][2] int64, zeroed by the caller, receives {sum, sum of squares}
   * of every (image, group) as 2^24 fixed point (integer atomics: deterministic); consumed by mvd_groupnorm_from_stats.  gn_hw =
   * rows per image (multiple of 16).  NULL = off. */
  long long* gn_stats;
  int gn_hw, gn_groups;
  /* Row statistics of the OUTPUT for a LayerNorm that is folded into the CONSUMER GEMM (MVD_EPI_STORE, n_store == N): rs_out =
   * [M][rs_ld] pairs of floats {sum, sum of squares} of the stored values of row m over one column slot (a wave tile of the kernel
   * that ran, or a 256-column span of the split-K reduce); the number of slots written per row goes to rs_count[0] (device int).
   * rs_ld >= N / 32 (the narrowest wave tile).  No atomics: the consumer adds the slots in order.  NULL = off.
   * Precision: a slot is a plain fp32 {sum x, sum x^2} over <= 256 columns and the consumer forms var = E[x^2] - mean^2 in double, so the
   * relative error of the variance is ~1e-7 mean^2 / var: validated for rows with |mean| / std <= 3 (tests/test_gpu_ops.py::
   * test_gemm_layernorm_fold, 2e-6); residual streams with outlier channels (|mean| / std in the hundreds) should use mvd_layernorm
   * (two-pass) -- the host mirror's switch is Ctx.ln_fold. */
  float* rs_out;
  int* rs_count;
  int rs_ld;
  /* LayerNorm folded into THIS GEMM (MVD_EPI_QKV / MVD_EPI_GEGLU): A holds the raw rows x (the producer's split planes), the packed
   * weight is W' = W * diag(gamma), and the epilogue forms  rstd_m (x_m . W'_n - mean_m ln_colsum[n]) + bias[n]  with
   * ln_colsum[n] = sum_k W'[n][k] (logical column order, like bias), bias[n] = sum_k beta[k] W[n][k] (+ the layer's own bias),
   * mean / rstd of row m over its ln_dim real columns from the producer's ln_stats = rs_out, ln_count = rs_count, ln_ld = rs_ld,
   * eps = ln_eps.  Exact algebra: LayerNorm is affine per row -- exact in arithmetic only as far as the MFMA operands carry W' and x:
   * use it with the fp16 hi + lo modes (MVD_PREC_X3 / X4 of the f16 library); with one product or bf16 operands the uncancelled
   * mean * sum(W~' - W') term grows with |mean| / std (the host mirror runs mvd_layernorm there).  Runs without split-K.  NULL = off. */
  const float* ln_stats;
  const int* ln_count;
  const float* ln_colsum;
  int ln_ld, ln_dim;
  float ln_eps;
  /* GroupNorm APPLY of the output behind the GEMM (openaimodel.py:201-204: GroupNorm32 -> SiLU -> conv of the next layer; needs gn_stats /
   * gn_hw / gn_groups, MVD_EPI_STORE, out != NULL): after the call gna_out_sp (M, N split planes) holds act(GroupNorm(out) * gamma + beta),
   * gna_flags bit 0 = SiLU, bit 1 = round the normalised value to fp16 first (as mvd_groupnorm_from_stats' `silu`), bit 2 = `out` itself
   * has no other reader: the fused path may leave it unwritten.  How: a split-K GEMM whose (image, group) slab of gn_hw x N / gn_groups
   * values fits 64 KB of LDS runs ONE reduce kernel -- a workgroup per (image, group) sums the slabs, applies the epilogue, keeps the values
   * in LDS, forms mean / rstd and writes the normalised planes: the separate reduce and apply launches and the fp32 round trip between them
   * disappear; otherwise the library launches mvd_groupnorm_from_stats behind the GEMM / the reduce.  NULL = off. */
  void* gna_out_sp;
  const float* gna_gamma;
  const float* gna_beta;
  float gna_eps;
  int gna_flags;
  /* ... over a CONCATENATION (unet.py:550: h = torch.cat([h, hs.pop()], 1) feeding the next ResBlock): with cat_b = the (M, cat_cb) fp32
   * skip tensor the GroupNorm of gna_* runs over [out | cat_b] (N + cat_cb channels; gn_stats / gna_out_sp / gamma / beta refer to the
   * concatenation), and cat_raw_sp (optional) receives the split planes of [out | cat_b] itself (operand of the ResBlock's 1x1 skip
   * convolution).  Split GEMM: the one reduce kernel reads the skip tensor next to the slabs; else mvd_concat_groupnorm runs behind the
   * GEMM.  Shapes: mvd_concat_groupnorm_fits(N, cat_cb, gn_hw, gn_groups).  NULL = off. */
  const float* cat_b;
  int cat_cb;
  void* cat_raw_sp;
  /* Optional DEVICE scalar multiplied into acc_scale (NULL = 1): the backward GEMMs undo the power-of-two scale of their gradient operand
   * (mvd_pow2_scale) with it, without the host ever reading the scale. */
  const float* acc_scale_dev;
  /* In-kernel weight prefetch (hosts: gemm_ws_kernel, cfg loop 7, and the fused reduce + GroupNorm kernel of a split GEMM with gna_out_sp;
   * ignored by every other kernel): pf_n entries of a device table of weights
   * that LATER launches of the step will read (mvd_prefetch_item: ptr, bytes; the other fields unused).  The launch's consumer wavefronts
   * request every 128-byte line of them once at kernel start and drop the data.  Why: a denoising step streams its whole weight set (3.4 GB of
   * packed operands at model_channels 320) through a 256 MB Infinity Cache once per step, so every GEMM meets its weights cold and its short
   * k-loop is a chain of HBM round trips (2 - 7 us per launch of the small and medium GEMMs, profiles/r05_prefetch_probe_whole_weight.log); the
   * role-split convolutions have idle consumer wavefronts and idle HBM bandwidth to spend on the launches behind them.  NULL = off. */
  const struct mvd_prefetch_item_s* pf_items;
  int pf_n;
} mvd_gemm_desc;
#define MVD_GNA_SILU 1
#define MVD_GNA_ROUND_F16 2
#define MVD_GNA_OUT_UNUSED 4

int mvd_gemm(const mvd_gemm_desc* d, mvd_stream_t stream);
/* 1 if kernel configuration `cfg` (see mvd_gemm_desc.cfg) serves the problem `d` describes (tile family vs epilogue, loop variant vs
 * tile, the input-patch kernel vs the convolution's geometry), else 0.  The host autotuner enumerates with it. */
int mvd_gemm_cfg_supported(const mvd_gemm_desc* d, int cfg);

/* fp32 (rows, cols) matrix with leading dim ldx -> split planes (rows, ldp), ldp % 32 == 0; columns [cols, ldp) are 0.
 * Used where a GEMM consumes a tensor that only exists in fp32 (residual stream into the 1x1 skip / up / down convs). */
int mvd_split_planes(const float* x, void* sp, size_t rows, int cols, int ldx, int ldp, mvd_stream_t stream);
/* ... of x * (*scale_dev): the power-of-two gradient scale of mvd_pow2_scale applied on the way into the planes (scale_dev NULL = 1). */
int mvd_split_planes_scaled(const float* x, void* sp, size_t rows, int cols, int ldx, int ldp, const float* scale_dev, mvd_stream_t stream);

/* fp32 matrix-vector products for the M<=16 cases (exact fp32 FMA):
 *   y[m, n] = act_out( sum_k W[n,k] * act_in(x[m,k]) + bias[n] ),  W (N,K) row-major fp32.
 * time_embed / emb_layers / adaLN / cc_projection / kv_len==1 cross-attention vectors:
 *   unet.py:309-314,537-538; openaimodel.py:218-224,264; view_attn_efficient2.py:58-61,64;
 *   viewfusion_zero_depth_rgb.py:110,126-132,276-279,322; attention.py:221 (context length 1). */
int mvd_gemv(const float* W, const float* bias, const float* x, float* y, int M, int N, int K, int ldx, int ldy,
             int act_in, int act_out, mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Normalisation.
 * GroupNorm(32 groups) on channels-last data; `silu` bit 0 = fused SiLU, bit 1 = round the normalised value to fp16 first
 * (the VAE decoder tail, diffusionmodules/model.py:564-570) (openaimodel.py:201-203,225-227 eps 1e-5;
 * attention.py:76,243,274 and mvdfusion/attention.py:92,132 eps 1e-6; unet.py:496-498).
 * ws: B * chunks * groups * 2 doubles with chunks = mvd_groupnorm_chunks(HW); ws_elems = its capacity in doubles (checked). */
int mvd_groupnorm_chunks(int HW);
/* y_sp: the normalised activations in split-planes format (B*HW, C), C % 32 == 0 -- GroupNorm only feeds GEMMs / convs. */
int mvd_groupnorm_nhwc(const float* x, void* y_sp, const float* gamma, const float* beta, int B, int HW, int C,
                       int groups, float eps, int silu, double* ws, size_t ws_elems, mvd_stream_t stream);
/* GroupNorm whose statistics were emitted by the producer of x (mvd_gemm_desc.gn_stats, mvd_concat_channels): one launch
 * (finalise mean / rstd per (image, group) from the fixed-point sums, apply, optional SiLU, write split planes). */
int mvd_groupnorm_from_stats(const float* x, void* y_sp, const float* gamma, const float* beta, const long long* stats, int B, int HW,
                             int C, int groups, float eps, int silu, mvd_stream_t stream);
/* Row softmax: y[r, :] = out_scale * softmax(scale * x[r, :]) of an fp32 (rows, cols) matrix (row stride ldx), written as
 * split planes (rows, cols), cols % 32 == 0, cols <= 4096.  out_scale (a power of two, e.g. 1024) lifts the probabilities of
 * wide rows out of the fp16 subnormal range; the consumer GEMM divides it out through its weight's acc_scale.  Replaces
 * F.softmax in the VAE AttnBlock (external/sd1/ldm/modules/diffusionmodules/model.py:191-193), whose single 512-wide head is
 * run as two GEMMs. */
int mvd_softmax_rows(const float* x, void* y_sp, int rows, int cols, int ldx, float scale, float out_scale,
                     mvd_stream_t stream);

/* LayerNorm over the last dim.  w/b may be NULL (no affine).  w_plus_one: y = norm * (1 + w) + b
 * (adaLN "modulate", view_attn_efficient2.py:15-16,51,53,65-66); attention.py:211-213, mvdfusion/attention.py:35-37. */
int mvd_layernorm(const float* x, void* y_sp, float* y_f32, const float* w, const float* b, int rows, int C, float eps,
                  int w_plus_one, mvd_stream_t stream); /* y_sp: split planes (rows, C), C % 32 == 0, and / or y_f32: fp32 (rows, C) */

/* ------------------------------------------------------------------------------------------------
 * Self-attention over the tokens of one view (CrossAttention with context=None, attention.py:170-193).
 * Operand planes are written by mvd_gemm(MVD_EPI_QKV):
 *   q/k : [B][heads][Lpad][dq]   16-bit hi/lo planes, dq = roundup(dhead, 16), zero padded (32-channel MFMA steps + one 16-channel tail step), q pre-scaled by
 *                                dhead^-0.5 * log2(e) (mvd_gemm_desc.qscale): the softmax is evaluated with exp2
 *   vt  : [B][heads][dv][Lpad]   16-bit hi/lo planes, dv = roundup(dhead, 16)   (V transposed: keys contiguous)
 * out : (B*L, ldo) split planes, head-major channels ('b n (h d)'): feeds the to_out GEMM. */
size_t mvd_attn_qk_plane_elems(int B, int heads, int L, int dhead);
size_t mvd_attn_vt_plane_elems(int B, int heads, int L, int dhead);
int mvd_attn_lpad(int L);
/* L = tokens (rows) per batch item; Lkeys (0 = L) = the leading tokens that take part as KEYS: a sequence padded to a multiple
 * of 4 rows (CLIP: 257 -> 260) keeps its padding rows out of every softmax. */
int mvd_attention(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* vt_hi,
                  const void* vt_lo, void* out_sp, int ldo, int B, int heads, int L, int Lkeys, int dhead, int prec,
                  mvd_stream_t stream);

/* Per-pixel cross attention of one query token against D context tokens (DualAttnetionBlock attn2,
 * mvdfusion/attention.py:56-62; D = n_pts_per_ray).  q (P, C), k/v (P*D, C), out (P, C), C = heads*dhead. */
int mvd_pixel_cross_attn(const float* q, const float* k, const float* v, void* out_sp, int P, int D, int heads,
                         int dhead, mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Layout / data-movement kernels. */
/* UNet input (unet.py:167-187): x (V,5,S,S) NCHW, input_latents (1,5,S,S) NCHW ->
 * out (2V*S*S, cpad) NHWC in split-planes format: rows [0,V) = [x, il[:4]/0.18215, il[4]] , rows [V,2V) = [x, 0]; channels >= 10 zero.
 * With cfg == 0 only the first V rows are produced. */
int mvd_unet_input(const float* x, const float* input_latents, void* out_sp, int V, int S, int cpad, int cfg,
                   mvd_stream_t stream);
/* out[r, 0:Ca] = a[r], out[r, Ca:Ca+Cb] = b[r]  (torch.cat([h, hs.pop()], dim=1), unet.py:550) */
int mvd_concat_channels(const float* a, int Ca, const float* b, int Cb, float* out, void* out_sp, int rows, long long* gn_stats,
                        int gn_hw, int gn_groups, mvd_stream_t stream);
/* concat + the GroupNorm (+ SiLU) that consumes it, in one launch (unet.py:550 -> openaimodel.py:201-204): y_sp (B*hw, Ca+Cb planes) =
 * act(GroupNorm([a | b]) * gamma + beta), raw_sp (optional) = planes of [a | b] itself (the ResBlock's 1x1 skip convolution reads them),
 * out (optional) = the fp32 concatenation, gn_stats (optional) = the {sum, sum of squares} slot of the result as mvd_concat_channels
 * writes it.  `silu`: bit 0 SiLU, bit 1 fp16 rounding first.  A workgroup per (image, group) holds its hw x (C / groups) values in LDS:
 * mvd_concat_groupnorm_fits() tells whether a shape is served (even Ca, Cb and group width, <= 128 KiB per group); otherwise use
 * mvd_concat_channels + mvd_groupnorm_from_stats. */
int mvd_concat_groupnorm_fits(int Ca, int Cb, int hw, int groups);
int mvd_concat_groupnorm(const float* a, int Ca, const float* b, int Cb, float* out, void* raw_sp, void* y_sp, const float* gamma,
                         const float* beta, long long* gn_stats, int B, int hw, int groups, float eps, int silu, mvd_stream_t stream);
/* out_sp optional: split planes for the 1x1 skip conv; gn_stats optional: GroupNorm statistics of `out` (as mvd_gemm_desc.gn_stats;
 * rows % 16 == 0, gn_hw % 16 == 0, (Ca + Cb) % gn_groups == 0, Ca + Cb <= 2560) */
/* area pooling by `factor` of vol (B, S, S, D, C) -> (B, S/f, S/f, D, C)  (unet.py:198-209).  Output: split planes with
 * ldp elements per row (0 = C): the pooled levels only feed GEMMs, and with D == 1 they are written straight into the
 * [attention output | volume features] operand of the merged to_out / cross-attention GEMM (out_sp then points at the
 * volume columns of that wider buffer). */
int mvd_area_pool(const float* vol, void* out_sp, int B, int S, int D, int C, int factor, int ldp, mvd_stream_t stream);
/* out[i] = 0 (memset as a kernel so it is graph-capturable on any stream) */
int mvd_fill_zero(float* p, size_t n, mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Per-step scalars.  A device table `steps` of nsteps x MVD_STEP_STRIDE floats, indexed by a device-resident
 * iteration counter *iter (so a captured graph needs no per-step host input):
 *   [0] t  [1] sqrt(alpha_bar_t)  [2] depth_std = sqrt(1-ab)/sqrt(ab)/10  [3] a_t  [4] a_prev  [5] sigma_t
 *   [6] sqrt(1-a_t)  [7] 1 if noise is added at this step (sampler.py:63-65) */
#define MVD_STEP_STRIDE 8
/* sinusoidal embedding, cos first (diffusionmodules/util.py:152-172; mvdfusion/embedder.py:114-134): out (dim).
 * freqs (dim/2) = exp(-ln(1e4) * i / (dim/2)) is computed once on the host exactly as the reference does. */
int mvd_timestep_embedding(const float* steps, const int* iter, const float* freqs, float* out, int dim,
                           mvd_stream_t stream);
int mvd_advance_iter(int* iter, mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GridAttn: depth-conditioned cross-view aggregation (mvdfusion/view_attn_efficient2.py).
 * Camera record (20 floats): R row-major (9), T (3), focal (2), principal point (2), centre C = -T R^T (3), pad. */
#define MVD_CAM_RECORD 20
#define MVD_TOKEN_DIM 723
#define MVD_TOKEN_LD 736
/* z_embedder: Linear(5->256)+GELU per pixel (:152,434-437).  lat (N,5,S,S) NCHW -> feat (N,S,S,256) NHWC. */
int mvd_zembed(const float* lat, const float* w, const float* b, float* feat, int N, int S, mvd_stream_t stream);
/* G1-G3 (:269-370, :418-432; utils/ray_utils.py:128-212,263-269,367-369; utils/common_utils.py:229-244):
 * depth sample -> unproject -> reproject into every view and the input view -> bilinear gather (border,
 * align_corners) -> Plucker / harmonic embeddings.  Writes the (T, MVD_TOKEN_LD) token matrix, row
 * ((b*S*S + pix)*D + d)*V + v_ref, T = V*S*S*D*V; columns >= 723 are zero.
 * x (V,5,S,S) NCHW noisy latents; depth_noise (nsteps, V, D, S, S) standard normal (host-ordered, trap T2). */
int mvd_gridattn_tokens(const float* x, const float* depth_noise, const float* steps, const int* iter,
                        const float* grid_lin /* (S) = linspace(1-1/S, -1+1/S, S), ray_utils.py:263-267 */,
                        const float* feat, const float* in_feat, const float* cams, const float* in_cam,
                        void* tokens_sp, int V, int q0, int Vq, int S, int D, float depth_scale, float depth_shift,
                        mvd_stream_t stream);
/* Fused G1-G4 (:269-397): tokens are generated in registers and pushed through pre_layer_b, the 3 DiTBlocks over the V views,
 * and the weight_layer softmax pooling in ONE launch; output = the pooled (Vq*S*S*D, 256) rows as split planes (the final
 * Linear 256->768 is a plain mvd_gemm).  1 <= V <= 16: a wavefront owns 16 token rows = 16 / Vp points, Vp = the next power of
 * two >= V; the Vp - V padding slots of a point are masked as attention keys and in the pooling (the reference ships V = 15, 7, 5:
 * configs/mvd_gso.yaml:97, mvd_train.yaml:90,97); Vq*S*S*D*Vp must be a multiple of 64.  V > 16: the unfused kernels above / below.
 *   prec    : MVD_PREC_X4 = all four partial products of the operand split, MVD_PREC_X3 = without lo*lo (every MFMA of the kernel:
 *             the five Linear layers per block and Q K^T)
 *   wstream : the aggregation weights as fp16 (bf16) hi + lo in the kernel's consumption order, mvd_gridattn_fused_slots()
 *             slots of 32 KiB (layout: csrc/gridattn_fused.hip header; packer: mvdfusion_amd/view_attn_efficient2.py)
 *   vecs    : mvd_gridattn_fused_vec_floats() floats -- per DiT block [adaLN modulation of this step 1536 | b_qkv 768 |
 *             b_proj 256 | b_fc1 512 | b_fc2 256], then [b_pre 256 | weight_layer.w 256 | weight_layer.b 1 ... accumulator
 *             scales at +520: pre, then (qkv, proj, fc1, fc2) per block] */
int mvd_gridattn_fused_slots(void);
size_t mvd_gridattn_fused_stream_bytes(void);
size_t mvd_gridattn_fused_vec_floats(void);
int mvd_gridattn_fused(const float* x, const float* depth_noise, const float* steps, const int* iter, const float* grid_lin,
                       const float* feat, const float* in_feat, const float* cams, const float* in_cam, const void* wstream,
                       const float* vecs, void* pooled_sp, int V, int q0, int Vq, int S, int D, float depth_scale,
                       float depth_shift, int prec /* MVD_PREC_X3 | MVD_PREC_X4: partial products per MAC, as mvd_gemm_desc.prec */,
                       mvd_stream_t stream);
/* timm Attention core over the V reference views (:52): qkv (Nseq*V, 3*heads*dhead) -> out (Nseq*V, heads*dhead) */
int mvd_view_mha(const float* qkv, void* out_sp, int Nseq, int V, int heads, int dhead,
                 mvd_stream_t stream); /* output: split planes */
/* weight_layer + softmax over V + weighted sum (:83,396-397): x (Nseq*V, C) -> out (Nseq, C) */
int mvd_view_pool(const float* x, const float* w, const float* b, void* out_sp, int Nseq, int V, int C,
                  mvd_stream_t stream); /* output: split planes */

/* ------------------------------------------------------------------------------------------------
 * CFG combine + DDIM update (unet.py:195; sampler.py:43-66), fused elementwise.
 * eps_nhwc (2V or V, S, S, ldc) UNet head output; x (V,5,S,S) NCHW updated IN PLACE; x0 (V,5,S,S) out.
 * ddim_noise (nsteps, V, 5, S, S).  eps_out (V,5,S,S) NCHW or NULL receives the guided prediction. */
int mvd_cfg_ddim_update(const float* eps_nhwc, int ldc, float* x, float* x0, float* eps_out,
                        const float* ddim_noise, size_t noise_stride /* floats between consecutive steps */,
                        const float* steps, const int* iter, int V, int S, int cfg, float cfg_scale, int do_update,
                        mvd_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * hipGraph capture of a whole denoising step and HIP-event timing on the caller's stream. */
int mvd_graph_begin(mvd_stream_t stream);
int mvd_graph_end(mvd_stream_t stream, void** graph_exec);
int mvd_graph_launch(void* graph_exec, mvd_stream_t stream);
int mvd_graph_destroy(void* graph_exec);
int mvd_event_create(void** ev);
int mvd_event_record(void* ev, mvd_stream_t stream);
int mvd_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int mvd_event_destroy(void* ev);

/* ------------------------------------------------------------------------------------------------
 * Backward of the conv / linear / GroupNorm family (training step, reference train.py:90-95 `loss.backward()`; SURVEY.md
 * section 8(f) rank 4).  The products run on mvd_gemm:  dgrad = mvd_gemm(planes of dY, packed W^T / rotated 3x3 filter);
 * wgrad = mvd_gemm(A = (dY)^T planes, B = MVD_B_PLANES (X)^T or (im2col X)^T planes) -> the parameter's own memory layout.
 * These entries produce the transposed operands, the bias gradient and the GroupNorm(+SiLU) backward.  Deterministic. */
/* x: fp32 (rows, ldx) [src_planes = 0] or split planes (rows, 2*ldx) [1]  ->  out_sp: split planes of x^T, (cols, 2*ldo),
 * ldo % 32 == 0, ldo >= rows rounded up to 32 (columns [rows, ceil32(rows)) are written as zeros). */
int mvd_transpose_planes(const void* x, int src_planes, int rows, int cols, int ldx, void* out_sp, int ldo, mvd_stream_t stream);
/* ... of an fp32 source multiplied by *scale_dev first (see mvd_split_planes_scaled). */
int mvd_transpose_planes_scaled(const float* x, int rows, int cols, int ldx, void* out_sp, int ldo, const float* scale_dev,
                                mvd_stream_t stream);
/* x_sp: channels-last (B,H,W,Cin) activation in split planes, Cin % 32 == 0  ->  out_sp (9*Cin, 2*ldo): row ci*9 + ky*3 + kx,
 * column = output pixel of the 3x3 / stride 1 / pad 1 conv (F.unfold order, transposed). */
int mvd_im2col3x3_t_planes(const void* x_sp, int B, int H, int W, int Cin, void* out_sp, int ldo, mvd_stream_t stream);
/* out2 = {s, 1/s}: the power of two s that brings max|x| of the n floats into [1024, 2048) (1 if the maximum is 0 or not finite) -- the
 * scale a gradient is multiplied with before its fp16 hi + lo split (operand range contract above), on the device, one launch.
 * scratch2: two zero-initialised 32-bit words owned by the caller; the kernel leaves them zero. */
int mvd_pow2_scale(const float* x, size_t n, float* out2, unsigned* scratch2, mvd_stream_t stream);
/* torch.optim.AdamW (no amsgrad) for a list of fp32 tensors in ONE launch (train.py:95 optimizer.step()).  tensors: device array of
 * {float* p; const float* g; float* m; float* v; unsigned long long numel; unsigned first_chunk; unsigned pad;} (48 bytes): parameter,
 * gradient, exp_avg, exp_avg_sq; first_chunk = running sum of ceil(numel / 4096) over the preceding entries, n_chunks = that sum over all.
 * bias_c1 = 1 - beta1^step, bias_c2_sqrt = sqrt(1 - beta2^step); grad_scale multiplies every gradient first (1 = none). */
int mvd_adamw_multi(const void* tensors, int n_tensors, int n_chunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                    float bias_c1, float bias_c2_sqrt, float grad_scale, mvd_stream_t stream);
/* out[c] = sum_r x[r][c] (bias gradient): fp64 partials, fixed order.  ws: mvd_col_sum_workspace_doubles(rows, cols) doubles. */
size_t mvd_col_sum_workspace_doubles(int rows, int cols);
int mvd_col_sum(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, mvd_stream_t stream);
/* mvd_col_sum and mvd_pow2_scale of the same (rows, cols) matrix (ldx == cols: the scale is over exactly the summed elements) in ONE pass:
 * out[c] = column sums, out2 = {s, 1/s}.  scratch1: one zero-initialised word per stream (left zero). */
int mvd_col_sum_pow2(const float* x, int rows, int cols, int ldx, float* out, double* ws, size_t ws_doubles, float* out2, unsigned* scratch1,
                     mvd_stream_t stream);
/* Backward of y = act(GroupNorm(x)) (act = SiLU when silu != 0; torch.nn.GroupNorm semantics, openaimodel.py GroupNorm32):
 * dy = dL/dy  ->  dx (B,HW,C), dgamma (C), dbeta (C).  ws: B*groups*2 + B*C*2 floats. */
int mvd_groupnorm_backward(const float* x, const float* dy, const float* gamma, const float* beta, int B, int HW, int C, int groups,
                           float eps, int silu, float* dx, float* dgamma, float* dbeta, float* ws, size_t ws_floats,
                           mvd_stream_t stream);

/* Backward of y = LayerNorm(x) * w + b (last dimension, w may be null): dx, and dyxhat = dy * xhat so that
 * dw = mvd_col_sum(dyxhat), db = mvd_col_sum(dy).  dyxhat may be null. */
int mvd_layernorm_backward(const float* x, const float* dy, const float* w, int rows, int C, float eps, float* dx, float* dyxhat,
                           mvd_stream_t stream);
/* Activations of the training step (viewfusion_zero_depth_rgb.py:362-397 -> view_attn_efficient2.py:42-67 DiTBlock / Mlp, pre_layer_b):
 * mvd_act_planes: y = act(x), act = MVD_ACT_GELU (exact erf) or MVD_ACT_SILU, of the fp32 matrix x (rows, cols; leading dim ldx) as split
 * planes (sp, ldp % 32 == 0, padded columns zero; NULL = none) and / or fp32 (y, leading dim ldy; NULL = none) in one pass.
 * mvd_act_backward: dx[i] = dy[i] * act'(x[i]) over n elements (dx may alias dy). */
int mvd_act_planes(const float* x, void* sp, float* y, size_t rows, int cols, int ldx, int ldp, int ldy, int act, mvd_stream_t stream);
int mvd_act_backward(const float* dy, const float* x, float* dx, size_t n, int act, mvd_stream_t stream);
/* Backward of GEGLU y = a * gelu(g), [a | g] = h (rows, 2*half) (sd1 attention.py:43-44): dh (rows, 2*half). */
int mvd_geglu_backward(const float* h, const float* dy, int rows, int half, float* dh, mvd_stream_t stream);
/* Backward of the self-attention core softmax(Q K^T / sqrt(d)) V per (batch, head); q, k, v, dout, dq, dk, dv: token-major
 * (B*L, heads*dhead) fp32.  stats: B*heads*L*3 floats of scratch ({max, sum, delta} per query row).  Training path, deterministic.
 * Sequences longer than 16 run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: fp32 operands and accumulation, two kernels -- dQ +
 * row statistics, then dK / dV); the L <= 16 sequences over GridAttn's reference views stay on the VALU kernel. */
int mvd_attention_backward(const float* q, const float* k, const float* v, const float* dout, int B, int heads, int L, int dhead,
                           float* dq, float* dk, float* dv, float* stats, size_t stats_floats, mvd_stream_t stream);
/* Backward of mvd_pixel_cross_attn (D context tokens per pixel): q, dout, dq (P, C); k, v, dk, dv (P*D, C). */
int mvd_pixel_cross_attn_backward(const float* q, const float* k, const float* v, const float* dout, int P, int D, int heads, int dhead,
                                  float* dq, float* dk, float* dv, mvd_stream_t stream);

/* Backward of mvd_gridattn_tokens w.r.t. the feature maps (grid_sample backward, view_attn_efficient2.py:320-341): dtok (T, ldt)
 * fp32 token gradients (columns [0,256) reference-view samples, [256,512) input-view samples) are scattered with the forward's
 * taps into 64-bit fixed-point accumulators dfeat_acc (V,S,S,256) / din_feat_acc (S,S,256) (value * scale; zeroed by the caller). */
int mvd_gridattn_tokens_backward(const float* x, const float* depth_noise, const float* steps, const int* iter, const float* grid_lin,
                                 const float* cams, const float* in_cam, const float* dtok, int ldt, long long* dfeat_acc,
                                 long long* din_feat_acc, float scale, int V, int q0, int Vq, int S, int D, float depth_scale,
                                 float depth_shift, mvd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MVD_HIP_H */
