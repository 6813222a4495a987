/*
 * wedetect_hip.h — C ABI of libwedetect_hip.so (MI355X / gfx950 only).
 *
 * The reference (WeChatCV/WeDetect) is pure Python: it has no FFI boundary of its own.
 * Its hot path bottoms out in ATen / cuDNN / cuBLAS ops and two third-party native NMS
 * ops.  This header is the boundary a maintainer binds instead (ctypes stub in
 * INTEGRATION.md); every entry point names the reference site(s) it replaces
 * (paths relative to the reference root).
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*.
 *   - every call is asynchronous on the caller's stream, allocates nothing, keeps no
 *     mutable global state (one-time kernel attribute setup excepted) and returns
 *     WD_OK or a negative WD_ERR_* code; it never throws.
 *   - activations are NHWC fp32: a tensor [B,H,W,C] is a row matrix [B*H*W, C] whose
 *     row stride ("ld", in floats) may exceed C so producers can write straight into a
 *     channel slice of a wider buffer (this is how the reference's torch.cat calls,
 *     yolo_world_pafpn.py:647,715,1127,1131, disappear).
 *   - index outputs are int32 (flat candidate index < 2^31: 33600 anchors x 1203 classes).
 */
#ifndef WEDETECT_HIP_H
#define WEDETECT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WD_OK 0
#define WD_ERR_BAD_ARG (-1)     /* shape / alignment / null-pointer contract violated */
#define WD_ERR_LAUNCH (-2)      /* hipGetLastError() after launch was not hipSuccess  */
#define WD_ERR_WORKSPACE (-3)   /* caller-provided workspace too small                */
#define WD_ERR_UNSUPPORTED (-4)

#define WD_ACT_NONE 0
#define WD_ACT_RELU 1
#define WD_ACT_SILU 2
#define WD_ACT_GELU 3           /* exact erf GELU (nn.GELU default, mm_backbone.py:100) */

#define WD_OUT_ROWS 0           /* C[m, n] at c + m*ldc + n                              */
#define WD_OUT_DECONV2X2 1      /* ConvTranspose2d k2 s2 scatter, see wd_conv_gemm        */

/* ABI version, bumped on any signature change. */
int wd_abi_version(void);
/* Human-readable message for a WD_ERR_* code. */
const char* wd_strerror(int code);

/* ------------------------------------------------------------------------------------
 * wd_conv_gemm — implicit-GEMM convolution / linear layer with fused epilogue.
 *
 *   C[m, n] = epi( sum_k A_im2col[m, k] * W[n, k] + bias[n] )
 *   m = (b, ho, wo) row-major, k = (kh, kw, ci) with ci fastest, W row-major [n][k].
 *   epi(v): v = act(v); v = v*out_scale + out_bias; if sigmoid v = 1/(1+exp(-v));
 *           if res: v += res_alpha * res[m*ldres + n].
 *   seg_rows > 0 (similarity GEMM over all head levels at once): rows are grouped per image
 *   (seg_rows rows each); pos = m % seg_rows selects the head level
 *   lvl = (pos >= seg_end0) + (pos >= seg_end1) and out_scale/out_bias are replaced by
 *   seg_scale[lvl] / seg_bias[lvl] (per-level exp(logit_scale) and bias scalars).
 *   c_batch_stride > 0: output row of m = (b, pos), pos < hout*wout, is b*c_batch_stride + pos
 *   (a level's embeddings go straight into the [B, all-anchors, 768] tensor).
 *   out_mode WD_OUT_DECONV2X2: the GEMM is a 1x1 conv over [B,H,W,cin] with n =
 *   (ty*2+tx)*(N/4) + co; element (m=(b,h,w), n) is stored at output pixel
 *   (b, 2h+ty, 2w+tx), channel co of a [B,2H,2W,*] tensor with row stride ldc.
 *
 * Replaces (all fp32, eval mode; BatchNorm folded into W/bias by the caller):
 *   nn.Conv2d 4x4 s4 stem / 2x2 s2 downsample   wedetect/models/backbones/mm_backbone.py:185-198
 *   nn.Linear pwconv1 + GELU, pwconv2 + gamma + residual   mm_backbone.py:117-124
 *   ConvModule_torch 1x1 / 3x3 (s1|s2) conv + BN + ReLU|SiLU   necks/yolo_world_pafpn.py:40-68
 *   BottleRep  "outputs + alpha * x"                  necks/yolo_world_pafpn.py:602-605
 *   ConvTranspose2d 2x2 s2 + bias                     necks/yolo_world_pafpn.py:195-208
 *   head 3x3 Conv+BN+SiLU and final 1x1 conv (+ folded contrastive BN)
 *                                                     dense_heads/yolo_world_head.py:194-232, 98
 *   BNContrastiveHead einsum 'bchw,bkc->bkhw' * exp(logit_scale) + bias (+ sigmoid of
 *   predict_by_feat)                                  dense_heads/yolo_world_head.py:101-108, 664
 *   Uni einsum 'bchw,kc->bkhw'                        generate_proposal.py:1130-1131, 1185
 *
 * Contract: cin % 4 == 0, lda % 4 == 0, a/w 16-byte aligned, k == kh*kw*cin,
 * m == batch*hout*wout; ldc/ldres/c/res need no alignment (vector stores are used only
 * when they are 16-byte friendly).
 * ---------------------------------------------------------------------------------- */
typedef struct WdConvGemm {
  const float* a;      /* input NHWC [batch, hin, win, cin], pixel stride lda           */
  const float* w;      /* weights [n][k]                                                 */
  const float* bias;   /* [n] or NULL                                                    */
  const float* res;    /* residual [m][ldres] or NULL (same row indexing as C)           */
  float* c;            /* output                                                         */
  int32_t batch, hin, win, cin, lda;
  int32_t kh, kw, stride, pad, hout, wout;
  int32_t m, n, k;
  int32_t ldc, ldres;
  int32_t act;
  int32_t out_mode;
  float res_alpha;
  float out_scale, out_bias;
  int32_t sigmoid;
  int32_t c_batch_stride;      /* rows; 0 = plain row-major output                       */
  int32_t seg_rows, seg_end0, seg_end1;
  float seg_scale[3], seg_bias[3];
  uint32_t* range_flag;        /* fp16x3 kernels only, may be NULL: *range_flag is set to 1 (sticky) when an accumulator
                                * of this launch is inf / NaN, i.e. an operand left the fp16 range; never cleared here */
  float* c2;                   /* fp16x3 + WD_SPLIT_C only, may be NULL: an fp32 copy of the output in plain rows (row
                                * stride ldc2) next to the fp16 hi/lo output in c — for consumers that read fp32 (the
                                * BottleRep residual, yolo_world_pafpn.py:602-605); WD_OUT_ROWS outputs only; with
                                * c_batch_stride (ABI 14) the copy takes the same per-image row mapping as c */
  int32_t ldc2;
  /* fp16x3 range management: power-of-two scales (0 is read as 1) that keep an operand's fp16 (hi, lo) halves inside the
   * fp16 normal range whatever the checkpoint's activation scale.  a_scale: fp32 activations are multiplied by it
   * before the loader splits them (loader-split layers).  c_split_scale: a WD_SPLIT_C output is multiplied by it before
   * it is split (the fp32 copy c2 is not).  The CONSUMER of a scaled operand divides it out by passing
   * w_unscale / scale — exact, powers of two.  wedetect_amd.engine.ImageTower.calibrate() chooses them. */
  float a_scale, c_split_scale;
  /* LayerNorm FOLDED into the GEMM (ABI 13; fp16x3, WD_SPLIT_A | WD_SPLIT_C layers only; both NULL = off).  The operand a is the
   * raw pre-norm tensor d (wd_dwconv7_stats), the weights carry the LayerNorm's gamma (W' = W * gamma), and the epilogue
   * computes  act( rstd_m * (acc - mean_m * ln_u[n]) + bias[n] )  with  ln_stats[m] = (mean_m, rstd_m)  from
   * wd_ln_stats_finalize,  ln_u[n] = sum_k W'[n][k],  bias[n] = (W beta + b)[n]  — which is  W LN(d) + b  term for term
   * (mm_backbone.py:114-117) without the normalised tensor ever being written. */
  const float* ln_stats;       /* [m][2] */
  const float* ln_u;           /* [n], 16-byte aligned */
} WdConvGemm;

int wd_conv_gemm(const WdConvGemm* p, void* stream);

/* Diagnostic: launch a specific tile configuration of the fp32 kernel (see conv_gemm.hip) for on-device A/B runs
 * (scripts/gemm_bench.py).  Not used by the product path; wd_conv_gemm picks the production tile itself. */
int wd_conv_gemm_tuned(const WdConvGemm* p, int32_t cfg, void* stream);

/* ---- fp16x3 variant of wd_conv_gemm --------------------------------------------------------
 * Same contraction, same WdConvGemm description (p->w is ignored), same epilogue, computed with
 * three v_mfma_f32_32x32x16_f16 per product on operands split into fp16 (hi, lo) halves:
 * fp32-equivalent accuracy (2^-22 relative per operand) at the fp16 matrix rate.  Replaces the
 * same reference sites as wd_conv_gemm (nn.Linear / nn.Conv2d in mm_backbone.py:112-125,
 * yolo_world_pafpn.py, yolo_world_head.py); the reference computes them in fp32.
 * Activations must be finite with |x| < 65504 (an fp16 hi half overflows to inf beyond): p->range_flag, when set,
 * receives 1 if a launch produced non-finite accumulators, so the host can re-run with wd_conv_gemm (fp32).
 * Weights are prepared once:
 *   wd_split_weights_bytes(n, k)  -> size of the split buffer (rows padded to 16 k; since ABI 13 also to a whole
 *     group of 8 ROWS, i.e. >= the n * k16 * 4 bytes wd_split_weights writes)
 *   wd_split_weights(w, n, k, scale, out): out <- halves of w * scale, scale a power of two
 *     chosen by the caller so that max|w| * scale <= 2^14; pass w_unscale = 1 / scale below.  Writes exactly n rows.
 *   wd_split_weights_padded(...) (ABI 13): the same, then ZERO rows up to the next multiple of 8 — all of
 *     wd_split_weights_bytes(n, k).  Required for both operands of wd_retrieval_max_split, whose 256 x 256
 *     kernel fetches whole 8-row groups; harmless everywhere else.
 * cfg < 0 picks the production tile; cfg >= 0 selects a tile for A/B runs (split_gemm.hip). */
int64_t wd_split_weights_bytes(int32_t n, int32_t k);
int wd_split_weights(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream);
int wd_split_weights_padded(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream);
int wd_conv_gemm_split(const WdConvGemm* p, const void* w_split, float w_unscale, int32_t flags, int32_t cfg,
                       void* stream);
/* flags: operands that are stored as fp16 (hi, lo) groups instead of fp32 — per row, per 8
 * consecutive elements: [8 x fp16 hi | 8 x fp16 lo] in the 32 bytes the 8 floats would occupy
 * (so buffers, row strides and lda / ldc keep their fp32 meaning).
 *   WD_SPLIT_A: p->a was written by wd_layernorm_rows_split or by a WD_SPLIT_C layer (cin % 8 == 0).  Any geometry:
 *               1x1 layers run the direct-to-LDS GEMM kernels, k x k / strided convolutions the implicit-GEMM
 *               LDS-DMA kernel of split_gemm_conv.hip (cin % 16 == 0; im2col by per-lane DMA addresses, a zero page
 *               for the halo).
 *   WD_SPLIT_C: write p->c in that format for a following WD_SPLIT_A layer (n % 8 == 0, ldc % 8 == 0).  With
 *               WD_SPLIT_A: any output addressing (rows, channel slices, c_batch_stride, the 2x2 deconv scatter), an
 *               fp32 residual and an fp32 copy in p->c2 are allowed.  Without WD_SPLIT_A (fp32 activations split by
 *               the loader): plain rows, no residual.
 * Results are bit-identical to the flags = 0 path (the same halves, produced earlier; same K order). */
#define WD_SPLIT_A 1
#define WD_SPLIT_C 2
/* wd_conv_gemm_split with a caller-owned workspace (16-byte aligned): when a launch has few tiles and a long
 * K loop (batch-1 inference, the 20 x 20 maps) K is split over `splits` workgroups per tile, partial sums go
 * to the workspace ([splits][m][n] fp32) and a second kernel adds them in split order and applies the
 * epilogue (deterministic; the summation order differs from the unsplit kernel's, so the last bits may).
 * splits = 0: decided by the library (1 if the launch already fills the chip or the workspace is too small);
 * splits > 0: forced. */
int wd_conv_gemm_split_ws(const WdConvGemm* p, const void* w_split, float w_unscale, int32_t flags, int32_t cfg,
                          void* workspace, int64_t workspace_bytes, int32_t splits, void* stream);
/* ---- ConvNeXt block MLP as one kernel (narrow stage: c == 128, hidden == 512; other widths: WD_ERR_UNSUPPORTED) ------------
 *   x[m, :] <- x[m, :] + W2 . GELU(W1 . a[m, :] + b1) + b2       (mm_backbone.py:117-124, gamma folded into W2 / b2)
 * a_split: the LayerNorm rows as fp16 hi/lo groups [rows][c] (wd_layernorm_rows_split); w1_split [hidden][c] and w2_split
 * [c][hidden] from wd_split_weights with their unscales; b1 [hidden], b2 [c]; x fp32 [rows][c], in place; rows % 128 == 0.
 * hid_scale: the power-of-two range scale applied to the hidden activations before they are split (the caller passes
 * w2_unscale already divided by it; 1 = none).  The 4c hidden activation never leaves the CU: bit-identical to
 * wd_conv_gemm_split(WD_SPLIT_A | WD_SPLIT_C, GELU) followed by wd_conv_gemm_split(WD_SPLIT_A, residual in place). */
int wd_mlp_fused_split(const void* a_split, int64_t rows, int32_t c, int32_t hidden, const void* w1_split, float w1_unscale,
                       const float* b1, const void* w2_split, float w2_unscale, const float* b2, float* x, float hid_scale,
                       uint32_t* range_flag, void* stream);
/* The same block MLP for the WIDE stages (c == 256 or 512, hidden == 4 c; round 4, split_gemm_mlpw.hip): one workgroup of
 * four waves owns 128 rows, the [128 x c] output tile stays in the waves' accumulators, the hidden dimension is walked in
 * chunks of 128 columns that pass through LDS only.  rows % 128 == 0.  w1_frag / w2_frag: the wd_split_weights buffers of
 * W1 [hidden][c] / W2 [c][hidden] re-ordered FRAGMENT-MAJOR at pack time — [n / 32][k / 16][2 (hi, lo)][64][16 B], lane
 * (k half) * 32 + (row in block) — so that a wave's MFMA operand is one contiguous 1 KB load from global memory (the
 * weights never touch LDS).  Same halves, same K order, same epilogue arithmetic: bit-identical to the two-launch chain.
 * workspace (optional, 16-byte aligned, wd_p8_workspace_bytes() bytes, zero-filled once before its first use — the park
 * workspace of the persistent 256 x 256 kernel serves): with more row blocks than CUs the kernel then runs one workgroup per
 * CU over contiguous ranges of (row block, hidden chunk) units, a row block cut between two CUs is handed over through
 * parked accumulators (same MFMA chain: still bit-identical). */
int wd_mlp_fused_wide(const void* a_split, int64_t rows, int32_t c, int32_t hidden, const void* w1_frag, float w1_unscale,
                      const float* b1, const void* w2_frag, float w2_unscale, const float* b2, float* x, float hid_scale,
                      uint32_t* range_flag, void* workspace, int64_t workspace_bytes, void* stream);
/* wd_mlp_fused_wide_ln (ABI 14) — the same kernel with the block's LayerNorm FOLDED into pwconv1 (mm_backbone.py:113-124:
 * dwconv -> norm -> pwconv1 -> GELU -> pwconv2 -> gamma -> residual, with WdConvGemm.ln_stats / ln_u semantics): d_split = the
 * raw depthwise output written by wd_dwconv7_stats, ln_stats = wd_ln_stats_finalize's per-row (mean, rstd), w1g_frag = the
 * fragment-major split of W1 * gamma, v = W1 beta + b1, u = (W1 gamma) 1:
 *     x <- x + W2 GELU(rstd (W1g d - mean u) + v) + b2
 * — the hidden epilogue of the two-launch fold, element for element (bit-identical to it).  c == 256. */
int wd_mlp_fused_wide_ln(const void* d_split, int64_t rows, int32_t c, int32_t hidden, const void* w1g_frag, float w1_unscale,
                         const float* v, const float* u, const float* ln_stats, const void* w2_frag, float w2_unscale,
                         const float* b2, float* x, float hid_scale, uint32_t* range_flag, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* Name of the tile wd_conv_gemm_split picks with cfg < 0 for an (m, n, k) problem; is_conv != 0:
 * not a 1x1 / stride 1 / pad 0 layer (diagnostic). */
const char* wd_conv_gemm_split_config(int32_t m, int32_t n, int32_t k, int32_t is_conv);
/* Workspace of the persistent 256 x 256 kernel (pre-split plain layers with at least one output tile per CU: the
 * ConvNeXt stage-3 / 4 MLPs at the benchmark batch).  One workgroup per CU walks a contiguous range of
 * (tile, K tile) units; a tile cut between two CUs is started by one, its raw accumulators are parked here, and
 * finished by the other from exactly those values (the same MFMA chain: bit-identical to an unsplit tile).
 * Hand a buffer of this size to wd_conv_gemm_split_ws (splits = 0) and the library picks that kernel where it
 * applies (wd_conv_gemm_split_config(m, n, k, 3) tells).  The first 4 KB must be ZERO before the first launch
 * (flags: the kernel leaves them zero again) and the buffer must not be shared with split-K launches. */
int64_t wd_p8_workspace_bytes(void);

/* ---------------------------------------------------------------------------------------------
 * wd_cv_resize_paste_u8 — OpenCV-style 8-bit resize of one uint8 HWC 3-channel image fused with the letter pad,
 * on the device.  Replaces the host pixels work of the mmdet test pipeline: WeDetectKeepRatioResize
 * (mmcv.imresize -> cv2.resize, INTER_AREA when shrinking / INTER_LINEAR when enlarging;
 * wedetect/datasets/transformers/transforms.py:94-123) followed by WeDetectLetterResize's constant pad
 * (transforms.py:237-258).  The arithmetic restates OpenCV 4.x modules/imgproc/src/resize.cpp (third-party,
 * absent offline: parity unpinned, oracle/cv2_resize.py).  The new_h x new_w result lands at (top, left) of
 * the [dst_h, dst_w, 3] canvas, every other pixel = fill; swap_rb != 0 writes channels reversed (BGR <-> RGB).
 * Tables are device arrays computed by the host (wedetect_amd/pipeline.py):
 *   WD_CVRESIZE_COPY       no resize (new size == source size), no tables
 *   WD_CVRESIZE_AREA_FAST  integer box sums: p0 = iscale_x, p1 = iscale_y, p2 = 1.f / (p0 * p1)
 *   WD_CVRESIZE_AREA       xa [new_w, 2] = (first tap, tap count), xidx / xw = source column and float32 weight
 *                          per tap (computeResizeAreaTab); ya / yidx / yw likewise for rows
 *   WD_CVRESIZE_LINEAR     xa [new_w, 2] = int16-range coefficient pair (11 fractional bits), xidx [new_w] =
 *                          source column, p0 = first output column that reads a single tap; ya [new_h, 2],
 *                          yidx [new_h] = un-clamped source row (clamped to [0, sh) here)
 * ------------------------------------------------------------------------------------------- */
#define WD_CVRESIZE_COPY 0
#define WD_CVRESIZE_AREA_FAST 1
#define WD_CVRESIZE_AREA 2
#define WD_CVRESIZE_LINEAR 3
int wd_cv_resize_paste_u8(const uint8_t* src, int32_t sh, int32_t sw, int32_t mode, const int32_t* xa,
                          const int32_t* xidx, const float* xw, const int32_t* ya, const int32_t* yidx,
                          const float* yw, int32_t p0, int32_t p1, float p2, uint8_t* dst, int32_t dst_h,
                          int32_t dst_w, int32_t new_h, int32_t new_w, int32_t top, int32_t left, int32_t fill,
                          int32_t swap_rb, void* stream);
/* wd_chw_to_hwc_u8 — a packed mmdet batch [batch, 3, h, w] (uint8, or fp32 0..255 when src_is_f32; BGR) ->
 * [batch, h, w, 3] uint8 with the channel order reversed (RGB): DetDataPreprocessor's bgr_to_rgb
 * (data_preprocessor.py:35-36, config/wedetect_base.py:44-48) + the NHWC layout the stem kernel reads. */
int wd_chw_to_hwc_u8(const void* src, int32_t src_is_f32, uint8_t* dst, int32_t batch, int32_t h, int32_t w,
                     void* stream);

/* ---------------------------------------------------------------------------------------------
 * wd_letterbox_u8 — keep-ratio resize + centred pad of one uint8 RGB HWC image on the device,
 * bit-exact with PIL's Image.resize(..., BILINEAR) + paste (generate_proposal.py:17-82 letterbox;
 * Pillow src/libImaging/Resample.c: two separable int32 passes with 22 fractional bits, uint8
 * between the passes).  bounds_* [out, 2] = (first input sample, count), kk_* [out, ksize] = int32
 * weights, both from Pillow's precompute_coeffs / normalize_coeffs_8bpc (wedetect_amd/preprocess.py
 * computes them on the host); tmp: h * new_w * 3 bytes; dst: the [dst_h, dst_w, 3] canvas, filled
 * with (fill_r, fill_g, fill_b) outside the pasted new_h x new_w rectangle at (left, top). */
int wd_letterbox_u8(const uint8_t* src, int32_t h, int32_t w, const int32_t* bounds_h, const int32_t* kk_h,
                    int32_t ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int32_t ksize_v, uint8_t* tmp,
                    uint8_t* dst, int32_t dst_h, int32_t dst_w, int32_t new_w, int32_t new_h, int32_t left, int32_t top,
                    int32_t fill_r, int32_t fill_g, int32_t fill_b, void* stream);

/* wd_retrieval_max in fp16x3 arithmetic (2-4x the fp32 kernel at large banks): both operands as fp16
 * (hi, lo) groups from wd_split_weights_padded (buffers padded with zero rows to whole groups of 8: the kernel
 * fetches 8-row groups) — e_split: the [n_img * rows_per_img, dim] region rows with scale 1, t_split: the [n_cls, dim]
 * bank with a power-of-two scale = 1 / t_unscale.  Same result definition as wd_retrieval_max
 * (retrieval_metric.py:367-377; sigmoid is applied to the maximum of the affine logits — it is monotone); out is
 * zeroed, then max-reduced with atomics (deterministic: max is order independent).  dim % 16 == 0; with
 * dim % 32 == 0 and 16-byte aligned scale / bias the 256 x 256 kernel runs (ABI 13).
 * range_flag (may be NULL; ABI 13): set to 1 (sticky) when an accumulator is inf / NaN, i.e. an operand left
 * the fp16 range — the caller repeats the step with wd_retrieval_max (fp32). */
/* LayerNorm fold (ABI 13), see WdConvGemm.ln_stats: the depthwise 7x7 of a ConvNeXt block writing its output as fp16 hi/lo groups
 * of d * scale plus per-(pixel, 32-channel block) statistics part [c/32][batch*h*w][2]; the finalize pass -> stats [rows][2] =
 * (mean, 1 / sqrt(var + eps)) over all c channels.  c % 32 == 0.  Replaces mm_backbone.py:113-116 (dwconv, permute, norm). */
int wd_dwconv7_stats(const float* x, const float* w7, const float* bias, void* y_split, float* part, int32_t batch, int32_t h,
                     int32_t w, int32_t c, float scale, void* stream);
int wd_ln_stats_finalize(const float* part, float* stats, int64_t rows, int32_t c, float eps, void* stream);

int wd_retrieval_max_split(const void* e_split, const void* t_split, float t_unscale, const float* scale,
                           const float* bias, const int32_t* count, float* out, int32_t n_img, int32_t rows_per_img,
                           int32_t n_cls, int32_t dim, uint32_t* range_flag, void* stream);

/* wd_similarity_split (ABI 14) — replaces, for large text banks, the `x = einsum('bchw,bkc->bkhw', x, w) * logit_scale.exp() + bias`
 * (+ sigmoid) of BNContrastiveHead.forward (wedetect/models/dense_heads/yolo_world_head.py:90-108) and the Uni script's
 * `(embeddings @ prompts.T) * exp(scale) + bias` (generate_proposal.py:1129-1131) — the same contraction as the fp32 similarity
 * launch of wd_conv_gemm, computed on the fp16x3 256 x 256 kernel: e_split = the region embeddings [rows][dim] as fp16 hi/lo groups
 * (what the embedding conv writes with WD_SPLIT_C; the BUFFER must hold a multiple of eight rows), t_split = the text rows
 * [n_cls][dim] from wd_split_weights_padded, unscale = 1 / (text pre-scale * embedding split scale).
 * out[row][cls] (row stride ldo) = (sigmoid)(<e, t> * unscale * seg_scale[l] + seg_bias[l]) with l = the pyramid level of
 * row % seg_rows (boundaries seg_end0 / seg_end1), exactly the `seg` epilogue of wd_conv_gemm.  dim % 32 == 0; any n_cls / ldo.
 * range_flag (optional): set to 1 when an accumulator is inf / NaN (an fp16 half overflowed). */
int wd_similarity_split(const void* e_split, int64_t rows, const void* t_split, float unscale, float* out, int32_t n_cls,
                        int32_t dim, int32_t ldo, int32_t seg_rows, int32_t seg_end0, int32_t seg_end1,
                        const float* seg_scale, const float* seg_bias, int32_t sigmoid, uint32_t* range_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text tower pieces (mm_backbone.py:341-390, HF XLMRobertaModel): the embedding
 * sum before the embedding LayerNorm, and self-attention for short sequences.  Everything else of
 * the encoder is wd_conv_gemm(_split) / wd_layernorm_rows / wd_l2norm_rows.
 *   wd_text_embed: out[t] = word[ids[t]] + pos[pos_ids[t]] + type0                 (dim % 4 == 0)
 *   wd_attention_small: qkv rows [n_seq * seq_len, >= 3 * heads * head_dim] = (Q | K | V), mask
 *     [n_seq, seq_len] (0 = padded key); out rows [n_seq * seq_len, heads * head_dim] =
 *     softmax(Q K^T / sqrt(head_dim)) V per head.  seq_len <= 64, head_dim in {16, 32, 64}. */
int wd_text_embed(const int32_t* ids, const int32_t* pos_ids, const float* word, const float* pos, const float* type0,
                  float* out, int64_t n_tok, int32_t dim, void* stream);
int wd_attention_small(const float* qkv, const int32_t* mask, float* out, int32_t n_seq, int32_t seq_len, int32_t heads,
                       int32_t head_dim, int32_t ld_qkv, int32_t ld_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Text-guided attention bricks (wedetect/models/layers/yolo_bricks.py).  Their
 * convolutions / Linear layers / LayerNorms are wd_conv_gemm(_split) and wd_layernorm_rows; these are
 * the remaining pieces.  All tensors fp32, NHWC rows, leading dimensions in floats (multiples of 4).
 *   wd_max_sigmoid_attn — MaxSigmoidAttnBlock.forward 214-243 after its two convolutions:
 *     embed rows [n_img * hw, >= heads * head_channels] (embed_conv output, or the input itself),
 *     guide [n_img, n_guide, heads * head_channels] (guide_fc output), head_bias [heads], head_scale
 *     [heads] or NULL (= 1).  For pixel r, head m:
 *       a = sigmoid(max_n <embed[r, m, :], guide[img, n, m, :]> / sqrt(head_channels) + head_bias[m]) * head_scale[m]
 *     and x[r, m * out_head_channels : (m + 1) * out_head_channels] (project_conv output) is multiplied by
 *     a IN PLACE.  head_channels in {8, 16, 32, 64, 128}; out_head_channels % 4 == 0.
 *   wd_adaptive_maxpool_nhwc — nn.AdaptiveMaxPool2d((pool, pool)) (yolo_bricks.py:609-612, 618-622) of
 *     x [n_img, h, w, channels]: out[img * out_img_stride + cell * ld_out + c], cell = py * pool + px,
 *     window [floor(i * H / pool), ceil((i + 1) * H / pool)) per axis.  The caller offsets ``out`` per
 *     level to build the concatenated [n_img, levels * pool * pool, E] patch table.
 *   wd_cross_attention_small — 624-646: out[img, r, m, :] = softmax_j(<q[img, r, m, :], k[img, j, m, :]>
 *     / sqrt(head_dim)) . v[img, j, m, :]; q rows [n_img * n_q, ld_q], k / v rows [n_img * n_k, ld_kv],
 *     n_k <= 64, head_dim in {8, 16, 32, 64}. */
int wd_max_sigmoid_attn(const float* embed, int32_t ld_embed, const float* guide, const float* head_bias,
                        const float* head_scale, float* x, int32_t ld_x, int32_t n_img, int32_t hw, int32_t n_guide,
                        int32_t heads, int32_t head_channels, int32_t out_head_channels, void* stream);
int wd_adaptive_maxpool_nhwc(const float* x, int32_t ld_x, float* out, int32_t ld_out, int64_t out_img_stride,
                             int32_t n_img, int32_t h, int32_t w, int32_t channels, int32_t pool, void* stream);
int wd_cross_attention_small(const float* q, int32_t ld_q, const float* k, const float* v, int32_t ld_kv, float* out,
                             int32_t ld_out, int32_t n_img, int32_t n_q, int32_t n_k, int32_t heads, int32_t head_dim,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * wd_recall_match — proposal-recall matching (eval_recall/recall.py:6-100): for
 * every image i and every proposal budget budgets[b], the greedy one-to-one assignment of the
 * reference over the fp32 IoU matrix of gts[gt_off[i]:gt_off[i+1]] x the first min(budget, count)
 * proposals of the image (already sorted by the caller as the reference does); out[b][gt_off[i] + j]
 * = IoU of the j-th assignment (the reference's gt_ious order).  Boxes xyxy fp32; scratch: one
 * slice of scratch_floats_per_block >= max gts x max budget floats per (image, budget); < 65536
 * gts / proposals per image.  legacy: the "+1" width/height convention. */
int64_t wd_recall_scratch_floats(int32_t max_gt, int32_t max_budget);
int wd_recall_match(const float* gts, const int32_t* gt_off, const float* props, const int32_t* prop_off, int32_t n_img,
                    const int32_t* budgets, int32_t n_budget, float* scratch, int64_t scratch_floats_per_block, float* out,
                    int32_t total_gt, int32_t legacy, void* stream);

/* wd_layernorm_rows with the output written as fp16 (hi, lo) groups (see WD_SPLIT_A); c % 8 == 0. */
int wd_layernorm_rows_split(const float* x, void* y, const float* gamma, const float* beta, int64_t rows, int32_t c,
                            int32_t ldx, int32_t ldy, float eps, void* stream);
/* The same with a SPACE-TO-DEPTH output for the 2 x 2 / stride-2 downsample convolutions of ConvNeXt (LayerNorm2d ->
 * Conv2d(k 2, s 2), mm_backbone.py downsample_layers): x is a [batch, h, w, c] map (h, w even), pixel (b, py, px) is
 * written to row (b, py / 2, px / 2) of a [batch * h/2 * w/2, 4 c] matrix at columns ((py & 1) * 2 + (px & 1)) * c —
 * the (kh, kw, cin) order of the convolution's GEMM rows — so that the convolution runs as a plain pre-split GEMM
 * (wd_conv_gemm_split with WD_SPLIT_A, kh = kw = 1, k = 4 c) on the DMA-fed kernels.  Same values as
 * wd_layernorm_rows_split, other addresses; not in place. */
int wd_layernorm_rows_split_s2d(const float* x, void* y, const float* gamma, const float* beta, int32_t batch, int32_t h,
                                int32_t w, int32_t c, float eps, void* stream);

/* Profiling hook.  The next GEMM kernel that wd_conv_gemm / wd_conv_gemm_tuned / wd_conv_gemm_split(_ws) /
 * wd_retrieval_max_split launches from the CALLING THREAD stamps its own begin and end into the two
 * hipEvent_t (created with timing enabled): hipEventElapsedTime(start, stop) is then that kernel's
 * duration as the hardware saw it — hipExtLaunchKernelGGL, so no barrier packets are added to the
 * stream and the step being measured is not disturbed (an hipEventRecord pair around a launch costs
 * ~5 us of idle per launch).  One-shot: cleared by the launch it applies to; (NULL, NULL) clears it.
 * With split K (wd_conv_gemm_split_ws choosing > 1 splits) only the first of the two kernels is stamped. */
int wd_time_next_gemm(void* start_event, void* stop_event);

/* Diagnostic micro-benchmark (not on the product path): the global -> LDS byte rate the LDS-fed GEMM kernels can draw on.
 * `grid` workgroups of 8 waves each stream iters x 8 x 8 KB with global_load_lds_dwordx4 from their own window of
 * window_bytes (walked cyclically: small = cache-resident, large = HBM stream); pattern 0 = 1 KB contiguous per
 * instruction, 1 = 16 rows x 64 B at a row pitch of pitch_bytes (a K = 16 operand stage).  src: grid * window_bytes bytes;
 * sink: grid words.  scripts/lds_dma_probe.py turns it into TB/s. */
int wd_probe_lds_dma(const void* src, int64_t window_bytes, int32_t grid, int32_t iters, int32_t pattern,
                     int32_t pitch_bytes, void* sink, void* stream);

/* Diagnostic: what a wave pays for VALU instructions placed between its own MFMAs.  Every wave of `grid` workgroups of 256
 * threads runs iters x 8 slots of { one v_mfma_f32_32x32x16_f16 + nv VALU instructions of `kind` }; mode 0: both, 1: MFMAs
 * only, 2: VALU only, 3 / 4: as 1 / 0 with ONE dependent MFMA chain.  out[0] = shader-clock ticks, out[1] = 100 MHz ticks of
 * workgroup 0.  Only the (mode, kind, nv) combinations scripts/issue_probe.py uses are instantiated (others: UNSUPPORTED). */
int wd_probe_issue(int32_t mode, int32_t kind, int32_t nv, int32_t grid, int32_t iters, void* out, void* sink, void* stream);

/* sizeof(WdConvGemm) as compiled into the library, so a binding can verify its mirror. */
int wd_sizeof_conv_gemm(void);

/* Name of the tile configuration wd_conv_gemm would pick for (m, n, k), e.g. "128x128".
 * Diagnostic only (bench / tests print it). */
const char* wd_conv_gemm_config(int32_t m, int32_t n, int32_t k);

/* ------------------------------------------------------------------------------------
 * wd_stem_patchify — uint8 RGB NHWC image -> fp32 patch matrix for the 4x4 s4 stem conv.
 *   out[(b, ho, wo), (kh*4+kw)*3 + c] = img[b, 4ho+kh, 4wo+kw, c] / 255
 * Replaces: uint8->float "/255" (generate_proposal.py:1096-1097; DetDataPreprocessor
 * mean 0 / std 255, config/wedetect_base.py:44-48) and the im2col of the stem conv
 * (mm_backbone.py:188-189).  h, w multiples of 4.
 * ---------------------------------------------------------------------------------- */
int wd_stem_patchify(const uint8_t* img, float* out, int32_t batch, int32_t h, int32_t w, void* stream);

/* The whole stem in one kernel: uint8 RGB NHWC image -> x / 255 -> Conv2d(3, c0, 4, stride 4) + bias -> LayerNorm over
 * channels (eps) -> fp32 rows [batch * h/4 * w/4, c0]   (mm_backbone.py:185-190; LayerNorm channels_first :25-47).
 * wgt: [c0, 48] in (kh, kw, cin) column order (the packed stem weight), bias / gamma / beta: [c0]; c0 in {64, 96, 128, 192}
 * (others: WD_ERR_UNSUPPORTED); h % 4 == 0, w % 4 == 0.  The image is read once and the rows are written once; results are
 * BIT-IDENTICAL to wd_stem_patchify -> wd_conv_gemm (fp32) -> wd_layernorm_rows. */
int wd_stem_fused(const uint8_t* img, int32_t batch, int32_t h, int32_t w, const float* wgt, const float* bias,
                  const float* gamma, const float* beta, int32_t c0, float eps, float* out, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_dwconv7 — depthwise 7x7, pad 3, + bias, NHWC.  w7 is [49][c] (tap-major).
 * Replaces nn.Conv2d(dim, dim, 7, padding=3, groups=dim)  mm_backbone.py:96-98, 114.
 * c % 4 == 0; x and y must not alias.
 * ---------------------------------------------------------------------------------- */
int wd_dwconv7(const float* x, const float* w7, const float* bias, float* y,
               int32_t batch, int32_t h, int32_t w, int32_t c, void* stream);
/* The same with the kernel form chosen by the caller (A/B measurements and the identity test; every form gives the same
 * bits): 0 = wd_dwconv7's own choice, 1 = generic strips from global memory, 2 = LDS tile / 1 x 4 strips, 3 = LDS tile /
 * 1 x 8 strips (h % 16 == 0), 4 = the 1 x 4-strip tile staged by LDS-DMA (round 6; the production form wherever c % 32 == 0).
 * 2-4 need c % 32 == 0 (WD_ERR_UNSUPPORTED otherwise). */
int wd_dwconv7_variant(const float* x, const float* w7, const float* bias, float* y,
                       int32_t batch, int32_t h, int32_t w, int32_t c, int32_t variant, void* stream);
/* wd_dwconv7_ln — the same depthwise conv followed by LayerNorm over the channels of each pixel (eps, affine
 * gamma / beta), one kernel: ConvNeXt Block's dwconv -> norm (mm_backbone.py:113-116).  c % 32 == 0.  split != 0:
 * the normalised rows are written as fp16 hi/lo groups (as wd_layernorm_rows_split).  Results are bit-identical to
 * wd_dwconv7 followed by wd_layernorm_rows(_split) in place. */
int wd_dwconv7_ln(const float* x, const float* w7, const float* bias, float* y, const float* gamma, const float* beta,
                  int32_t batch, int32_t h, int32_t w, int32_t c, float eps, int32_t split, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_layernorm_rows — y[r,:] = (x[r,:]-mean)*rsqrt(var+eps)*gamma + beta over c channels,
 * biased variance, two-pass fp32.  In-place allowed (y == x).
 * Replaces F.layer_norm (channels_last, mm_backbone.py:146-149) and the hand-written
 * channels_first LayerNorm (mm_backbone.py:150-155).  c % 4 == 0, c <= 2048.
 * ---------------------------------------------------------------------------------- */
int wd_layernorm_rows(const float* x, float* y, const float* gamma, const float* beta,
                      int64_t rows, int32_t c, int32_t ldx, int32_t ldy, float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_l2norm_rows — y[r,:] = x[r,:] / max(||x[r,:]||_2, 1e-12).
 * Replaces F.normalize(w, dim=-1, p=2) on the text bank
 * (yolo_world_head.py:101; extract_embedding.py:1713).
 * ---------------------------------------------------------------------------------- */
int wd_l2norm_rows(const float* x, float* y, int64_t rows, int32_t c, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_dfl_decode — DFL expectation + distance->box decode for one head level.
 *   dist [batch, hl*wl, 64] (row stride ld): channel = side*16 + bin;
 *   d_side = sum_i softmax(dist[side,:])_i * i;  box = prior -/+ d*stride,
 *   prior = ((x+0.5)*stride, (y+0.5)*stride); written to boxes[b, anchor_off + y*wl + x, 0:4].
 * Replaces yolo_world_head.py:279-288 (DFL), generate_proposal.py:880-905 (priors),
 * distance_point_bbox_coder.py:51-53 + generate_proposal.py:1021-1026 (decode).
 * ---------------------------------------------------------------------------------- */
int wd_dfl_decode(const float* dist, int32_t ld, float* boxes, int32_t batch, int32_t hl, int32_t wl,
                  int32_t stride, int32_t anchor_off, int32_t anchors_total, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_topk_candidates — per image: candidates (anchor, class) with score > thr, sorted by
 * (score desc, flat index asc), truncated to nms_pre.  flat index = anchor*k + class.
 *   scores [batch, n_anchor*k] fp32 (already sigmoid-ed, > 0)
 *   out_idx    [batch, cap] int32 flat indices in sorted order (cap = wd_topk_capacity)
 *   out_score  [batch, cap] fp32
 *   out_count  [batch] int32  (<= nms_pre)
 * Replaces filter_scores_and_topk (generate_proposal.py:85-131; mmdet twin called at
 * yolo_world_head.py:721-722) with the stabilised total order of SURVEY.md §7.
 * Workspace: wd_topk_workspace_bytes(batch, n, nms_pre) bytes, 256-byte aligned.
 * ---------------------------------------------------------------------------------- */
int64_t wd_topk_workspace_bytes(int32_t batch, int64_t n_per_image, int32_t nms_pre);
int32_t wd_topk_capacity(int32_t nms_pre);   /* power of two >= nms_pre */
/* Non-finite guard: if an image's score row holds a NaN or an inf (what an fp16 overflow in an upstream fp16x3 layer
 * turns into), out_count[image] = -1 and wd_nms_gather reports out_count = -1 for it: the host must not trust that
 * image and re-run the step with fp32 kernels (wedetect_amd.engine.ImageTower does). */
int wd_topk_candidates(const float* scores, int32_t batch, int64_t n_per_image, float thr,
                       int32_t nms_pre, int32_t* out_idx, float* out_score, int32_t* out_count,
                       void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_nms_gather — class-aware greedy NMS over sorted candidates + gather of the results.
 *   cand_idx/cand_score/cand_count: output of wd_topk_candidates (row stride cap).
 *   boxes [batch, n_anchor, 4] network-input pixels.  Per image meta (8 floats):
 *     {pad_x, pad_y, inv-unused, scale_x, scale_y, ori_w, ori_h, pre_nms_rescale}
 *   pre_nms_rescale != 0 : mmdet order — box = (box - pad) / scale BEFORE NMS, clamp after
 *                          (yolo_world_head.py:728-746);
 *   pre_nms_rescale == 0 : Uni order — NMS in network pixels, then (box - pad) / scale and
 *                          clamp (generate_proposal.py:1106-1115, 1210).
 *   In every form a box i suppresses a later, not yet suppressed box j that it is compared
 *   with iff   inter / (area_i + area_j - inter) > iou_thr     (fp32, IEEE division,
 *   area = (x2-x1)*(y2-y1), inter = max(0,.)*max(0,.)) — the loop of torchvision's
 *   nms_kernel_impl and of mmcv's nms_cpu (offset 0).  nms_mode selects WHICH boxes are
 *   compared, per image, from its candidate count n (the libraries' own branch conditions):
 *     WD_NMS_VANILLA      same label only, boxes as given (== torchvision _batched_nms_vanilla).
 *     WD_NMS_TORCHVISION  torchvision.ops.batched_nms (torchvision/ops/boxes.py), the call at
 *                         generate_proposal.py:1210: 4*n > mode_param -> vanilla; otherwise
 *                         _batched_nms_coordinate_trick: boxes + label*(max over all 4n
 *                         coordinates + 1) in fp32, then class-AGNOSTIC NMS.  mode_param = 4000
 *                         (CPU tensors; 20000 reproduces a run on a GPU).  torchvision compares
 *                         the fp32 IoU with a C++ double threshold: pass iou_thr rounded DOWN to
 *                         fp32 (largest float <= the Python value) for the same decisions.
 *     WD_NMS_MMCV         mmcv.ops.batched_nms 2.1.0 (mmcv/ops/nms.py) as reached from mmdet's
 *                         _bbox_post_process (yolo_world_head.py:740-744, config/wedetect_base.py:18-25):
 *                         ALWAYS boxes + label*(max+1) in fp32; n < mode_param (split_thr,
 *                         default 10000) -> one class-agnostic NMS, else NMS per class on the
 *                         offset boxes.  float threshold: pass iou_thr rounded to nearest.
 *   (The offset forms quantise the boxes — 0.008 px at label 79, 0.125 px at label 1202 on
 *   1280-px coordinates — and the agnostic ones let boxes of different classes meet when
 *   coordinates fall below -1; both effects are reproduced, not corrected.)
 *   Outputs (row stride max_out): out_boxes [batch,max_out,4], out_scores, out_labels
 *   (int32), out_anchors (int32), out_count [batch]; if embed != NULL also
 *   out_embed [batch, max_out, embed_dim] = embed[b, anchor, :].
 *   Workspace: wd_nms_workspace_bytes(batch) bytes, 4-byte aligned (per-image coordinate
 *   bounds of the offset forms; may be NULL for WD_NMS_VANILLA).
 * Replaces torchvision.ops.batched_nms(...)[:300] (generate_proposal.py:1210),
 * mmdet _bbox_post_process -> mmcv.ops.batched_nms + max_per_img
 * (yolo_world_head.py:740-744), and the index gathers at generate_proposal.py:1208-1217.
 * ---------------------------------------------------------------------------------- */
#define WD_NMS_VANILLA 0
#define WD_NMS_TORCHVISION 1
#define WD_NMS_MMCV 2
int64_t wd_nms_workspace_bytes(int32_t batch);
int wd_nms_gather(const int32_t* cand_idx, const float* cand_score, const int32_t* cand_count,
                  int32_t cand_stride, const float* boxes, int32_t n_anchor, int32_t k,
                  const float* meta, float iou_thr, int32_t max_out, int32_t nms_mode, int32_t mode_param,
                  const float* embed, int32_t embed_dim,
                  float* out_boxes, float* out_scores, int32_t* out_labels, int32_t* out_anchors,
                  int32_t* out_count, float* out_embed, int32_t batch,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * wd_retrieval_max — object-retrieval similarity with fused epilogue:
 *   out[i, c] = max_{r < count[i]} sigmoid( <E[i,r,:], T[c,:]> * exp(scale[i,r]) + bias[i,r] )
 *   E [n_img, rows_per_img, dim], T [n_cls, dim], scale/bias [n_img, rows_per_img],
 *   count [n_img] int32 (0 -> out row = 0).  Logits are never materialised.
 * Replaces eval_retrieval/retrieval_metric.py:369-375 (einsum, sigmoid, max over regions).
 * dim % 4 == 0; rows_per_img <= 320.
 * ---------------------------------------------------------------------------------- */
int wd_retrieval_max(const float* e, const float* t, const float* scale, const float* bias,
                     const int32_t* count, float* out, int32_t n_img, int32_t rows_per_img,
                     int32_t n_cls, int32_t dim, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WEDETECT_HIP_H */
