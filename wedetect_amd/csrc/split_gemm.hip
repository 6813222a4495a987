// split_gemm.hip — the same implicit-GEMM contraction as conv_gemm.hip, computed on the fp16
// matrix pipe with fp32-equivalent accuracy ("fp16x3").
//
// Every fp32 operand value x is carried as two halves, x ~ hi + lo with hi = fp16(x),
// lo = fp16(x - hi) (round to nearest), and a product is evaluated as
//        x * w  ~  hi_x*hi_w + hi_x*lo_w + lo_x*hi_w          (lo*lo < 2^-22 |x w| dropped)
// by three v_mfma_f32_32x32x16_f16 accumulating in fp32.  hi + lo reproduces x to 2^-22
// relative (or 2^-25 absolute below the fp16 normal range), i.e. to the level of fp32's own
// rounding; on the oracle network the embeddings / scores differ from a float64 run by
// 9.5e-6 / 9.2e-7 against 8.5e-6 / 8.0e-7 for plain fp32 (tests/probe_split_precision.py).
// The fp16 pipe's dense peak is 2.5 PFLOP/s, so three passes still offer 5.3x the 157 TFLOP/s
// of v_mfma_f32_16x16x4_f32.  Range: activations must stay below 65504 in magnitude (they are
// LayerNorm / activation outputs here); weights are pre-scaled by a power of two at pack time
// (wd_split_weights) and the inverse scale is applied in the epilogue.
//
// Layout / tiling (per workgroup of WM x WN waves; wave tile 64 x 64 = 2 x 2 MFMA tiles):
//   activations stay fp32 in HBM and are split by the loader (v_cvt_pk_f16_f32, 2.5 VALU per
//   element); weights are split once at pack time into 32-byte groups [hi k..k+7 | lo k..k+7];
//   LDS rows hold, per 16 k: [hi 32 B | lo 32 B], row stride BK*4 + 16 B (80 / 144 B: odd
//   multiples of 16, so the 16 lanes of a ds_read_b128 group hit 16 distinct bank quads);
//   weights are the MFMA "A" operand, activations "B": a lane ends with 4 consecutive output
//   channels of a pixel per accumulator quad, and the epilogue is the one of conv_gemm.hip.
#include <stdlib.h>
#include "split_gemm_impl.h"

namespace {

// ---------------------------------------------------------------------------------------
// weight preparation: fp32 [n][k] -> per row, per 8 k: [8 x fp16 hi | 8 x fp16 lo] of w*scale,
// rows zero-padded to a multiple of 16 k.
// ---------------------------------------------------------------------------------------
// n_out >= n rows are written: rows [n, n_out) are zeros (wd_split_weights_padded: whole groups of eight rows for the DMA-fed
// kernels that fetch 8-row groups)
__global__ void split_weights_kernel(const float* __restrict__ w, int n, int k, int k16, float scale,
                                     unsigned short* __restrict__ out, int n_out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n_out * k16) return;
  const int row = (int)(idx / k16), kk = (int)(idx % k16);
  const float x = (kk < k && row < n) ? w[(size_t)row * k + kk] * scale : 0.0f;
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)(x - (float)hi);
  unsigned short* o = out + (size_t)row * 2 * k16 + (kk >> 3) * 16 + (kk & 7);
  o[0] = __builtin_bit_cast(unsigned short, hi);
  o[8] = __builtin_bit_cast(unsigned short, lo);
}

// split-K second pass: out = epilogue(sum_s ws[s][m][n]) with the partial sums added in split order
// (deterministic), one thread per 4 consecutive channels of a row.
template <int ACT, bool SPECIAL>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const WdConvGemm p, const float* __restrict__ ws, int splits,
                                                            float unscale, int vec_c, int vec_res, int vec_bias) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nq = p.n >> 2;
  if (idx >= (long long)p.m * nq) return;
  const int m = (int)(idx / nq), n = (int)(idx - (long long)m * nq) * 4;
  const size_t plane = (size_t)p.m * p.n;
  f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)m * p.n + n);
  for (int s = 1; s < splits; ++s) v += *reinterpret_cast<const f32x4*>(ws + s * plane + (size_t)m * p.n + n);
  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  epi_quad<ACT, SPECIAL>(p, epi_row<SPECIAL>(p, m), ev, m, n, v);
}

template <bool SPECIAL>
int launch_reduce(const WdConvGemm& p, const float* ws, int splits, float unscale, hipStream_t st) {
  const long long total = (long long)p.m * (p.n >> 2);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c) && (p.out_mode == WD_OUT_ROWS || (p.n % 16 == 0));
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  switch (p.act) {
    case WD_ACT_RELU: hipLaunchKernelGGL((splitk_reduce_kernel<WD_ACT_RELU, SPECIAL>), grid, block, 0, st, p, ws, splits, unscale, vec_c, vec_res, vec_bias); break;
    case WD_ACT_SILU: hipLaunchKernelGGL((splitk_reduce_kernel<WD_ACT_SILU, SPECIAL>), grid, block, 0, st, p, ws, splits, unscale, vec_c, vec_res, vec_bias); break;
    case WD_ACT_GELU: hipLaunchKernelGGL((splitk_reduce_kernel<WD_ACT_GELU, SPECIAL>), grid, block, 0, st, p, ws, splits, unscale, vec_c, vec_res, vec_bias); break;
    default: hipLaunchKernelGGL((splitk_reduce_kernel<WD_ACT_NONE, SPECIAL>), grid, block, 0, st, p, ws, splits, unscale, vec_c, vec_res, vec_bias); break;
  }
  return wd_launch_status();
}

}  // namespace

extern "C" int64_t wd_split_weights_bytes(int32_t n, int32_t k) {
  if (n <= 0 || k <= 0) return 0;
  return (int64_t)((n + 7) & ~7) * ((k + 15) / 16 * 16) * 4;          // rows padded to 8: DMA-fed kernels fetch whole 8-row groups
}

static int split_weights_launch(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream, int n_out) {
  if (!w || !out || n <= 0 || k <= 0 || !(scale > 0.0f) || !wd_aligned16(out)) return WD_ERR_BAD_ARG;
  const int k16 = (k + 15) / 16 * 16;
  const long long total = (long long)n_out * k16;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, n, k, k16, scale, static_cast<unsigned short*>(out), n_out);
  return wd_launch_status();
}

// writes exactly n rows (n * k16 * 4 bytes): the contract of ABI <= 12 — callers that size their own buffers stay correct
extern "C" int wd_split_weights(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream) {
  return split_weights_launch(w, n, k, scale, out, stream, n);
}

// ABI 13: the same, then zero rows up to the next multiple of eight — wd_split_weights_bytes(n, k) bytes in all.  What
// wd_retrieval_max_split wants for both operands (its 256 x 256 kernel fetches whole 8-row groups).
extern "C" int wd_split_weights_padded(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream) {
  return split_weights_launch(w, n, k, scale, out, stream, (n + 7) & ~7);
}

// Production tile choice (profiles/r01_split_gemm_ab.txt): 128 x 128 x 16 (4 waves, four
// workgroups per CU), 128 x 128 x 32 when there are few tiles, 64-wide tiles for narrow n.
static int pick_split_cfg(const WdConvGemm& p) {
  const int m = p.m, n = p.n;
  const long long pad128 = (long long)((n + 127) / 128) * 128, pad64 = (long long)((n + 63) / 64) * 64;
  if (pad64 < pad128) return m >= 65536 ? 52 : 53;                 // narrow n: 64-wide tiles
  const bool plain = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0;
  if (plain && p.k >= 1024) return 55;                             // long K: loads two stages ahead
  const long long tiles = (long long)((m + 127) / 128) * ((n + 127) / 128);
  return tiles < 1024 ? 50 : 51;                                    // few tiles: deeper K stage
}

long long wd_p8_workspace_floats();                                   // split_gemm_p8.hip
bool wd_p8_persist_ok(int m, int n);
// split_gemm_conv.hip: implicit-GEMM LDS-DMA kernel for pre-split activations (any geometry, every output form)
bool wd_conv_pp_ok(const WdConvGemm& p, int flags);
int wd_launch_conv_pp(const WdConvGemm& p, const void* w, float unscale, int flags, hipStream_t st, int ksplits, float* ws,
                      long long ws_floats, int variant);
// split_gemm_conv3.hip: 3 x 3 / stride 1 / pad 1, one stage per (filter row, channel chunk) shared by the row's three taps
bool wd_conv3_ok(const WdConvGemm& p, int flags);
const char* wd_conv3_config_name(int m, int n);
int wd_launch_conv3(const WdConvGemm& p, const void* w, float unscale, int flags, hipStream_t st, int variant, int ksplits, float* ws,
                    long long ws_floats);

extern "C" int64_t wd_p8_workspace_bytes(void) { return 4 * wd_p8_workspace_floats(); }

// Production kernel of a plain layer with pre-split operands (k % 16 == 0).  `park`: the caller offers a
// workspace of wd_p8_workspace_bytes() for the persistent work-unit kernel.  Measured on the ConvNeXt-Base MLP
// shapes at batch 32 (profiles/r02_p8_ab.txt; us per launch, cfg 60 / 63 / 64 / 65):
//   stage 3 pwconv2 51200 x 512 x 2048   405 / 393 / 344 / 333      stage 4 pwconv1 12800 x 4096 x 1024  372 / 375 / 349 / 317
//   stage 3 pwconv1 51200 x 2048 x 512   406 / 443 / 406 / 402      stage 4 pwconv2 12800 x 1024 x 4096  414 / 351 / 294 /  -
//   stage 2 pwconv2 204800 x 256 x 1024  434 / 436 / 442 / 398      stage 2 pwconv1 204800 x 1024 x 256  527 / 567 / 539 / 544
// (after the round-2 epilogue work, profiles/r02_p8_ab.txt second part: stage 2 pwconv1 511 / 481 / 490 for cfg 63 / 64 / 66,
//  stage 1 pwconv1 819200 x 512 x 128: 740 / 799 / 695)
//   65  256 x 256 x 32 persistent (split_gemm_p8.hip): whole column tiles, K >= 256, at least one tile per CU
//   64  the same tile, one workgroup per tile: whole column tiles, K >= 256, at least 128 tiles
//   66  128 x 256 x 16, two workgroups per CU (split_gemm_p4.hip): whole column tiles, shorter K, very long m (stage 1 pwconv1)
//   63  256 x 128 x 16 ping-pong: very long m with narrow n (stage 1 pwconv2)
//   60  128 x 128 x 16 direct-to-LDS: everything else
static int pick_presplit_cfg(int m, int n, int k, bool park) {
  if (k % 32 == 0 && k >= 256 && m % 8 == 0 && n % 256 == 0) {
    const long long tiles = (long long)((m + 255) / 256) * (n / 256);
    // persistent (gang-scheduled, round 4) from K = 1024.  Isolated, the persistent form wins from K = 512 (stage-2 pwconv1, K = 256:
    // 452 -> 473 us, profiles/r04_persist_pmc.txt); inside the step the stage-3 pwconv1 (K = 512, 1 600 tiles of 16 K tiles: 6.25
    // pieces per CU, each with its own prologue) is 1 % of the STEP faster on the tile form, the K >= 1024 layers 0.2 - 0.3 %
    // faster persistent (profiles/r04_persist_mink.txt; $WD_P8_PERSIST_MINK overrides for A/B runs)
    static const int mink = [] { const char* e = getenv("WD_P8_PERSIST_MINK"); return e ? atoi(e) : 1024; }();
    if (park && k >= mink && wd_p8_persist_ok(m, n)) return 65;
    if (tiles >= 128) return 64;
  }
  if (k % 16 == 0 && n % 256 == 0 && m % 16 == 0 && m >= 131072) return 66;   // short K, very long m (stage-1 pwconv1)
  return m >= 131072 ? 63 : 60;
}

// is_conv: 0 = plain, 1 = conv, 2 = plain with pre-split operands, 3 = the same with a park workspace on offer, 5 = 3 x 3 / stride 1 (conv3),
// 4 = pre-split activations through the implicit-GEMM LDS-DMA kernel (k x k / strided convs, scatter / batch-stride / dual outputs)
extern "C" const char* wd_conv_gemm_split_config(int32_t m, int32_t n, int32_t k, int32_t is_conv) {
  if (is_conv == 4) {
    const bool narrow = (n % 128) != 0 && ((n + 63) / 64) * 64 < ((n + 127) / 128) * 128;
    return narrow ? "fp16x3 256x64x16/8w/dma" : "fp16x3 256x128x16/8w/dma";
  }
  if (is_conv == 5) return wd_conv3_config_name(m, n);               // 3 x 3 / stride 1 / pad 1 on pre-split activations: the row-sharing kernels
  if ((is_conv == 2 || is_conv == 3) && k % 16 == 0) {
    switch (pick_presplit_cfg(m, n, k, is_conv == 3)) {
      case 65: return "fp16x3 256x256x32/8w/p8s";
      case 64: return "fp16x3 256x256x32/8w/p8";
      case 66: return "fp16x3 128x256x16/4w/p4";
      case 63: return "fp16x3 256x128x16/8w/pingpong";
      default: return "fp16x3 128x128x16/4w/glds";
    }
  }
  WdConvGemm q{};
  q.m = m; q.n = n; q.k = k; q.kh = q.kw = q.stride = 1;
  if (is_conv) q.kh = q.kw = 3;                                      // any non-1x1 geometry selects the conv loader
  switch (pick_split_cfg(q)) {
    case 50: return "fp16x3 128x128x32/4w";
    case 51: return "fp16x3 128x128x16/4w";
    case 52: return "fp16x3 256x64x16/4w";
    case 55: return "fp16x3 128x128x32/4w/pf2";
    default: return "fp16x3 128x64x16/2w";
  }
}

// K splits for an under-filled launch: few tiles and a long K loop are latency-bound (one workgroup walks
// K serially); S workgroups per tile walk K / S each and a second pass adds the partial sums.
static int pick_ksplits(const WdConvGemm& p, int cfg, int flags, long long ws_floats) {
  if (ws_floats <= 0 || (flags & WD_SPLIT_C) || (p.n & 3) || cfg == 55 || cfg == 63 || cfg >= 64) return 1;
  const int bm = 128, bn = (cfg == 52 || cfg == 53) ? 64 : 128, bk = (cfg == 50) ? 32 : 16;
  const long long tiles = (long long)((p.m + (cfg == 52 ? 255 : bm - 1)) / (cfg == 52 ? 256 : bm)) * ((p.n + bn - 1) / bn);
  const int nk = (p.k + bk - 1) / bk;
  if (tiles >= 128 || nk < 16) return 1;
  int s = (int)(512 / tiles);                               // aim at ~512 workgroups (two per CU)
  if (s > nk / 8) s = nk / 8;                               // at least 8 K stages per workgroup
  if (s > 16) s = 16;
  while (s > 1 && (long long)s * p.m * p.n > ws_floats) --s;
  return s < 2 ? 1 : s;
}

static int conv_gemm_split_impl(const WdConvGemm* pp, const void* w_split, float w_unscale, int32_t flags, int32_t cfg,
                                float* ws, int64_t ws_bytes, int32_t force_splits, hipStream_t st) {
  if (!pp) return WD_ERR_BAD_ARG;
  const WdConvGemm& p = *pp;
  const int rc = check_split_args(p, w_split, w_unscale);
  if (rc != WD_OK) return rc;
  if (flags & ~(WD_SPLIT_A | WD_SPLIT_C)) return WD_ERR_BAD_ARG;
  if (ws && !wd_aligned16(ws)) return WD_ERR_BAD_ARG;
  if (p.c2 && !(flags & WD_SPLIT_C)) return WD_ERR_BAD_ARG;
  const bool production = cfg < 0;
  if (cfg < 0) cfg = pick_split_cfg(p);
  const bool plain = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0;
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
  if (p.ln_stats || p.ln_u) {
    // LayerNorm fold: only the epilogue of the plain pre-split C-split GEMM kernels applies it (epi_lds_tile_csplit)
    if (!p.ln_stats || !p.ln_u || flags != (WD_SPLIT_A | WD_SPLIT_C) || !plain || special || p.res || p.c2 || force_splits > 1 ||
        !wd_aligned16(p.ln_u) || (reinterpret_cast<uintptr_t>(p.ln_stats) & 7u))
      return WD_ERR_BAD_ARG;
    if (cfg == 70 || cfg == 73 || cfg == 74 || (cfg >= 75 && cfg <= 79) || (cfg >= 700 && cfg < 716)) return WD_ERR_UNSUPPORTED;
  }
  if (flags & WD_SPLIT_A) {
    // Pre-split activations.  Plain row-output 1x1 layers keep the GEMM kernels tuned for the ConvNeXt MLPs (below);
    // everything else — k x k / strided convolutions, scatter / batch-stride outputs, residual or dual-format outputs of a
    // WD_SPLIT_C layer — runs the implicit-GEMM LDS-DMA kernel (split_gemm_conv.hip).  cfg 70 / 73 / 74 force it (A/B runs).
    const bool covered = plain && !special && !p.c2 && !((flags & WD_SPLIT_C) && p.res);
    const bool forced = cfg == 70 || cfg == 73 || cfg == 74 || (cfg >= 700 && cfg < 716);
    // cfg 75 forces the row-sharing 3 x 3 kernel; production takes it wherever it applies and K is not split
    // ($WEDETECT_CONV3=0 keeps the tap-per-stage kernel: A/B runs)
    static const bool conv3_on = [] { const char* e = getenv("WEDETECT_CONV3"); return !(e && e[0] == '0'); }();
    // (latency mode — a workspace on offer with the split count left to the library — stays with the tap-per-stage kernel's rule)
    if ((cfg >= 75 && cfg <= 79) || (production && conv3_on && !(force_splits == 0 && ws != nullptr))) {
      if (wd_conv3_ok(p, flags) && wd_conv_pp_ok(p, flags))
        return wd_launch_conv3(p, w_split, w_unscale, flags, st, cfg == 76 ? 2 : cfg == 77 ? 7 : cfg == 78 ? 8 : cfg == 79 ? 9 : 0, force_splits > 1 ? force_splits : 1, ws, ws ? ws_bytes / 4 : 0);
      if (cfg >= 75 && cfg <= 79) return WD_ERR_UNSUPPORTED;
    }
    if (forced || (production && !covered)) {
      if (!wd_conv_pp_ok(p, flags)) return forced || !covered ? WD_ERR_UNSUPPORTED : WD_ERR_BAD_ARG;
      const long long wsf = ws ? ws_bytes / 4 : 0;
      int splits = force_splits > 0 ? force_splits : 1;
      if (force_splits == 0 && wsf > 0 && !forced) {                 // latency mode: under-filled launches split K
        const long long tiles = (long long)((p.m + 255) / 256) * ((p.n + 127) / 128);
        const int nk = p.k / 16;
        if (tiles < 128 && nk >= 32) {
          int s2 = (int)(256 / tiles);
          if (s2 > nk / 16) s2 = nk / 16;
          if (s2 > 8) s2 = 8;
          while (s2 > 1 && (long long)s2 * p.m * p.n > wsf) --s2;
          if (s2 > 1) splits = s2;
        }
      }
      return wd_launch_conv_pp(p, w_split, w_unscale, flags, st, splits, ws, wsf,
                               cfg == 73 ? 3 : cfg == 74 ? 4 : cfg >= 700 ? cfg - 600 : 0);
    }
  }
  if (flags == WD_SPLIT_C) {
    // fp32 activations split by the loader, output written as fp16 hi/lo groups (the neck layers that read the ConvNeXt
    // residual streams and feed pre-split consumers): plain 1x1 layers, plain rows, no residual
    if (!plain || special || p.res || p.c2 || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c))
      return WD_ERR_UNSUPPORTED;
    constexpr int VC = SVAR_XCD | SVAR_PIN | SVAR_LDSEPI | SVAR_CSPLIT;
    const long long tiles = (long long)((p.m + 127) / 128) * ((p.n + 127) / 128);
    return tiles < 1024 ? launch_split<2, 2, 2, 2, 32, VC, 1>(p, w_split, w_unscale, st)
                        : launch_split<2, 2, 2, 2, 16, VC, 1>(p, w_split, w_unscale, st);
  }
  if (flags != 0) {
    // pre-split operands: 128-wide tiles only (every layer on that path has n % 128 == 0 in the
    // shipped towers; other widths still work, with padding).  Plain layers go global -> LDS
    // directly (cfg 60 / 63); the 2x2 downsample conv and odd K keep the register-staged kernels.
    // long-m layers (ConvNeXt stages 1-2): the ping-pong 256 x 128 kernel, else 128 x 128
    if (production && plain && !special && p.k % 16 == 0)
      cfg = pick_presplit_cfg(p.m, p.n, p.k, ws != nullptr && force_splits == 0 && ws_bytes / 4 >= wd_p8_workspace_floats() &&
                                                 p.lda % 8 == 0 && !(flags & ~(WD_SPLIT_A | WD_SPLIT_C)));
    if (cfg != 50 && cfg != 51 && cfg != 55 && cfg != 60 && cfg != 63 && cfg != 64 && cfg != 65 && cfg != 66 && !(cfg >= 640 && cfg < 768)) cfg = 51;
    if ((flags & WD_SPLIT_C) && cfg == 55) cfg = 50;
  }
  const long long ws_floats = ws ? ws_bytes / 4 : 0;
  int splits = force_splits > 0 ? force_splits : pick_ksplits(p, cfg, flags, ws_floats);
  {
    const int bk = (cfg == 50) ? 32 : 16, nk = (p.k + bk - 1) / bk;
    if (splits > nk) splits = nk;                             // never an empty split
  }
  if (splits > 1) {
    if ((flags & WD_SPLIT_C) || (p.n & 3) || cfg == 55 || cfg == 63 || cfg >= 64) return WD_ERR_UNSUPPORTED;
    if ((long long)splits * p.m * p.n > ws_floats) return WD_ERR_WORKSPACE;
  }
  int lrc;
  if (flags != 0) {
    lrc = wd_launch_presplit(p, w_split, w_unscale, cfg, flags, st, splits, ws, ws_floats);
  } else {
    switch (cfg) {
      case 41: lrc = launch_split<2, 2, 2, 4, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st, splits, ws); break;  // 128 x 256 x 16, 8 waves
      case 50: lrc = launch_split<2, 2, 2, 2, 32, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st, splits, ws); break;  // 128 x 128 x 32, 4 waves
      case 51: lrc = launch_split<2, 2, 2, 2, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st, splits, ws); break;  // 128 x 128 x 16, 4 waves
      case 52: lrc = launch_split<2, 2, 4, 1, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st, splits, ws); break;  // 256 x 64 x 16, 4 waves
      case 53: lrc = launch_split<2, 2, 2, 1, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st, splits, ws); break;  // 128 x 64 x 16, 2 waves
      // prefetch distance 2 (two staging register sets; only the BK = 32 plain kernel holds them without spilling)
      case 55: lrc = launch_split<2, 2, 2, 2, 32, SVAR_XCD | SVAR_PF2 | SVAR_LDSEPI>(p, w_split, w_unscale, st); break;
      default: return WD_ERR_UNSUPPORTED;
    }
  }
  if (lrc != WD_OK || splits <= 1) return lrc;
  return special ? launch_reduce<true>(p, ws, splits, w_unscale, st) : launch_reduce<false>(p, ws, splits, w_unscale, st);
}

// cfg < 0: production choice.  Other values select a tile for on-device A/B runs.
extern "C" int wd_conv_gemm_split(const WdConvGemm* pp, const void* w_split, float w_unscale, int32_t flags,
                                  int32_t cfg, void* stream) {
  return conv_gemm_split_impl(pp, w_split, w_unscale, flags, cfg, nullptr, 0, 0, static_cast<hipStream_t>(stream));
}

// The same with a caller-owned workspace (>= 16-byte aligned) that lets under-filled launches split K.
// splits = 0: decided here (1 when the launch already fills the chip); > 0: forced (tests / A/B runs).
extern "C" int wd_conv_gemm_split_ws(const WdConvGemm* pp, const void* w_split, float w_unscale, int32_t flags,
                                     int32_t cfg, void* workspace, int64_t workspace_bytes, int32_t splits, void* stream) {
  if (splits < 0 || workspace_bytes < 0) return WD_ERR_BAD_ARG;
  return conv_gemm_split_impl(pp, w_split, w_unscale, flags, cfg, static_cast<float*>(workspace), workspace_bytes, splits,
                              static_cast<hipStream_t>(stream));
}
