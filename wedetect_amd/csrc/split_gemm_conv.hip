// split_gemm_conv.hip — fp16x3 implicit-GEMM convolution for PRE-SPLIT activations: the neck / head
// convolution family of the image tower (yolo_world_pafpn.py:40-68, 566-647, 692-715; yolo_world_head.py:194-232).
//
// Round 1-2 ran these layers on the register-staged loader-split kernel (split_gemm_impl.h): fp32 activations are
// re-split by every column tile and every filter tap that re-reads them (2.5 VALU per element), staged through
// VGPRs and ds_write (20-35 % of the LDS cycles were bank conflicts), one barrier and one exposed load latency per
// 12-24 MFMAs: 0.08-0.26 of the fp16x3 roof, 9.2 ms of the 41.6 ms step.  Here the producers write their outputs
// as fp16 hi/lo groups (the WD_SPLIT_C epilogue below, also for residual / dual-format / scatter outputs), and the
// K loop is pure data movement like the ConvNeXt MLP kernels':
//   * every operand byte goes global -> LDS with global_load_lds_dwordx4 (1 KB per instruction, no VGPR staging,
//     no ds_write, no VALU); im2col is done by the DMA's PER-LANE source address: lane (row, 16-byte slot) of an
//     activation instruction points at input pixel (ho*stride - pad + kh, wo*stride - pad + kw), channel chunk of
//     the K stage, or at a zero page when that tap falls outside the image (halo) or the row is past m;
//   * unpadded 64-byte LDS rows, XOR-swizzled on the global side (slot ^ (row/4 & 3)): conflict-free ds_read_b128;
//   * 256 x 128 (or 256 x 64) tile, 8 waves of 64 x 64 (64 x 32), an LDS ring of NBUF stages of k = 16 with the DMA
//     NBUF-1 stages ahead and counted s_waitcnt vmcnt (never 0 in steady state), one barrier per stage.  These
//     launches are small (100-1600 tiles): the ring depth, not occupancy, is what hides the L2 / fabric latency.
// What bounds it (profiles/r03_convpp_ablations_*.txt, timing-only builds): the 32 x 40 x 40, 128 -> 128 layer moves
// 345 MB global -> LDS per launch (im2col re-reads the input nine times, the weight panel once per 256 rows) in ~50 us,
// i.e. the ~6.5 TB/s every LDS-fed kernel of this library ends at; its MFMAs alone take 27 us.
// Same MFMA chain per accumulator and the same K order as the loader-split kernels: BIT-IDENTICAL results
// (tests/test_gpu_split.py).
#include "split_epi_oct.h"

namespace {

constexpr int CV_BM = 256, CV_WROWS = 128, CV_ROWB = 64, CV_STAGE = (CV_BM + CV_WROWS) * CV_ROWB, CV_NI = 3;

__device__ __forceinline__ void cv_dma16(unsigned lds_addr, const unsigned char* src) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

// wave-uniform: let the newest `groups` DMA groups (CV_NI instructions each) stay outstanding
__device__ __forceinline__ void cv_wait_groups(int groups) {
  if (groups <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (groups == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if (groups == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (groups == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if (groups == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}

// ---------------------------------------------------------------------------------------
// TN = 2: 256 x 128 output tile (wave = 64 x 64); TN = 1: 256 x 64 (wave = 64 x 32).  The LDS stage always has 128
// weight rows (rows past n come from the zero page and are never read when TN = 1).
// ksplits > 1: workgroup (tile, ks) walks K stages [ks * per, ...) and writes raw partial sums to ws[ks][m][n];
// splitk_oct_reduce_kernel adds them in split order and applies the epilogue.
// ---------------------------------------------------------------------------------------
// ABL (timing-only, WD_DEBUG_ABLATIONS builds, WRONG results): 1 = no DMA after the prologue, 2 = no fragment reads, 4 = no MFMAs
template <int NBUF, int TN, bool CSPLIT, int ABL = 0>
__global__ void __launch_bounds__(512, NBUF <= 3 ? 4 : 2)
split_conv_pp_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero, int k16,
                     float unscale, int nbn, int ksplits, float* __restrict__ ws) {
  constexpr int TM = 2, BM = CV_BM, BN = 64 * TN, ROWB = CV_ROWB, STAGE = CV_STAGE, NI = CV_NI, DIST = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int group = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ntiles = gridDim.x / ksplits;
  const int ks = tile / ntiles;
  tile -= ks * ntiles;
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk_all = p.k >> 4;
  const int per_split = (nk_all + ksplits - 1) / ksplits;
  const int s_begin = ks * per_split;
  const int nk = (s_begin + per_split < nk_all ? s_begin + per_split : nk_all) - s_begin;

  // DMA sources: instruction j of this wave fills combined rows [(wave*3 + j)*16, +16) of a stage (rows 0..255 =
  // activation rows of the tile, 256..383 = weight rows); lane = (row in the group of 16, 16-byte slot).
  // Activation rows keep the address of their (kh, kw) = (0, 0) input pixel and one validity bit per filter tap;
  // weight rows their row pointer.  Per stage the tap / channel offset is wave-uniform (cin % 16 == 0: a stage never
  // straddles a tap).
  const unsigned char* base[NI];
  unsigned mask[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (wave * NI + j) * 16 + (lane >> 2);
    const int logical = (lane & 3) ^ ((row >> 2) & 3);
    const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
    if (row < BM) {
      const int m = m0 + row;
      const bool ok = m < p.m;
      const int mm = ok ? m : 0;
      const int wo = mm % p.wout;
      const int q = mm / p.wout;
      const int ho = q % p.hout;
      const int b = q / p.hout;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      const long long pix = ((long long)b * p.hin + hi0) * p.win + wi0;
      base[j] = reinterpret_cast<const unsigned char*>(p.a) + pix * p.lda * 4 + memchunk * 16;
      unsigned mk = 0, bit = 1u;                     // no integer division per tap: this runs per lane and per tile
      for (int kh = 0; kh < p.kh; ++kh) {
        const bool hok = (unsigned)(hi0 + kh) < (unsigned)p.hin;
        for (int kw = 0; kw < p.kw; ++kw, bit <<= 1)
          mk |= (hok && (unsigned)(wi0 + kw) < (unsigned)p.win) ? bit : 0u;
      }
      mask[j] = ok ? mk : 0u;
    } else {
      const int n = n0 + row - BM;
      const bool ok = n < p.n;
      base[j] = wsp + (size_t)(ok ? n : 0) * k16 * 4 + memchunk * 16;
      mask[j] = ok ? 0xFFFFFFFFu : 0u;
    }
  }
  // stage cursor of the DMA stream (wave-uniform): filter tap, channel offset inside the tap, weight byte offset
  int i_tap, i_ci, i_kh, i_kw, i_w;
  {
    const int kofs = s_begin * 16;
    i_tap = kofs / p.cin;
    i_ci = kofs - i_tap * p.cin;
    i_kh = i_tap / p.kw;
    i_kw = i_tap - i_kh * p.kw;
    i_w = s_begin * ROWB;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const unsigned char* zp = reinterpret_cast<const unsigned char*>(zero);
  bool prologue = true;
  auto issue = [&](int buf) {
    if ((ABL & 1) && !prologue) return;
    const unsigned lbase = lds0 + buf * STAGE + wave * NI * 1024;
    const int a_off = ((i_kh * p.win + i_kw) * p.lda + i_ci) * 4;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const bool is_a = (wave * NI + j) * 16 < BM;                     // wave-uniform
      const int off = is_a ? a_off : i_w;
      const unsigned bit = is_a ? (mask[j] >> i_tap) & 1u : mask[j] & 1u;
      cv_dma16(__builtin_amdgcn_readfirstlane(lbase + j * 1024), bit ? base[j] + off : zp);
    }
    i_w += ROWB;
    i_ci += 16;
    if (i_ci == p.cin) {
      i_ci = 0;
      ++i_tap;
      if (++i_kw == p.kw) { i_kw = 0; ++i_kh; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  int aoff_h[TM], aoff_l[TM], boff_h[TN], boff_l[TN];
  const int hsel = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = group * 128 + wm * 64 + i * 32 + (lane & 31), f = (row >> 2) & 3;
    aoff_h[i] = row * ROWB + ((hsel ^ f) << 4);
    aoff_l[i] = row * ROWB + (((2 + hsel) ^ f) << 4);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 32 * TN + j * 32 + (lane & 31), f = (row >> 2) & 3;
    boff_h[j] = BM * ROWB + row * ROWB + ((hsel ^ f) << 4);
    boff_l[j] = BM * ROWB + row * ROWB + (((2 + hsel) ^ f) << 4);
  }
  h8 xh[TM], xl[TM], wh[TN], wl[TN];
  auto read = [&](int buf) {
    if ((ABL & 2) && !prologue) return;
    const unsigned char* sp = smem_raw + buf * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      xh[i] = *reinterpret_cast<const h8*>(sp + aoff_h[i]);
      xl[i] = *reinterpret_cast<const h8*>(sp + aoff_l[i]);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      wh[j] = *reinterpret_cast<const h8*>(sp + boff_h[j]);
      wl[j] = *reinterpret_cast<const h8*>(sp + boff_l[j]);
    }
  };
  auto mfma = [&]() {
    if (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
  };

  // prologue: stages 0 .. DIST-1 in flight, all landed before anyone reads
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nk) issue(d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (ABL & 2) read(0);
  prologue = false;
  // Steady state, all eight waves in phase, ONE barrier per K stage:
  //   read the fragments of stage s | issue the DMA of stage s + NBUF - 1 into the buffer every wave finished reading
  //   before the last barrier | 12 MFMAs | wait (counted) until the own share of stage s + 1 has landed | barrier
  // Built on this kernel, bit-identical, measured and dropped: a ping-pong schedule (two wave groups half a stage apart,
  // two barriers per stage: +5-12 %), fragment reads software-pipelined inside the wave (second register set: +-0), and
  // four dedicated loader waves that issue every DMA while the eight compute waves only read and multiply (12 waves, ring
  // of 5-6: +6-9 %) — profiles/r03_convpp_modes.txt, r03_convpp_loader_waves.txt.  The timing-only ablations
  // (r03_convpp_ablations_inphase.txt) price the 32 x 40 x 40, 128 -> 128 layer at 21 us launch + prologue + barriers,
  // 27 us MFMAs, 9 us epilogue and 18 us of operand movement that none of the four schedules hides.
  int bcur = 0, bfill = DIST % NBUF;
  for (int s = 0; s < nk; ++s) {
    read(bcur);
    if (s + DIST < nk) issue(bfill);
    __builtin_amdgcn_sched_barrier(0);
    mfma();
    __builtin_amdgcn_sched_barrier(0);
    {                                          // issued so far: up to stage s+DIST; stage s+1 must be complete
      const int last = s + DIST < nk - 1 ? s + DIST : nk - 1;
      cv_wait_groups(last - (s + 1));
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    bcur = bcur == NBUF - 1 ? 0 : bcur + 1;
    bfill = bfill == NBUF - 1 ? 0 : bfill + 1;
  }
  __syncthreads();

  if (ABL & 8) {                                       // timing only: keep the accumulators live, one store per lane
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (m0 + (t & 255) < p.m) p.c[(size_t)(m0 + (t & 255)) * p.ldc + n0 + (t >> 8)] = sacc;
    return;
  }
  const int mw = m0 + group * 128 + wm * 64, nw = n0 + wn * 32 * TN;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  static_assert(2 * STAGE >= 8 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
  if (ksplits > 1) {
    WdConvGemm pr = p;                                 // raw partial sums, plain rows [m][n] of this split
    pr.bias = nullptr; pr.res = nullptr; pr.c2 = nullptr; pr.range_flag = nullptr;
    pr.c = ws + (size_t)ks * p.m * p.n; pr.ldc = p.n;
    pr.out_mode = WD_OUT_ROWS; pr.c_batch_stride = 0; pr.seg_rows = 0; pr.sigmoid = 0; pr.out_scale = 1.0f; pr.out_bias = 0.0f;
    pr.act = WD_ACT_NONE;
    EpiOctOperands<TM, TN, false> ops;               // no bias, no residual: nothing to load
    ops.load(pr, mw, nw, lane);
    EpiOctWalk<0, TM, TN, WD_ACT_NONE, false, false, false>::run(pr, 1.0f, mw, nw, lane, acc, patch, ops);
    return;
  }
  epi_oct_all<TM, TN, CSPLIT, (NBUF == 4 || TN == 1)>(p, unscale, mw, nw, lane, acc, patch);   // the 128-register forms keep their residual loads in the walk
}

// split-K second pass: one thread per (row, 8 channels); partial sums added in split order (deterministic, and the
// same order as splitk_reduce_kernel: results equal the loader-split kernels' split-K results bit for bit)
template <int ACT, bool SPECIAL, bool CSPLIT>
__global__ void __launch_bounds__(256) splitk_oct_reduce_kernel(const WdConvGemm p, const float* __restrict__ ws, int splits,
                                                                float unscale) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int no = p.n >> 3;
  if (idx >= (long long)p.m * no) return;
  const int m = (int)(idx / no), n = (int)(idx - (long long)m * no) * 8;
  const size_t plane = (size_t)p.m * p.n;
  const float* src = ws + (size_t)m * p.n + n;
  f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
  for (int s = 1; s < splits; ++s) {
    v0 += *reinterpret_cast<const f32x4*>(src + s * plane);
    v1 += *reinterpret_cast<const f32x4*>(src + s * plane + 4);
  }
  epi_oct<ACT, SPECIAL, CSPLIT>(p, unscale, m, n, v0, v1);
}

template <bool CSPLIT>
int launch_oct_reduce(const WdConvGemm& p, const float* ws, int splits, float unscale, hipStream_t st) {
  const long long total = (long long)p.m * (p.n >> 3);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
#define WD_RED(A, S) hipLaunchKernelGGL((splitk_oct_reduce_kernel<A, S, CSPLIT>), grid, block, 0, st, p, ws, splits, unscale)
  if (special) {
    switch (p.act) {
      case WD_ACT_RELU: WD_RED(WD_ACT_RELU, true); break;
      case WD_ACT_SILU: WD_RED(WD_ACT_SILU, true); break;
      case WD_ACT_GELU: WD_RED(WD_ACT_GELU, true); break;
      default: WD_RED(WD_ACT_NONE, true); break;
    }
  } else {
    switch (p.act) {
      case WD_ACT_RELU: WD_RED(WD_ACT_RELU, false); break;
      case WD_ACT_SILU: WD_RED(WD_ACT_SILU, false); break;
      case WD_ACT_GELU: WD_RED(WD_ACT_GELU, false); break;
      default: WD_RED(WD_ACT_NONE, false); break;
    }
  }
#undef WD_RED
  return wd_launch_status();
}

template <int NBUF, int TN, bool CSPLIT, int ABL = 0>
int launch_conv_pp(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, int ksplits, float* ws) {
  constexpr int BN = 64 * TN, LDS = NBUF * CV_STAGE;
  const int nbm = (p.m + CV_BM - 1) / CV_BM, nbn = (p.n + BN - 1) / BN;
  const long long nblk = (long long)nbm * nbn * ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  auto k = split_conv_pp_kernel<NBUF, TN, CSPLIT, ABL>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(512), LDS, st, p, static_cast<const unsigned char*>(wsp), zero, k16, unscale,
                 nbn, ksplits, ws);
  return wd_launch_status();
}

}  // namespace

// What this family covers: pre-split activations (WD_SPLIT_A), cin % 16 == 0, at most 16 filter taps, n % 8 == 0,
// vector-friendly pointers / strides; any geometry (1x1, 3x3 stride 1 / 2, ...), any output form of the epilogue above.
// second pass of a split-K launch of the (row, 8 channels) epilogue family, also for split_gemm_conv3.hip
int wd_launch_oct_reduce(const WdConvGemm& p, const float* ws, int splits, float unscale, bool csplit, hipStream_t st) {
  return csplit ? launch_oct_reduce<true>(p, ws, splits, unscale, st) : launch_oct_reduce<false>(p, ws, splits, unscale, st);
}

bool wd_conv_pp_ok(const WdConvGemm& p, int flags) {
  if (!(flags & WD_SPLIT_A)) return false;
  if (p.cin % 16 || p.k % 16 || p.lda % 8 || p.kh * p.kw > 16 || p.n % 8) return false;
  if (!wd_aligned16(p.a) || !wd_aligned16(p.c)) return false;
  if (p.bias && !wd_aligned16(p.bias)) return false;
  if (p.res && (!wd_aligned16(p.res) || p.ldres % 4)) return false;
  if (p.c2 && (!(flags & WD_SPLIT_C) || !wd_aligned16(p.c2) || p.ldc2 % 4 || p.ldc2 < p.n || p.out_mode != WD_OUT_ROWS))
    return false;                                     // (c_batch_stride: c2 takes c's row mapping, split_epi_oct.h)
  if ((flags & WD_SPLIT_C) ? (p.ldc % 8 != 0) : (p.ldc % 4 != 0)) return false;
  if (p.out_mode == WD_OUT_DECONV2X2 && (p.n % 32)) return false;
  return true;
}

// variant: 0 = production choice; 3 / 4 = ring depth forced (A/B runs)
int wd_launch_conv_pp(const WdConvGemm& p, const void* w, float unscale, int flags, hipStream_t st, int ksplits, float* ws,
                      long long ws_floats, int variant) {
  if (!wd_conv_pp_ok(p, flags)) return WD_ERR_UNSUPPORTED;
  const bool csplit = (flags & WD_SPLIT_C) != 0;
  const bool narrow = (p.n % 128) != 0 && ((p.n + 63) / 64) * 64 < ((p.n + 127) / 128) * 128;
  const long long tiles = (long long)((p.m + CV_BM - 1) / CV_BM) * (narrow ? (p.n + 63) / 64 : (p.n + 127) / 128);
  if (ksplits < 1) ksplits = 1;
  if (ksplits > 1 && (!ws || (long long)ksplits * p.m * p.n > ws_floats)) return WD_ERR_WORKSPACE;
  if (ksplits > (p.k >> 4)) ksplits = p.k >> 4;
  // ring depth: launches that give a CU at most one workgroup hide the load latency with a deeper ring (4 stages,
  // 96 KB); fuller launches run two workgroups per CU on 3 stages (72 KB each) — profiles/r03_convpp_modes.txt
  int nbuf = (tiles * ksplits <= 256) ? 4 : 3;
  if (variant == 3 || variant == 4) nbuf = variant;
  int rc;
#ifdef WD_DEBUG_ABLATIONS
  if (variant >= 100 && variant < 116 && !narrow && csplit) {     // timing-only builds (scripts/conv_pp_abl.py)
    switch (variant - 100) {
      case 1: return launch_conv_pp<4, 2, true, 1>(p, w, unscale, st, 1, ws);
      case 2: return launch_conv_pp<4, 2, true, 2>(p, w, unscale, st, 1, ws);
      case 3: return launch_conv_pp<4, 2, true, 3>(p, w, unscale, st, 1, ws);
      case 4: return launch_conv_pp<4, 2, true, 4>(p, w, unscale, st, 1, ws);
      case 7: return launch_conv_pp<4, 2, true, 7>(p, w, unscale, st, 1, ws);
      case 8: return launch_conv_pp<4, 2, true, 8>(p, w, unscale, st, 1, ws);
      case 11: return launch_conv_pp<4, 2, true, 11>(p, w, unscale, st, 1, ws);
      case 15: return launch_conv_pp<4, 2, true, 15>(p, w, unscale, st, 1, ws);
      default: break;
    }
  }
#endif
#define WD_CPP(NB, TNN, CS) launch_conv_pp<NB, TNN, CS>(p, w, unscale, st, ksplits, ws)
  if (nbuf == 4) {
    if (narrow) rc = csplit ? WD_CPP(4, 1, true) : WD_CPP(4, 1, false);
    else rc = csplit ? WD_CPP(4, 2, true) : WD_CPP(4, 2, false);
  } else {
    if (narrow) rc = csplit ? WD_CPP(3, 1, true) : WD_CPP(3, 1, false);
    else rc = csplit ? WD_CPP(3, 2, true) : WD_CPP(3, 2, false);
  }
#undef WD_CPP
  if (rc != WD_OK || ksplits <= 1) return rc;
  return csplit ? launch_oct_reduce<true>(p, ws, ksplits, unscale, st) : launch_oct_reduce<false>(p, ws, ksplits, unscale, st);
}
