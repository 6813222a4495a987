// split_gemm_pre.hip — fp16x3 GEMM instantiations whose activation operand is already stored as
// fp16 hi/lo groups by its producer (wd_layernorm_rows_split, or a preceding wd_conv_gemm_split
// with WD_SPLIT_C): the K loop then moves both operands as plain 16-byte copies — no conversion
// work is repeated by every column tile that re-reads the same activation rows.
// Layers on this path in the ConvNeXt tower (mm_backbone.py:112-125, 185-198):
//   LayerNorm -> pwconv1 (+GELU, output written split) -> pwconv2 (+residual, fp32 output),
//   LayerNorm -> 2x2 stride-2 downsample conv.
#include <cstdlib>
#include <stdlib.h>
#include "split_gemm_impl.h"

namespace { constexpr int EPI_RES_DEPTH = 2; }   // residual tiles in flight ahead of the one being stored (ping-pong kernel; the 128 x 128 kernel has registers for one)

namespace {
constexpr int VA = SVAR_XCD | SVAR_PIN | SVAR_LDSEPI | SVAR_ASPLIT;
constexpr int VAC = VA | SVAR_CSPLIT;
constexpr int VA_PF2 = SVAR_XCD | SVAR_PF2 | SVAR_LDSEPI | SVAR_ASPLIT;

// ---------------------------------------------------------------------------------------
// Direct-to-LDS variant for plain (1x1) layers with both operands pre-split: every operand byte
// goes global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write, no VALU in
// the K loop).  An LDS-DMA instruction writes 64 lanes x 16 B to CONSECUTIVE LDS addresses, so
// rows are unpadded (BK*4 bytes) and bank conflicts are avoided by an XOR swizzle of the 16-byte
// slot index, applied on the GLOBAL side (each lane picks the chunk that belongs in its slot):
//   physical slot = logical slot ^ f(row),  f(row) = (row / (16 / CH)) & (CH - 1),  CH = slots per row
// (for the 16 lanes of a ds_read_b128 group that gives 16 distinct bank quads).
// Logical slots per 16 k: [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]; memory chunks per 16 k:
// [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15].  128 x 128 tile, 4 waves of 64 x 64, K % BK == 0.
// ---------------------------------------------------------------------------------------
template <int BK, int VAR>
__global__ void __launch_bounds__(256, BK == 16 ? 4 : 3)
split_gemm_glds_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero,
                       int k16, float unscale, int nbn, int vec_c, int vec_res, int vec_bias, int ngrp, int nbm,
                       int ksplits, float* __restrict__ ws) {
  constexpr int TM = 2, TN = 2, WN = 2, BM = 128, BN = 128, KS = BK / 16;
  constexpr int ROWB = BK * 4, CH = ROWB / 16, RPI = 64 / CH, NI = (BM / RPI) / 4, STAGE = (BM + BN) * ROWB;
  constexpr int FDIV = 16 / CH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (ksplits > 1): see split_gemm_kernel
  const int ntiles = gridDim.x / ksplits;
  const int ks = tile / ntiles;
  tile -= ks * ntiles;
  // raster order: column tiles are walked in groups of ngrp (all row panels of a group before the
  // next group), so the weight panels live at any time are ngrp * 128 rows instead of all of n
  const int gsz = ngrp * nbm;
  const int grp = tile / gsz, rem = tile - grp * gsz;
  const int bm = rem / ngrp, bn = grp * ngrp + (rem - bm * ngrp);
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk_all = p.k / BK;
  const int per_split = (nk_all + ksplits - 1) / ksplits;
  const int s_begin = ks * per_split;
  const int nk = (s_begin + per_split < nk_all ? s_begin + per_split : nk_all) - s_begin;

  // per-lane DMA sources: instruction j of this wave fills rows [(wave*NI + j)*RPI, +RPI) of the operand
  const unsigned char* pa[NI];
  const unsigned char* pb[NI];
  int sa[NI], sb[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (wave * NI + j) * RPI + lane / CH;
    const int logical = (lane % CH) ^ ((row / FDIV) & (CH - 1));
    const int memchunk = (logical & ~3) | ((logical & 1) << 1) | ((logical >> 1) & 1);
    const bool aok = m0 + row < p.m, bok = n0 + row < p.n;
    pa[j] = aok ? reinterpret_cast<const unsigned char*>(p.a) + (size_t)(m0 + row) * p.lda * 4 + memchunk * 16
                : reinterpret_cast<const unsigned char*>(zero);
    pb[j] = bok ? wsp + (size_t)(n0 + row) * k16 * 4 + memchunk * 16 : reinterpret_cast<const unsigned char*>(zero);
    sa[j] = aok ? ROWB : 0;
    sb[j] = bok ? ROWB : 0;
    pa[j] += (size_t)sa[j] * s_begin;
    pb[j] += (size_t)sb[j] * s_begin;
  }
  auto issue = [&](int buf) {
    unsigned char* abase = smem_raw + buf * STAGE + wave * NI * RPI * ROWB;
    unsigned char* bbase = abase + BM * ROWB;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa[j],
                                       (__attribute__((address_space(3))) void*)(abase + j * RPI * ROWB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pb[j],
                                       (__attribute__((address_space(3))) void*)(bbase + j * RPI * ROWB), 16, 0, 0);
      pa[j] += sa[j];
      pb[j] += sb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // fragment addresses: row r of the operand, logical slot ks*4 + 2*plane + (lane >> 5)
  int arow[TM], brow[TN], af[TM], bf[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    arow[i] = wm * TM * 32 + i * 32 + (lane & 31);
    af[i] = (arow[i] / FDIV) & (CH - 1);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    brow[j] = wn * TN * 32 + j * 32 + (lane & 31);
    bf[j] = (brow[j] / FDIV) & (CH - 1);
  }
  const int hsel = lane >> 5;

  if (nk > 0) issue(0);                               // an empty K split (never produced by the host side) writes zeros
  __syncthreads();
  for (int s = 0; s < nk; ++s) {
    const unsigned char* as = smem_raw + (s & 1) * STAGE;
    const unsigned char* bs = as + BM * ROWB;
    h8 xh[KS][TM], xl[KS][TM], wh[KS][TN], wl[KS][TN];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        xh[ks][i] = *reinterpret_cast<const h8*>(as + arow[i] * ROWB + (((ks * 4 + hsel) ^ af[i]) << 4));
        xl[ks][i] = *reinterpret_cast<const h8*>(as + arow[i] * ROWB + (((ks * 4 + 2 + hsel) ^ af[i]) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wh[ks][j] = *reinterpret_cast<const h8*>(bs + brow[j] * ROWB + (((ks * 4 + hsel) ^ bf[j]) << 4));
        wl[ks][j] = *reinterpret_cast<const h8*>(bs + brow[j] * ROWB + (((ks * 4 + 2 + hsel) ^ bf[j]) << 4));
      }
    }
    if (s + 1 < nk) issue((s + 1) & 1);               // lands during this stage's MFMAs
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks][j], xh[ks][i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks][j], xl[ks][i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks][j], xh[ks][i], acc[i][j], 0, 0, 0);
    }
    // keep the MFMAs ahead of the wait: hipcc otherwise hoists "s_waitcnt vmcnt(0); s_barrier" above
    // them (they touch no memory) and the DMA latency is exposed instead of hidden under the matrix work
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                   // DMA of stage s+1 landed (vmcnt 0), stage s fully read
  }

  if (ksplits > 1) {
    WdConvGemm pr = p;
    pr.bias = nullptr; pr.res = nullptr; pr.c = ws + (size_t)ks * p.m * p.n; pr.ldc = p.n;
    const EpiVec er{(p.n & 3) == 0, 0, 0, 1.0f};
    split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(pr, er, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, acc,
                                                   reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT);
    return;
  }
  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  static_assert(2 * STAGE >= 4 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
  if (VAR & SVAR_CSPLIT) {
    switch (p.act) {
      case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane, acc, patch); break;
    }
  } else if (epi_res_prefetch_ok(p, ev, nw, TN * 32) && mw < p.m) {
    split_epilogue_res_prefetch<TM, TN, 1>(p, ev, mw, nw, lane, acc, patch);
  } else {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue_lds<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: split_epilogue_lds<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: split_epilogue_lds<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      default: split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane, acc, patch); break;
    }
  }
}

template <int BK, int VAR>
int launch_glds(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, int ksplits = 1, float* ws = nullptr) {
  constexpr int STAGE = 256 * BK * 4, LDS = 2 * STAGE;
  const int nbm = (p.m + 127) / 128, nbn = (p.n + 127) / 128;
  if (ksplits < 1 || (ksplits > 1 && (!ws || (VAR & SVAR_CSPLIT)))) return WD_ERR_BAD_ARG;
  const long long nblk = (long long)nbm * nbn * ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c);
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  // column-tile group of the raster order: 4 (measured: -3 % on the wide pwconv1 layers, whose 4 MB of
  // weight panels otherwise compete with the activation panels for the 4 MB L2); must divide nbn
  int ngrp = 4;
#ifdef WD_DEBUG_ABLATIONS
  static const int env_grp = [] { const char* e = getenv("WD_GLDS_NGROUP"); return e ? atoi(e) : 0; }();   // tuning hook, debug builds only
  if (env_grp > 0) ngrp = env_grp;
#endif
  if (ngrp > nbn || nbn % ngrp) ngrp = nbn;
  auto k = split_gemm_glds_kernel<BK, VAR>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(256), LDS, st, p, static_cast<const unsigned char*>(wsp), zero, k16,
                     unscale, nbn, vec_c, vec_res, vec_bias, ngrp, nbm, ksplits, ws);
  return wd_launch_status();
}


// Tried on top of this kernel and dropped (profiles/r01_split_gemm_ab.txt): BK = 32 stages (fewer
// resident waves: -15..-25 %), and a three-stage LDS ring with the DMA issued through inline asm two
// stages ahead and the next stage's fragments prefetched into a second register set under the
// current stage's MFMAs (bit-identical, but 218 registers = two waves per SIMD: -10..-15 %; at
// three waves per SIMD it spills).  As for the fp32 kernel, resident waves beat per-wave pipelining.
// The ping-pong kernel below (kept: it wins 3-14 % on the long-m stage-1/2 shapes) was also run with
// four LDS stages / DMA three stages ahead (one workgroup per CU: -18 %), with 256 x 256 tiles
// (64 x 128 wave tiles, one workgroup per CU: -5..-10 %) and with s_setprio around the MFMA slot (0 %).
// All structures end between 250 and 300 TFLOP/s on the stage-3/4 shapes.

// ---------------------------------------------------------------------------------------
// Ping-pong variant: 256 x 128 tile, 8 waves in two groups of four (group g owns activation rows
// [128 g, 128 g + 128) of the tile, both share the weight columns).  Waves w and w + 4 sit on the
// same SIMD, one from each group, and the groups run half an iteration apart:
//     slot 2s    : group 0 issues the 12 MFMAs of stage s   | group 1 DMAs its share of stage s+2, reads its stage-s fragments
//     slot 2s+1  : group 0 DMAs its share of stage s+2, reads its stage-(s+1) fragments | group 1 issues the MFMAs of stage s
// with one workgroup barrier between slots, so every SIMD's matrix pipe always has exactly one wave of
// the workgroup feeding it while the partner wave does the memory work (one fragment register set is
// enough: a wave's read slot and MFMA slot never overlap).  Three LDS stages of 24 KB (unpadded,
// XOR-swizzled rows as above); the DMA goes through inline asm so that hipcc does not wait for it
// at the next ds_read; a wave waits for its own DMAs (vmcnt 0) at the end of its MFMA slot, one
// full slot after issuing them, which is before any wave reads that stage.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void dma16(unsigned lds_addr, const unsigned char* src) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

template <int N, int NI>
__device__ __forceinline__ void wait_dma_groups(int groups_in_flight) {
  // wave-uniform: allow the newest `groups_in_flight` DMA groups (NI instructions each) to stay outstanding
  static_assert(NI == 3 || NI == 4, "vmcnt immediates below");
  if (groups_in_flight <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (groups_in_flight == 1 || N == 1) { if (NI == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
  else { if (NI == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
}

// Retrieval epilogue of the ping-pong kernel (SVAR_RETRMAX): the GEMM rows are the region rows of all
// images back to back ([n_img * rows_per_img, dim]); instead of storing logits the tile computes
// sigmoid(logit * exp(scale[row]) + bias[row]) for the valid rows and folds it into
// out[image][class] with a max: segmented over the lanes of a wave (rows of a 32-row tile may belong
// to two images), then one atomicMax per (segment, class) — order-independent, hence deterministic.
constexpr int SVAR_RETRMAX = 4096;
struct RetrArgs {
  const float* scale;
  const float* bias;
  const int* count;
  float* out;
  int rows_per_img, n_cls;
};

template <int TM, int TN>
__device__ __forceinline__ void retrieval_epilogue(const WdConvGemm& p, const RetrArgs& ra, float unscale, int mw, int nw,
                                                   int lane, const f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mw + i * 32 + (lane & 31);
    const bool in = m < p.m;
    const int img = in ? m / ra.rows_per_img : -1;
    const bool valid = in && (m - img * ra.rows_per_img) < ra.count[img];
    const float es = valid ? expf(ra.scale[m]) * unscale : 0.f;
    const float b = valid ? ra.bias[m] : 0.f;
    const int prev_img = __shfl_up(img, 1, 32);
    const bool head = in && ((lane & 31) == 0 || prev_img != img);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = valid ? wd_sigmoid(fmaf(acc[i][j][r], es, b)) : 0.f;     // sigmoid > 0: 0 is the identity of max
        // segmented max towards the first lane of each image's run of rows (runs are contiguous in lane order)
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
          const float o = __shfl_down(v, d, 32);
          const int oi = __shfl_down(img, d, 32);
          if ((lane & 31) + d < 32 && oi == img) v = fmaxf(v, o);
        }
        const int n = nw + j * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (head && n < ra.n_cls)
          atomicMax(reinterpret_cast<unsigned int*>(ra.out + (size_t)img * ra.n_cls + n), __float_as_uint(v));
      }
    }
  }
}

__global__ void zero_f32_kernel(float* __restrict__ x, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 0.f;
}

template <int VAR, int NBUF, int TN>
__global__ void __launch_bounds__(512, (NBUF == 3 && TN == 2) ? 4 : 2)
split_gemm_pingpong_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero,
                           int k16, float unscale, int nbn, int vec_c, int vec_res, int vec_bias, const RetrArgs ra) {
  constexpr int TM = 2, BM = 256, BN = 64 * TN, BK = 16, ROWB = 64, DIST = NBUF - 1;   // DMA runs DIST stages ahead
  constexpr int STAGE = (BM + BN) * ROWB, NI = (BM + BN) / 16 / 8;   // one-KB DMA instructions per stage and wave: 3 or 4
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int group = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int nk = p.k / BK;

  // DMA sources: instruction j of this wave fills combined rows [(wave*3 + j)*16, +16) of a stage
  // (rows 0..255 = activations, 256..383 = weights); lane = (row in the group of 16, 16-byte slot)
  const unsigned char* ps[NI];
  int st[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (wave * NI + j) * 16 + (lane >> 2);
    const int logical = (lane & 3) ^ ((row >> 2) & 3);
    const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
    const bool is_a = row < BM;
    const bool ok = is_a ? (m0 + row < p.m) : (n0 + row - BM < p.n);
    const unsigned char* src = is_a ? reinterpret_cast<const unsigned char*>(p.a) + (size_t)(m0 + row) * p.lda * 4
                                    : wsp + (size_t)(n0 + row - BM) * k16 * 4;
    ps[j] = ok ? src + memchunk * 16 : reinterpret_cast<const unsigned char*>(zero);
    st[j] = ok ? ROWB : 0;
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  auto issue = [&](int buf) {
    const unsigned base = lds0 + buf * STAGE + wave * NI * 1024;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      dma16(__builtin_amdgcn_readfirstlane(base + j * 1024), ps[j]);
      ps[j] += st[j];
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
    __builtin_amdgcn_s_setprio(1);            // the wave feeding the matrix pipe goes first; its partner only moves data
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
    __builtin_amdgcn_s_setprio(0);
  };

  // prologue: stages 0 .. DIST-1 in flight, all landed before anyone reads
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nk) issue(d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // ring: stage t lives in buffer t % NBUF
  int bcur = 0, bnext = 1, bfill = DIST % NBUF;
  auto advance = [&]() {
    bcur = bcur == NBUF - 1 ? 0 : bcur + 1;
    bnext = bnext == NBUF - 1 ? 0 : bnext + 1;
    bfill = bfill == NBUF - 1 ? 0 : bfill + 1;
  };
  // slot boundary: nothing (not even an MFMA, which touches no memory) may be scheduled across it
#define WD_SLOT_BARRIER()                 \
  __builtin_amdgcn_sched_barrier(0);      \
  __syncthreads();                        \
  __builtin_amdgcn_sched_barrier(0)
  // Before the barrier that ends slot 2s every wave's share of stage s+1 must have landed (group 0
  // reads it in slot 2s+1); DMA groups issued after it may stay in flight.
  if (group == 0) {
    read(0);                                   // fragments of stage 0 for the first MFMA slot
    for (int s = 0; s < nk; ++s) {
      mfma();                                  // slot 2s
      __builtin_amdgcn_sched_barrier(0);
      {                                        // issued so far: up to stage s+DIST-1
        const int last = s + DIST - 1 < nk - 1 ? s + DIST - 1 : nk - 1;
        wait_dma_groups<DIST - 2, NI>(last - (s + 1));
      }
      WD_SLOT_BARRIER();
      if (s + DIST < nk) issue(bfill);         // slot 2s+1
      if (s + 1 < nk) read(bnext);
      WD_SLOT_BARRIER();
      advance();
    }
  } else {
    for (int s = 0; s < nk; ++s) {
      if (s + DIST < nk) issue(bfill);         // slot 2s
      read(bcur);
      __builtin_amdgcn_sched_barrier(0);
      {                                        // issued so far: up to stage s+DIST
        const int last = s + DIST < nk - 1 ? s + DIST : nk - 1;
        wait_dma_groups<DIST - 1, NI>(last - (s + 1));
      }
      WD_SLOT_BARRIER();
      mfma();                                  // slot 2s+1
      WD_SLOT_BARRIER();
      advance();
    }
  }
#undef WD_SLOT_BARRIER
  __syncthreads();

  if (VAR & SVAR_RETRMAX) {
    retrieval_epilogue<TM, TN>(p, ra, unscale, m0 + group * 128 + wm * 64, n0 + wn * 32 * TN, lane, acc);
    return;
  }
  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  const int mw = m0 + group * 128 + wm * 64, nw = n0 + wn * 32 * TN;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  static_assert(NBUF * STAGE >= 8 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
  if (VAR & SVAR_CSPLIT) {
    switch (p.act) {
      case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane, acc, patch); break;
    }
  } else if (epi_res_prefetch_ok(p, ev, nw, TN * 32) && mw < p.m) {
    split_epilogue_res_prefetch<TM, TN, EPI_RES_DEPTH>(p, ev, mw, nw, lane, acc, patch);
  } else {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue_lds<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: split_epilogue_lds<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: split_epilogue_lds<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      default: split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane, acc, patch); break;
    }
  }
}

template <int VAR, int NBUF, int TN>
int launch_pingpong(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, const RetrArgs& ra = RetrArgs{}) {
  constexpr int BN = 64 * TN, LDS = NBUF * (256 + BN) * 64;
  const int nbm = (p.m + 255) / 256, nbn = (p.n + BN - 1) / BN;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c);
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  auto k = split_gemm_pingpong_kernel<VAR, NBUF, TN>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(512), LDS, st, p, static_cast<const unsigned char*>(wsp), zero, k16,
                     unscale, nbn, vec_c, vec_res, vec_bias, ra);
  return wd_launch_status();
}

// plain layer, rows output, no per-level affine: what the direct-to-LDS kernel's epilogue covers
bool glds_ok(const WdConvGemm& p, int bk) {
  const bool plain = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0;
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
  return plain && !special && p.k % bk == 0;
}
}  // namespace

int wd_launch_p8(const WdConvGemm& p, const void* w, float unscale, bool csplit, hipStream_t st, int abl, bool persist, float* ws,
                 long long ws_floats);   // split_gemm_p8.hip

int wd_launch_p4(const WdConvGemm& p, const void* w, float unscale, bool csplit, hipStream_t st);   // split_gemm_p4.hip

int wd_launch_presplit(const WdConvGemm& p, const void* w, float unscale, int cfg, int flags, hipStream_t st, int ksplits,
                       float* ws, long long ws_floats) {
  const bool csplit = (flags & WD_SPLIT_C) != 0;
  if (!(flags & WD_SPLIT_A)) return WD_ERR_UNSUPPORTED;            // C-only split: not needed by any layer yet
  if (p.k % 8 || p.cin % 8 || p.lda % 8) return WD_ERR_BAD_ARG;
  if (cfg == 66) {                       // 128 x 256 tiles, four waves, two workgroups per CU (split_gemm_p4.hip)
    if (!glds_ok(p, 16)) return WD_ERR_UNSUPPORTED;
    return wd_launch_p4(p, w, unscale, csplit, st);
  }
  if (cfg == 64 || cfg == 65 || (cfg >= 640 && cfg < 768)) {   // 256 x 256 tiles, four phases per K tile of 32, counted DMA waits (split_gemm_p8.hip)
    if (!glds_ok(p, 32)) return WD_ERR_UNSUPPORTED;
    // 65: persistent work-unit form (needs the workspace); 640 + ablation mask: debug builds only
    return wd_launch_p8(p, w, unscale, csplit, st, cfg >= 640 ? cfg - 640 : 0, cfg == 65, ws, ws_floats);
  }
  if (cfg == 63) {                       // direct-to-LDS, ping-pong wave groups, 256 x 128 tiles
    if (!glds_ok(p, 16)) return WD_ERR_UNSUPPORTED;
    if (csplit) {
      if (p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c)) return WD_ERR_BAD_ARG;
      return launch_pingpong<SVAR_CSPLIT, 3, 2>(p, w, unscale, st);
    }
    return launch_pingpong<0, 3, 2>(p, w, unscale, st);
  }
  if (cfg == 60) {                       // direct-to-LDS kernel
    if (!glds_ok(p, 16)) return WD_ERR_UNSUPPORTED;
    if (csplit) {
      if (p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c)) return WD_ERR_BAD_ARG;
      return launch_glds<16, SVAR_CSPLIT>(p, w, unscale, st);
    }
    return launch_glds<16, 0>(p, w, unscale, st, ksplits, ws);
  }
  if (csplit) {
    const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                         p.out_scale != 1.0f || p.out_bias != 0.0f;
    if (special || p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c))
      return WD_ERR_BAD_ARG;
    switch (cfg) {
      case 50: return launch_split<2, 2, 2, 2, 32, VAC, 1>(p, w, unscale, st);
      case 51: return launch_split<2, 2, 2, 2, 16, VAC, 1>(p, w, unscale, st);
      default: return WD_ERR_UNSUPPORTED;
    }
  }
  switch (cfg) {
    case 50: return launch_split<2, 2, 2, 2, 32, VA, 0>(p, w, unscale, st, ksplits, ws);
    case 51: return launch_split<2, 2, 2, 2, 16, VA, 0>(p, w, unscale, st, ksplits, ws);
    case 55: return launch_split<2, 2, 2, 2, 32, VA_PF2, 1>(p, w, unscale, st);
    default: return WD_ERR_UNSUPPORTED;
  }
}

// wd_retrieval_max with fp16x3 arithmetic: e_split = region embeddings [n_img * rows_per_img, dim] and
// t_split = text bank [n_cls, dim], both as fp16 hi/lo groups (wd_split_weights; the bank pre-scaled
// by 1 / t_unscale).  out is zeroed here, then filled by atomic max.  Round 5: on the 256 x 256 kernel with the operand
// roles swapped (split_gemm_p8.hip: P8Retr) wherever it applies — dim % 32 == 0, aligned scale / bias — in bank chunks of
// 2^20 classes (32-bit DMA offsets); the 256 x 128 ping-pong form below remains for the other shapes and as the A/B
// reference ($WEDETECT_RETR_P8=0).
int wd_launch_p8_retrieval(const void* t_split, int n_cls, const void* e_split, int n_rows, int dim, float unscale,
                           const float* scale, const float* bias, const int* count, int rows_per_img, float* out, int ldo,
                           unsigned* range_flag, hipStream_t st);   // split_gemm_p8.hip

extern "C" int wd_retrieval_max_split(const void* e_split, const void* t_split, float t_unscale, const float* scale,
                                      const float* bias, const int32_t* count, float* out, int32_t n_img,
                                      int32_t rows_per_img, int32_t n_cls, int32_t dim, uint32_t* range_flag, void* stream) {
  if (!e_split || !t_split || !scale || !bias || !count || !out) return WD_ERR_BAD_ARG;
  if (n_img <= 0 || n_cls <= 0 || rows_per_img <= 0 || dim <= 0 || dim % 16 || !(t_unscale > 0.f)) return WD_ERR_BAD_ARG;
  if ((long long)n_img * rows_per_img > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(e_split) || !wd_aligned16(t_split)) return WD_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long total = (long long)n_img * n_cls;
  hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, total);
  const long long n_rows = (long long)n_img * rows_per_img;
  const char* env_p8 = getenv("WEDETECT_RETR_P8");                // read per call: the tests toggle it
  const bool use_p8 = !(env_p8 && env_p8[0] == '0');
  if (use_p8 && dim % 32 == 0 && wd_aligned16(scale) && wd_aligned16(bias) &&
      (unsigned long long)((n_rows + 7) & ~7ll) * dim * 4 < (1ull << 32)) {
    // classes per launch: the kernel addresses the bank chunk with 32-bit byte offsets (a fixed 2^20 overran them from
    // dim = 1024 and the error reached the caller after `out` had been zeroed)
    long long chunk_ll = ((1ll << 32) - 1) / ((long long)dim * 4) - 8;
    chunk_ll &= ~7ll;
    const int chunk = (int)(chunk_ll < (1 << 20) ? chunk_ll : (1 << 20));
    for (long long c0 = 0; c0 < n_cls; c0 += chunk) {
      const int nc = (int)(n_cls - c0 < chunk ? n_cls - c0 : chunk);
      const int rc = wd_launch_p8_retrieval(static_cast<const unsigned char*>(t_split) + (size_t)c0 * dim * 4, nc, e_split, (int)n_rows,
                                            dim, t_unscale, scale, bias, count, rows_per_img, out + c0, n_cls, range_flag, st);
      if (rc != WD_OK) return rc;
    }
    return WD_OK;
  }
  WdConvGemm p{};
  p.a = static_cast<const float*>(e_split);
  p.c = out;
  p.batch = 1; p.hin = 1; p.win = n_img * rows_per_img; p.cin = dim; p.lda = dim;
  p.kh = p.kw = p.stride = 1; p.hout = 1; p.wout = p.win;
  p.m = n_img * rows_per_img; p.n = n_cls; p.k = dim; p.ldc = n_cls;
  p.out_scale = 1.0f;
  const RetrArgs ra{scale, bias, count, out, rows_per_img, n_cls};
  return launch_pingpong<SVAR_RETRMAX, 3, 2>(p, t_split, t_unscale, st, ra);
}

int wd_launch_p8_similarity(const WdConvGemm& p, const void* t_split, float unscale, hipStream_t st);   // split_gemm_p8.hip

// wd_similarity_split (ABI 14): the region x text similarity GEMM on the fp16x3 256 x 256 kernel.
extern "C" int wd_similarity_split(const void* e_split, int64_t rows, const void* t_split, float unscale, float* out, int32_t n_cls,
                                   int32_t dim, int32_t ldo, int32_t seg_rows, int32_t seg_end0, int32_t seg_end1,
                                   const float* seg_scale, const float* seg_bias, int32_t sigmoid, uint32_t* range_flag, void* stream) {
  if (!e_split || !t_split || !out || rows <= 0 || rows > 0x7ffffff0LL || n_cls <= 0 || dim <= 0 || ldo < n_cls || !(unscale > 0.f))
    return WD_ERR_BAD_ARG;
  if (seg_rows < 0 || (seg_rows > 0 && (!seg_scale || !seg_bias || !(0 <= seg_end0 && seg_end0 <= seg_end1 && seg_end1 <= seg_rows))))
    return WD_ERR_BAD_ARG;
  WdConvGemm p{};
  p.a = static_cast<const float*>(e_split);
  p.c = out;
  p.batch = 1; p.hin = 1; p.win = (int)rows; p.cin = dim; p.lda = dim;
  p.kh = p.kw = p.stride = 1; p.hout = 1; p.wout = (int)rows;
  p.m = (int)rows; p.n = n_cls; p.k = dim; p.ldc = ldo;
  p.out_scale = 1.0f;
  p.sigmoid = sigmoid ? 1 : 0;
  p.seg_rows = seg_rows; p.seg_end0 = seg_end0; p.seg_end1 = seg_end1;
  for (int i = 0; i < 3; ++i) {
    p.seg_scale[i] = seg_rows > 0 ? seg_scale[i] : 1.0f;
    p.seg_bias[i] = seg_rows > 0 ? seg_bias[i] : 0.0f;
  }
  p.range_flag = range_flag;
  if (!p.sigmoid && p.seg_rows == 0) return WD_ERR_UNSUPPORTED;   // a plain product: wd_conv_gemm_split covers it
  return wd_launch_p8_similarity(p, t_split, unscale, static_cast<hipStream_t>(stream));
}
