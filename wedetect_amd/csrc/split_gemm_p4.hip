// split_gemm_p4.hip — fp16x3 GEMM for plain layers with pre-split operands (the ConvNeXt pointwise MLPs,
// mm_backbone.py:115-124), TWO independent workgroups per CU: 128 x 256 output tile, four waves (one per SIMD),
// K steps of 16, a three-slot LDS ring.
//
// Why next to the 256 x 256 kernel (split_gemm_p8.hip).  That kernel owns the whole CU: while its eight waves run
// the epilogue — GELU + fp16 split + 420 MB of stores for a pwconv1 launch, a third of the launch — the matrix
// pipes idle, and while they run the K loop HBM idles.  Here a wave keeps the same 128 x 64 output block (eight
// 32 x 32 accumulators, the same ds_read_b128 : MFMA ratio), but a workgroup is only four waves and 72 KB of LDS,
// so two of them share a CU, each SIMD holds one wave of either, and nothing synchronises the two: one workgroup's
// epilogue and prologue run beside the other's K loop.  The price is 1.5 x the global -> LDS traffic per MFMA
// (128 + 256 operand rows per 128 x 256 outputs instead of 256 + 256 per 256 x 256).
//
//   * LDS ring of three K steps, 24 KB each: activation rows [0, 128) x 64 B at +0 (8 KB), weight rows
//     [0, 256) x 64 B at +8 KB (16 KB).  A row is the k16 slice of the [hi x 8 | lo x 8] layout: four 16-byte
//     chunks (2 kgrp + lo), stored at chunk ^ ((row >> 2) & 3) — conflict-free for ds_read_b128's 16-lane groups;
//     the XOR is applied on the GLOBAL side of the LDS-DMA (which writes lane-linear) and on the read address.
//   * a K step is two phases of 12 MFMAs (rows 0-63, then rows 64-127, both 32-column blocks of the wave):
//       phase a: read 4 activation + 4 weight fragments; request [A rows 64-127, W rows 128-255] of step t + 2
//       phase b: read 4 activation fragments (weights stay in registers); request [W rows 0-127, A rows 0-63] of t + 3
//     each request goes into the ring slot whose previous content every wave finished reading before the
//     preceding barrier; three 1 KB global_load_lds_dwordx4 per wave per phase.
//   * phase = [ds_reads, DMA issue, lgkmcnt(0), counted vmcnt] s_barrier [12 MFMAs]: ONE barrier per phase.  The
//     barrier certifies both that the operands of the next phase have landed (every wave retired its own requests up
//     to three phases back: vmcnt(9) in steady state, never 0) and that this phase's reads are complete.
//   * per-accumulator MFMA order: k ascending, lo.hi -> hi.lo -> hi.hi inside a k16 step — the order of every
//     other fp16x3 kernel: bit-identical results.
//
// Measured (profiles/r02_p4_ab.txt, us per launch, this kernel / the 256 x 256 kernel): stage-3 pwconv1 387 / 420,
// stage-4 pwconv1 333 / 352, stage-2 pwconv1 521 / 540 (ping-pong 550), stage-3 pwconv2 338 / 299.  One workgroup per
// CU alone runs 487 us on the stage-3 pwconv1 shape, two run 392: the second workgroup buys 1.24 x, not 2 x — the
// store stream of one workgroup's epilogue and the LDS-DMA stream of the other's K loop share the CU's vector-memory
// path.  In the step the C-split layers on this kernel are worth 0.4 % (44.95 vs 45.14 ms), so it is NOT the
// production choice: explicit configuration 66 only (scripts/p8_bench.py, tests).
#include "split_gemm_impl.h"

namespace {

constexpr int P4_ROWB = 64, P4_A = 128 * P4_ROWB, P4_W = 256 * P4_ROWB, P4_SLOT = P4_A + P4_W, P4_LDS = 3 * P4_SLOT;

__device__ __forceinline__ void p4_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

#define P4_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

template <int VAR>
__global__ void __launch_bounds__(256, 2)
split_gemm_p4_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, int k16, float unscale, int nbn,
                     int vec_c, int vec_res, int vec_bias, int ngrp, int nbm) {
  constexpr int TM = 4, TN = 2, BM = 128, BN = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nk = p.k >> 4;

  // ---- tile of this workgroup: XCD-contiguous ranges of the (column-group-major) tile list
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int gsz = ngrp * nbm;
  const int grp = tile / gsz, rem = tile - grp * gsz;
  const int bm = rem / ngrp, bn = grp * ngrp + (rem - bm * ngrp);
  const int m0 = bm * BM, n0 = bn * BN;

  const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.a);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const unsigned dma_dst = lds0 + wave * 1024;            // a wave's instruction fills 16 rows x 64 B

  // lane part of the DMA source offset: row (lane >> 2) of the 16-row group, physical chunk (lane & 3) holds the
  // logical chunk (lane & 3) ^ f, f = ((row >> 2) & 3) = (lane >> 4) & 3 (groups start at multiples of 16 rows)
  unsigned la, lw;
  {
    const int r16 = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    la = (unsigned)r16 * (unsigned)p.lda * 4u + chunk * 16;
    lw = (unsigned)r16 * (unsigned)k16 * 4u + chunk * 16;
  }
  // fragment read offsets inside a slot: row (lane & 31) of a 32-row block, chunk (2 hsel + lo) ^ ((lane >> 2) & 3)
  int a_hi, w_hi;
  {
    const int f = (lane >> 2) & 3, hsel = lane >> 5;
    a_hi = (lane & 31) * P4_ROWB + (((2 * hsel) ^ f) << 4);
    w_hi = P4_A + (wave * 64 + (lane & 31)) * P4_ROWB + (((2 * hsel) ^ f) << 4);
  }

  // request one half of an operand of K step kt into ring slot `slot`: is_w ? weight rows [half * 128, + 128) (two
  // instructions per wave) : activation rows [half * 64, + 64) (one)
  auto stage_a = [&](int kt, int slot, int half) {
    int row = m0 + half * 64 + wave * 16;
    const int limit = p.m - 16;
    row = row < limit ? row : limit;
    const unsigned char* src = abase + (size_t)row * ((size_t)p.lda * 4u) + (size_t)kt * P4_ROWB;
    p4_dma(dma_dst + slot * P4_SLOT + half * 4096, la, src);
  };
  auto stage_w = [&](int kt, int slot, int half) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = n0 + half * 128 + j * 64 + wave * 16;
      const int limit = p.n - 16;
      row = row < limit ? row : limit;
      const unsigned char* src = wsp + (size_t)row * ((size_t)k16 * 4u) + (size_t)kt * P4_ROWB;
      p4_dma(dma_dst + slot * P4_SLOT + P4_A + half * 8192 + j * 4096, lw, src);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  h8 xh[2], xl[2], wh[2], wl[2];
  auto read_a = [&](int slot, int half) {
    int b = a_hi;
    asm volatile("" : "+v"(b));
    const unsigned char* base = smem_raw + slot * P4_SLOT + half * 4096;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xh[i] = *reinterpret_cast<const h8*>(base + i * 2048 + b);
      xl[i] = *reinterpret_cast<const h8*>(base + i * 2048 + (b ^ 16));
    }
  };
  auto read_w = [&](int slot) {
    int b = w_hi;
    asm volatile("" : "+v"(b));
    const unsigned char* base = smem_raw + slot * P4_SLOT;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      wh[j] = *reinterpret_cast<const h8*>(base + j * 2048 + b);
      wl[j] = *reinterpret_cast<const h8*>(base + j * 2048 + (b ^ 16));
    }
  };
  // 12 MFMAs: rows [i0 * 32, + 64) x both column blocks; each accumulator gets lo.hi, hi.lo, hi.hi in that order
  auto mfma12 = [&](f32x16& c00, f32x16& c10, f32x16& c01, f32x16& c11) {
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], xh[0], c00, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], xh[1], c10, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1], xh[0], c01, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1], xh[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], xl[0], c00, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], xl[1], c10, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], xl[0], c01, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], xl[1], c11, 0, 0, 0);
    c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], xh[0], c00, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], xh[1], c10, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], xh[0], c01, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], xh[1], c11, 0, 0, 0);
  };
#define P4_SYNC(n)                                             \
  do {                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         \
    asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");      \
    P4_BARRIER();                                              \
  } while (0)

  // ---- prologue: the requests the steady state would have issued before step 0, in its order; step 0 must have
  // landed and been published before the first reads, the rest (9 instructions when nk >= 3) stays in flight
  stage_w(0, 0, 0); stage_a(0, 0, 0);                       // "phase b(-3)"
  stage_a(0, 0, 1); stage_w(0, 0, 1);                       // "phase a(-2)"
  if (nk > 1) { stage_w(1, 1, 0); stage_a(1, 1, 0); stage_a(1, 1, 1); stage_w(1, 1, 1); }
  if (nk > 2) {
    stage_w(2, 2, 0); stage_a(2, 2, 0);
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  P4_BARRIER();

  // slot of step kt = kt % 3, tracked incrementally: s0 = slot(kt), s2 = slot(kt + 2)
  int s0 = 0, s2 = 2;
  if (nk >= 3) {
    // steady state: both phases request; three phases of requests (9 instructions of this wave) stay in flight
    for (int kt = 0; kt + 3 < nk; ++kt) {
      read_a(s0, 0);
      read_w(s0);
      stage_a(kt + 2, s2, 1); stage_w(kt + 2, s2, 1);
      P4_SYNC(9);
      mfma12(acc[0][0], acc[1][0], acc[0][1], acc[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_a(s0, 1);
      stage_w(kt + 3, s0, 0); stage_a(kt + 3, s0, 0);
      P4_SYNC(9);
      mfma12(acc[2][0], acc[3][0], acc[2][1], acc[3][1]);
      __builtin_amdgcn_sched_barrier(0);
      s2 = s0; s0 = s0 == 2 ? 0 : s0 + 1;
    }
    // step nk - 3: phase a still requests (the second half of step nk - 1), phase b has nothing left
    read_a(s0, 0); read_w(s0);
    stage_a(nk - 1, s2, 1); stage_w(nk - 1, s2, 1);
    P4_SYNC(9);
    mfma12(acc[0][0], acc[1][0], acc[0][1], acc[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_a(s0, 1);
    P4_SYNC(6);
    mfma12(acc[2][0], acc[3][0], acc[2][1], acc[3][1]);
    __builtin_amdgcn_sched_barrier(0);
    s0 = s0 == 2 ? 0 : s0 + 1;
    // step nk - 2
    read_a(s0, 0); read_w(s0);
    P4_SYNC(3);
    mfma12(acc[0][0], acc[1][0], acc[0][1], acc[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_a(s0, 1);
    P4_SYNC(0);
    mfma12(acc[2][0], acc[3][0], acc[2][1], acc[3][1]);
    __builtin_amdgcn_sched_barrier(0);
    s0 = s0 == 2 ? 0 : s0 + 1;
    // step nk - 1: everything has landed and been published
    read_a(s0, 0); read_w(s0);
    mfma12(acc[0][0], acc[1][0], acc[0][1], acc[1][1]);
    __builtin_amdgcn_sched_barrier(0);
    read_a(s0, 1);
    mfma12(acc[2][0], acc[3][0], acc[2][1], acc[3][1]);
  } else {
    // short K (one or two steps): everything came with the prologue
    for (int kt = 0; kt < nk; ++kt) {
      read_a(s0, 0); read_w(s0);
      if (kt + 2 < nk) { stage_a(kt + 2, s2, 1); stage_w(kt + 2, s2, 1); }
      P4_SYNC(0);
      mfma12(acc[0][0], acc[1][0], acc[0][1], acc[1][1]);
      __builtin_amdgcn_sched_barrier(0);
      read_a(s0, 1);
      if (kt + 3 < nk) { stage_w(kt + 3, s0, 0); stage_a(kt + 3, s0, 0); }
      P4_SYNC(0);
      mfma12(acc[2][0], acc[3][0], acc[2][1], acc[3][1]);
      __builtin_amdgcn_sched_barrier(0);
      s2 = s0; s0 = s0 == 2 ? 0 : s0 + 1;
    }
  }
#undef P4_SYNC
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  P4_BARRIER();                                              // every read of the ring is over: LDS becomes epilogue patches

  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  const int mw = m0, nw = n0 + wave * 64;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));          // epilogue address arithmetic starts here, not before the K loop
  static_assert(P4_LDS >= 4 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
  if (VAR & SVAR_CSPLIT) {
    switch (p.act) {
      case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane_e, acc, patch); break;
      case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane_e, acc, patch); break;
      case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane_e, acc, patch); break;
      default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane_e, acc, patch); break;
    }
  } else if (epi_res_prefetch_ok(p, ev, nw, 64)) {
    split_epilogue_res_prefetch<TM, TN, 3>(p, ev, mw, nw, lane_e, acc, patch);
  } else {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue_lds<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
      case WD_ACT_SILU: split_epilogue_lds<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
      case WD_ACT_GELU: split_epilogue_lds<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
      default: split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane_e, acc, patch); break;
    }
  }
}

template <int VAR>
int launch_p4(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st) {
  const int nbm = (p.m + 127) / 128, nbn = (p.n + 255) / 256;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  if ((unsigned long long)p.m * p.lda * 4 >= (1ull << 32) || (unsigned long long)p.n * k16 * 4 >= (1ull << 32))
    return WD_ERR_UNSUPPORTED;                                     // 32-bit DMA offsets from the operand bases
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c);
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  int ngrp = 2;                                                    // column tiles walked in pairs
  if (ngrp > nbn || nbn % ngrp) ngrp = nbn;
  auto k = split_gemm_p4_kernel<VAR>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), P4_LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(256), P4_LDS, st, p, static_cast<const unsigned char*>(wsp), k16, unscale, nbn,
                 vec_c, vec_res, vec_bias, ngrp, nbm);
  return wd_launch_status();
}

}  // namespace

// cfg 66: plain 1x1 layer, both operands pre-split, K % 16 == 0, m % 16 == 0, n % 16 == 0.
int wd_launch_p4(const WdConvGemm& p, const void* w, float unscale, bool csplit, hipStream_t st) {
  if (p.k % 16 || p.k < 16 || p.lda % 8 || p.m % 16 || p.n % 16 || p.m < 16 || p.n < 16 || !wd_aligned16(p.a) || !wd_aligned16(w))
    return WD_ERR_UNSUPPORTED;                                     // DMA row groups of 16 are clamped as a whole
  if (csplit) {
    if (p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c)) return WD_ERR_BAD_ARG;
    return launch_p4<SVAR_CSPLIT>(p, w, unscale, st);
  }
  return launch_p4<0>(p, w, unscale, st);
}
