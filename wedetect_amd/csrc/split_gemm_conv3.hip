// split_gemm_conv3.hip — fp16x3 implicit-GEMM for the 3 x 3 / stride 1 / pad 1 convolutions of the neck and the head on
// PRE-SPLIT activations (yolo_world_pafpn.py:587-605, 692-715; yolo_world_head.py:194-232), round 4.
//
// split_conv_pp_kernel (split_gemm_conv.hip) does im2col with the DMA's per-lane source address: every K stage of 16
// channels of ONE filter tap brings its own copy of the 256 activation rows global -> LDS — nine copies of the input per
// launch (345 MB for the 26 MB input of the 32 x 40 x 40, 128 -> 128 layer; its header) and one barrier per 12 MFMAs.  But
// the three taps of a filter ROW read the same pixels one column apart: in a row-major (b, ho, wo) tile, tap (kh, kw) of
// output pixel m is input pixel m + (kh - 1) W + (kw - 1).  Here a stage is one (filter row kh, 16-channel chunk):
//   * activation rows: the 256 pixels of the tile shifted by (kh - 1) W, PLUS one pixel on either side (258 rows, staged as
//     17 DMA groups of 16), brought in ONCE for the three taps — global -> LDS activation traffic / 3, and the kw = 0 / 2
//     taps read the same LDS rows one row lower / higher (fragment address + kw * 64 B; the XOR swizzle by (row / 4) & 3 stays
//     conflict-free under that shift: the four aligned quads of a ds_read_b128 lane group still land on four distinct
//     swizzle classes);
//   * what the shifted read gets wrong is the LEFT / RIGHT image border: for wo = 0 (kw = 0) and wo = W - 1 (kw = 2) the
//     neighbour in the flattened order is a pixel of the previous / next image row.  Those lanes' operand fragments are
//     zeroed in registers (8 v_cndmask per fragment pair, two of the three taps) — the zero padding the reference's conv
//     applies.  The TOP / BOTTOM border stays with the DMA: a staged row whose source row ho + kh - 1 lies outside the
//     image (or whose pixel lies outside the tensor) comes from the zero page;
//   * weights: the three taps' [BN x 16] blocks of the stage;
//   * 36 MFMAs per wave per barrier instead of 12, 6 DMA instructions per wave per stage (3 per 12 MFMAs before: 2 now),
//     ring of three stages, counted vmcnt waits.
// K ORDER.  Per output the products are summed filter row by filter row, inside a row 16-channel chunk by chunk, inside a
// chunk tap by tap — (kh, ci / 16, kw, ci % 16) — where every other conv kernel of this library runs (kh, kw, ci).  Same
// halves, same MFMA triple per k16 step; the sums differ in the last bits (1e-7 relative).  That is why this kernel could
// not exist before the margin-robust goldens of round 4 (tests/golden/make_golden.py: search_robust): index parity no longer
// hangs on the summation order.  tests/test_gpu_split.py::test_conv3_* holds it to the fp32-MFMA kernel within 2e-5 like
// every fp16x3 kernel, and to the tap-major kernel within 3e-5 of the output's rms (measured 3e-6 ... 1e-5).
// Split-K (the fixed two-way split of the <= 20 x 20 maps, engine.py): workgroup (tile, ks) walks the stages [ks * per, ...) and
// writes raw partial sums to ws[ks][m][n]; splitk_oct_reduce_kernel (split_gemm_conv.hip) adds them in split order and applies
// the epilogue.  Not covered (split_conv_pp_kernel keeps them): stride 2, 1 x 1, the deconv scatter.
#include <stdlib.h>
#include "split_epi_oct.h"

namespace {

constexpr int C3_BM = 256, C3_AG = 17;                     // activation rows per tile; DMA groups of 16 staged rows (258 used)
constexpr int C3_AROWS = C3_AG * 16, C3_ROWB = 64;

__device__ __forceinline__ void c3_dma16(unsigned lds_addr, const unsigned char* src) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

template <int TN, int NBUF_ = 3>
struct C3 {
  static constexpr int WROWS = 64 * TN;                    // weight rows per tap
  static constexpr int WG = WROWS / 16;                    // DMA groups per tap
  static constexpr int NG = C3_AG + 3 * WG;                // DMA groups per stage: 41 (TN = 2) / 29 (TN = 1)
  static constexpr int NJ = (NG + 7) / 8;                  // DMA instructions per wave per stage (uniform: spare slots fetch the zero page)
  static constexpr int A_BYTES = C3_AROWS * C3_ROWB;
  static constexpr int STAGE = A_BYTES + 3 * WROWS * C3_ROWB + 1024;   // + 1 KB that the spare slots write
  static constexpr int NBUF = NBUF_;                       // 3: one workgroup per CU, DMA two stages ahead; 2: two workgroups per CU, one ahead
  static constexpr int LDS = NBUF * STAGE;
  static_assert(LDS <= 160 * 1024, "LDS");
  static_assert(NJ == 6 || NJ == 4, "the counted waits of the K loop");
  static_assert(LDS >= 8 * 32 * EPI_LDT * 4, "operand LDS must hold one epilogue patch per wave");
};

#define C3_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

// STAG (round 6): the K loop of the 256 x 256 MLP kernel (split_gemm_p8.hip) applied to this one.  In the plain form all eight
// waves walk [DMA issue | 3 x (8 ds_reads, 12 MFMAs) | vmcnt | barrier] in step, so the two waves of a SIMD read LDS at the same time
// and feed the matrix pipe at the same time: per stage ~2 100 cycles of LDS work (24 ds_read_b128 x 8 waves + the DMA's writes) and
// 2 304 cycles of MFMAs are paid one after the other.  Here a tap is a phase = [8 ds_reads, DMA share, vmcnt] s_barrier [12 MFMAs]
// s_barrier, and the second row group (waves 4-7; waves w and w + 4 share a SIMD) runs one barrier behind the first: a SIMD's matrix
// pipe is fed by one wave while its partner reads and requests.  The DMA of stage s + 2 goes into the buffer of stage s - 1 and is
// issued in the tap-1 / tap-2 phases of stage s (three instructions each): the other group's last reads of that buffer were
// consumed by MFMAs at least one barrier earlier.  Left / right image border: instead of zeroing fragments in registers the
// border lanes READ a staged row that is always zero (rows 258 .. 271 of a stage: their DMA source is the zero page).
// Per accumulator the MFMA sequence is the plain form's: results are bit-identical to it.
template <int TN, bool CSPLIT, int NB = 3, bool STAG = false>
__global__ void __launch_bounds__(512, NB == 2 ? 4 : 2)
split_conv3_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero, int k16,
                   float unscale, int nbn, int ksplits, float* __restrict__ ws) {
  using T = C3<TN, NB>;
  constexpr int TM = 2, BM = C3_BM, BN = 64 * TN, ROWB = C3_ROWB, NJ = T::NJ, NBUF = T::NBUF, DIST = NBUF - 1;
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
  const int nci = p.cin >> 4, nk_all = 3 * nci;             // stages: (kh, 16-channel chunk)
  const int per_split = (nk_all + ksplits - 1) / ksplits;
  const int s_begin = ks * per_split;
  const int nk = (s_begin + per_split < nk_all ? s_begin + per_split : nk_all) - s_begin;

  // ---- DMA slots of this wave: slot j handles stage group g = wave + 8 j.  g < 17: activation rows [16 g, + 16) of the
  // staged window (staged row r <-> centre pixel m0 - 1 + r); 17 <= g < NG: tap (g - 17) / WG, weight rows
  // [16 ((g - 17) % WG), + 16) of the column tile; g >= NG: spare.  lane = (row in the group, 16-byte slot); the slot
  // holds memory chunk (slot ^ ((row / 4) & 3)) in the [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15] order of the operand rows.
  const unsigned char* base[NJ];
  unsigned vmask[NJ];                                       // activation rows: bit kh = source row ho + kh - 1 is inside the image
  int kind[NJ], tapoff[NJ];                                 // wave-uniform: 0 = activation, 1 = weight, 2 = spare; weight: tap * cin * 4
  unsigned ldst[NJ];
  const unsigned char* zp = reinterpret_cast<const unsigned char*>(zero);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int g = wave + 8 * j;
    const int rl = lane >> 2;
    if (g < C3_AG) {
      const int row = g * 16 + rl;
      const int logical = (lane & 3) ^ ((row >> 2) & 3);
      const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
      const long long mc = (long long)m0 - 1 + row;
      const bool ok = row < BM + 2 && mc >= 0 && mc < p.m;
      const long long mm = ok ? mc : 0;
      const int ho = (int)((mm / p.wout) % p.hout);
      base[j] = reinterpret_cast<const unsigned char*>(p.a) + mm * p.lda * 4 + memchunk * 16;
      unsigned mk = 0;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) mk |= ((unsigned)(ho + kh - 1) < (unsigned)p.hin) ? (1u << kh) : 0u;
      vmask[j] = ok ? mk : 0u;
      kind[j] = 0; tapoff[j] = 0;
      ldst[j] = (unsigned)(g * 1024);
    } else if (g < T::NG) {
      const int gw = g - C3_AG, tap = gw / T::WG, r16 = gw - tap * T::WG;
      const int row = r16 * 16 + rl;                        // row inside the tap's block: the swizzle the fragment reads expect
      const int logical = (lane & 3) ^ ((row >> 2) & 3);
      const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
      const int n = n0 + row;
      const bool ok = n < p.n;
      base[j] = wsp + (size_t)(ok ? n : 0) * k16 * 4 + memchunk * 16;
      vmask[j] = ok ? 7u : 0u;
      kind[j] = 1; tapoff[j] = tap * p.cin * 4;
      ldst[j] = (unsigned)(T::A_BYTES + (tap * T::WROWS + r16 * 16) * ROWB);
    } else {
      base[j] = zp; vmask[j] = 0u; kind[j] = 2; tapoff[j] = 0;
      ldst[j] = (unsigned)(T::STAGE - 1024);
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  int i_kh = s_begin / nci, i_ci = (s_begin - (s_begin / nci) * nci) * 16;   // stage cursor of the DMA stream
  // instructions [J0, J1) of this wave's share of the stage under the cursor; LAST moves the cursor on
  auto issue_part = [&](int buf, auto j0_c, auto j1_c, auto last_c) {
    constexpr int J0 = decltype(j0_c)::value, J1 = decltype(j1_c)::value;
    const unsigned lbase = lds0 + buf * T::STAGE;
    const int a_off = ((i_kh - 1) * p.wout * p.lda + i_ci) * 4;          // hout == hin, wout == win (stride 1, pad 1)
    const int w_off = (i_kh * 3 * p.cin + i_ci) * 4;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      const int off = kind[j] == 0 ? a_off : w_off + tapoff[j];
      const unsigned bit = kind[j] == 0 ? (vmask[j] >> i_kh) & 1u : vmask[j] & 1u;
      c3_dma16(__builtin_amdgcn_readfirstlane(lbase + ldst[j]), bit ? base[j] + off : zp);
    }
    if constexpr (decltype(last_c)::value) {
      i_ci += 16;
      if (i_ci == p.cin) { i_ci = 0; ++i_kh; }
    }
  };
  using IC0 = std::integral_constant<int, 0>;
  using ICH = std::integral_constant<int, NJ / 2>;
  using ICN = std::integral_constant<int, NJ>;
  auto issue = [&](int buf) { issue_part(buf, IC0{}, ICN{}, std::true_type{}); };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- fragment addresses.  Activation: output pixel (local row q) reads staged row q + kw for tap kw.
  int aoff_h[3][TM], aoff_l[3][TM], boff_h[TN], boff_l[TN];
  bool zl[TM], zr[TM];                                      // this lane's pixel sits in the first / last image column
  const int hsel = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int q = group * 128 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = q + kw, f = (row >> 2) & 3;
      aoff_h[kw][i] = row * ROWB + ((hsel ^ f) << 4);
      aoff_l[kw][i] = row * ROWB + (((2 + hsel) ^ f) << 4);
    }
    const long long m = (long long)m0 + q;
    const int wo = (int)(m % p.wout);
    zl[i] = wo == 0;
    zr[i] = wo == p.wout - 1;
    if (STAG) {                                             // border lanes read the all-zero staged row 264 (any 16-byte slot of it)
      constexpr int ZROW = (C3_BM + 8) * ROWB;
      if (zl[i]) { aoff_h[0][i] = ZROW; aoff_l[0][i] = ZROW + 32; }
      if (zr[i]) { aoff_h[2][i] = ZROW; aoff_l[2][i] = ZROW + 32; }
    }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 32 * TN + j * 32 + (lane & 31), f = (row >> 2) & 3;
    boff_h[j] = T::A_BYTES + row * ROWB + ((hsel ^ f) << 4);
    boff_l[j] = T::A_BYTES + row * ROWB + (((2 + hsel) ^ f) << 4);
  }
  const h8 zero8 = {};
  auto tap = [&](const unsigned char* sp, int kw) {
    h8 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      xh[i] = *reinterpret_cast<const h8*>(sp + aoff_h[kw][i]);
      xl[i] = *reinterpret_cast<const h8*>(sp + aoff_l[kw][i]);
      if (kw == 0) { xh[i] = zl[i] ? zero8 : xh[i]; xl[i] = zl[i] ? zero8 : xl[i]; }
      if (kw == 2) { xh[i] = zr[i] ? zero8 : xh[i]; xl[i] = zr[i] ? zero8 : xl[i]; }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      wh[j] = *reinterpret_cast<const h8*>(sp + kw * (T::WROWS * ROWB) + boff_h[j]);
      wl[j] = *reinterpret_cast<const h8*>(sp + kw * (T::WROWS * ROWB) + boff_l[j]);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
  };

  if constexpr (STAG) {
    static_assert(NBUF == 3, "the staggered form refills the buffer of stage s - 1 while stage s is read");
#ifdef C3_TRACE
    // timing-only build (scripts/conv3_trace.py): waves 0 and 4 of tile 0 stamp s_memtime on arrival at and on release from every
    // barrier into ws[(wave / 4) * 4096 + i] (64-bit ticks = shader cycles)
    unsigned long long* trc = (tile == 0 && ks == 0 && (wave & 3) == 0 && ws) ? reinterpret_cast<unsigned long long*>(ws) + (wave >> 2) * 4096 : nullptr;
    int trc_i = 0;
#define C3_STAMP() do { if (trc && lane == 0 && trc_i < 4096) trc[trc_i] = __builtin_readcyclecounter(); ++trc_i; } while (0)
#define C3_SBARRIER() do { C3_STAMP(); C3_BARRIER(); C3_STAMP(); } while (0)
    C3_STAMP();
#else
#define C3_STAMP() do {} while (0)
#define C3_SBARRIER() C3_BARRIER()
#endif
    h8 xh[TM], xl[TM], wh[TN], wl[TN];
    auto rd = [&](const unsigned char* sp, int kw) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        xh[i] = *reinterpret_cast<const h8*>(sp + aoff_h[kw][i]);
        xl[i] = *reinterpret_cast<const h8*>(sp + aoff_l[kw][i]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wh[j] = *reinterpret_cast<const h8*>(sp + kw * (T::WROWS * ROWB) + boff_h[j]);
        wl[j] = *reinterpret_cast<const h8*>(sp + kw * (T::WROWS * ROWB) + boff_l[j]);
      }
    };
    auto mm = [&]() {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[j], xh[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xl[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[j], xh[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    if (0 < nk) issue(0);
    if (1 < nk) issue(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    C3_SBARRIER();
    if (group == 1) C3_SBARRIER();                            // the second row group runs one barrier behind
    int bcur = 0, bfill = 2;
    for (int s = 0; s < nk; ++s) {
      const unsigned char* sp = smem_raw + bcur * T::STAGE;
      const bool more = s + 2 < nk;
      rd(sp, 0);
      C3_SBARRIER();
      mm();
      C3_SBARRIER();
      rd(sp, 1);
      if (more) issue_part(bfill, IC0{}, ICH{}, std::false_type{});
      C3_SBARRIER();
      mm();
      C3_SBARRIER();
      rd(sp, 2);
      if (more) {
        issue_part(bfill, ICH{}, ICN{}, std::true_type{});
        if constexpr (NJ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      C3_SBARRIER();
      mm();
      C3_SBARRIER();
      bcur = bcur == 2 ? 0 : bcur + 1;
      bfill = bfill == 2 ? 0 : bfill + 1;
    }
    if (group == 0) C3_SBARRIER();                            // pairs with the extra barrier of group 1
    C3_SBARRIER();
  } else {
  // prologue: stages 0 .. DIST - 1 in flight, all landed before anyone reads
#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (d < nk) issue(d);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // steady state, ONE barrier per stage of 36 MFMAs: [taps 0 .. 2 of stage s | DMA of stage s + 2 into the buffer everybody
  // left before the last barrier] -> counted wait for the own share of stage s + 1 -> barrier
  int bcur = 0, bfill = DIST % NBUF;
  for (int s = 0; s < nk; ++s) {
    const unsigned char* sp = smem_raw + bcur * T::STAGE;
    if (s + DIST < nk) issue(bfill);
    __builtin_amdgcn_sched_barrier(0);
    tap(sp, 0);
    tap(sp, 1);
    tap(sp, 2);
    __builtin_amdgcn_sched_barrier(0);
    // issued so far: up to stage min(s + 2, nk - 1); stage s + 1 must be complete, stage s + 2 (NJ instructions of this wave)
    // may stay in flight
    if (DIST == 2 && s + 2 <= nk - 1) {
      if constexpr (NJ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    bcur = bcur == NBUF - 1 ? 0 : bcur + 1;
    bfill = bfill == NBUF - 1 ? 0 : bfill + 1;
  }
  __syncthreads();
  }

  const int mw = m0 + group * 128 + wm * 64, nw = n0 + wn * 32 * TN;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
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
  epi_oct_all<TM, TN, CSPLIT, TN == 1>(p, unscale, mw, nw, lane, acc, patch);
#ifdef C3_TRACE
  if constexpr (STAG) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tile == 0 && ks == 0 && (wave & 3) == 0 && ws && lane == 0) {
      unsigned long long* trc = reinterpret_cast<unsigned long long*>(ws) + (wave >> 2) * 4096;
      trc[4095] = __builtin_readcyclecounter();
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// split_conv3w_kernel (round 6, production form of the ring-of-three tiles): twelve waves — eight MFMA waves and four DMA waves.
// What the in-kernel timelines of the eight-wave forms say (scripts/conv3_trace.py, profiles/r06_conv3_trace.txt; shader cycles,
// 32 x 40 x 40, 128 -> 128): the staggered form's read phases that carry three LDS-DMA instructions take 650 - 880 cycles (a
// global_load_lds costs its issuing wave ~140), the bare ones 340, an MFMA phase ~400 + ~110 of barrier release: a stage costs
// ~4 900 cycles against 2 304 of MFMAs.  Steps from there, each measured (profiles/r06_conv3_forms.txt):
//   * the DMA moved to four producer waves (one per SIMD; 168 registers per wave), consumers still two row groups one barrier
//     apart: stage ~4 000 cycles instrumented, no faster un-instrumented — an interval is then MFMA phase + barrier release;
//   * consumers free-running with ONE barrier per stage and two fragment sets (the reads of tap t + 1 issued before the MFMAs
//     of tap t): no faster either — hipcc puts the eight ds_reads in front of the twelve MFMAs, and the two waves of a SIMD, in
//     step, issue their reads at the same time;
//   * the same with the reads pinned INTO the MFMA stream by inline asm, one ds_read_b128 behind each of the first eight MFMAs
//     of a tap (an MFMA's 32-cycle issue shadow hides it): stage 2 820 - 2 940 cycles, 57 us against 65 (in-step) / 60
//     (staggered); the big head convs 653 against 722 us.  This is the form below;
//   * LDS counters instead of the per-stage barrier (no s_barrier in the loop at all): equal within noise — dropped.
// Schedule.  Producers: stage k + 2 goes into the buffer of stage k - 1 right after BARRIER(k) — the consumers arrive there with
// lgkmcnt(0), i.e. their reads of stage k - 1 have RETURNED — and the counted vmcnt wait before BARRIER(k + 1) stands between stage
// k + 1 and its first reader.  Consumers: [tap 0 | tap 1 | lgkmcnt(0), BARRIER(s + 1) | tap 2]; tap 2 carries the reads of the next
// stage's tap 0.  The asm ds_reads are invisible to the compiler's waitcnt pass: every tap starts with lgkmcnt(0) (the only reads
// in flight then are the ones issued one tap earlier) and the build runs scripts/check_asm_ds_reads.py over the ISA.
// Same MFMA chain per accumulator as split_conv3_kernel: bit-identical results.
// ---------------------------------------------------------------------------------------------------------------------------
template <int TN, bool CSPLIT>
__global__ void __launch_bounds__(768, 1)
split_conv3w_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero, int k16,
                    float unscale, int nbn, int ksplits, float* __restrict__ ws) {
  using T = C3<TN, 3>;
  constexpr int TM = 2, BM = C3_BM, BN = 64 * TN, ROWB = C3_ROWB;
  constexpr int NJP = (T::NG + 3) / 4;                      // DMA instructions per producer wave per stage: 11 (TN = 2) / 8 (TN = 1)
  static_assert(4 * NJP * 1024 <= T::STAGE + 3 * 1024, "spare slots");
  static_assert(NJP == 11 || NJP == 8, "the counted waits of the producer loop");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
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
  const int nci = p.cin >> 4, nk_all = 3 * nci;             // stages: (kh, 16-channel chunk)
  const int per_split = (nk_all + ksplits - 1) / ksplits;
  const int s_begin = ks * per_split;
  const int nk = (s_begin + per_split < nk_all ? s_begin + per_split : nk_all) - s_begin;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
#ifdef C3_TRACE
  unsigned long long* trc = (tile == 0 && ks == 0 && (wave & 3) == 0 && ws) ? reinterpret_cast<unsigned long long*>(ws) + (wave >> 2) * 4096 : nullptr;
  int trc_i = 0;
  C3_STAMP();
#endif

  if (wave >= 8) {
    // ---- producer: slot j of producer pw handles stage group g = pw + 4 j (the group -> rows map of split_conv3_kernel)
    const int pw = wave - 8;
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(zero);
    const unsigned char* base[NJP];
    unsigned vmask[NJP];
#pragma unroll
    for (int j = 0; j < NJP; ++j) {
      const int g = pw + 4 * j;
      const int rl = lane >> 2;
      if (g < C3_AG) {
        const int row = g * 16 + rl;
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
        const int mc = m0 - 1 + row;                         // fits: the launcher refuses p.m + 2 BM >= 2^31
        const bool ok = row < BM + 2 && mc >= 0 && mc < p.m;
        const unsigned mm = ok ? (unsigned)mc : 0u;
        const int ho = (int)((mm / (unsigned)p.wout) % (unsigned)p.hout);
        base[j] = reinterpret_cast<const unsigned char*>(p.a) + (size_t)mm * p.lda * 4 + memchunk * 16;
        unsigned mk = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) mk |= ((unsigned)(ho + kh - 1) < (unsigned)p.hin) ? (1u << kh) : 0u;
        vmask[j] = ok ? mk : 0u;
      } else if (g < T::NG) {
        const int gw = g - C3_AG, tap = gw / T::WG, r16 = gw - tap * T::WG;
        const int row = r16 * 16 + rl;
        const int logical = (lane & 3) ^ ((row >> 2) & 3);
        const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
        const int n = n0 + row;
        const bool ok = n < p.n;
        base[j] = wsp + (size_t)(ok ? n : 0) * k16 * 4 + memchunk * 16 + (size_t)tap * p.cin * 4;
        vmask[j] = ok ? 7u : 0u;
      } else {
        base[j] = zp; vmask[j] = 0u;
      }
    }
    int i_kh = s_begin / nci, i_ci = (s_begin - (s_begin / nci) * nci) * 16;   // stage cursor of the DMA stream
    auto issue = [&](int buf) {
      const unsigned lbase = lds0 + buf * T::STAGE;
      const int a_off = ((i_kh - 1) * p.wout * p.lda + i_ci) * 4;
      const int w_off = (i_kh * 3 * p.cin + i_ci) * 4;
#pragma unroll
      for (int j = 0; j < NJP; ++j) {
        const int g = pw + 4 * j;                              // wave-uniform
        const bool is_a = g < C3_AG;
        const unsigned ldst = is_a ? (unsigned)(g * 1024)
                                   : (g < T::NG ? (unsigned)(T::A_BYTES + (((g - C3_AG) / T::WG) * T::WROWS + ((g - C3_AG) % T::WG) * 16) * ROWB)
                                                : (unsigned)(T::STAGE - 1024));
        const unsigned bit = is_a ? (vmask[j] >> i_kh) & 1u : vmask[j] & 1u;
        c3_dma16(__builtin_amdgcn_readfirstlane(lbase + ldst), bit ? base[j] + (is_a ? a_off : w_off) : zp);
      }
      i_ci += 16;
      if (i_ci == p.cin) { i_ci = 0; ++i_kh; }
    };
    if (0 < nk) issue(0);
    if (1 < nk) issue(1);
    int bfill = 2;
    for (int k = 0; k < nk; ++k) {
      if (k + 1 < nk) {
        if constexpr (NJP == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");   // stage k has landed, stage k + 1 may fly
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      C3_SBARRIER();                                           // BARRIER(k)
      if (k + 2 < nk) issue(bfill);
      bfill = bfill == 2 ? 0 : bfill + 1;
    }
    C3_SBARRIER();
    return;
  }

  // ---- consumers
  const int group = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // fragment addresses (LDS byte addresses inside a stage): output pixel (local row q) reads staged row q + kw for tap kw; the
  // left / right border lanes read the all-zero staged row 264 (its DMA source is the zero page)
  unsigned aoff_h[3][TM], aoff_l[3][TM], boff_h[TN], boff_l[TN];
  const int hsel = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int q = group * 128 + wm * 64 + i * 32 + (lane & 31);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int row = q + kw, f = (row >> 2) & 3;
      aoff_h[kw][i] = lds0 + row * ROWB + ((hsel ^ f) << 4);
      aoff_l[kw][i] = lds0 + row * ROWB + (((2 + hsel) ^ f) << 4);
    }
    const unsigned wo = (unsigned)(m0 + q) % (unsigned)p.wout;
    constexpr unsigned ZROW = (C3_BM + 8) * ROWB;
    if (wo == 0u) { aoff_h[0][i] = lds0 + ZROW; aoff_l[0][i] = lds0 + ZROW + 32; }
    if (wo == (unsigned)p.wout - 1u) { aoff_h[2][i] = lds0 + ZROW; aoff_l[2][i] = lds0 + ZROW + 32; }
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * 32 * TN + j * 32 + (lane & 31), f = (row >> 2) & 3;
    boff_h[j] = lds0 + T::A_BYTES + row * ROWB + ((hsel ^ f) << 4);
    boff_l[j] = lds0 + T::A_BYTES + row * ROWB + (((2 + hsel) ^ f) << 4);
  }
  // two fragment sets: tap t computes on set t & 1 while the reads of tap t + 1 fill the other
  h8 xh[2][TM] = {}, xl[2][TM] = {}, wh[2][TN] = {}, wl[2][TN] = {};
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  // one tap on set C; the 2 TM + 2 TN reads of tap KWN of the stage at byte offset `nb` go to set 1 - C, one behind each of the first MFMAs
  auto tap = [&](unsigned nb, auto kwn_c, auto cur_c) {
    constexpr int KWN = decltype(kwn_c)::value, C = decltype(cur_c)::value, N = 1 - C;
    constexpr int NR = 2 * TM + 2 * TN;
    unsigned ra[NR];
#pragma unroll
    for (int i = 0; i < TM; ++i) { ra[2 * i] = nb + aoff_h[KWN][i]; ra[2 * i + 1] = nb + aoff_l[KWN][i]; }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
      ra[2 * TM + 2 * jj] = nb + KWN * (T::WROWS * ROWB) + boff_h[jj];
      ra[2 * TM + 2 * jj + 1] = nb + KWN * (T::WROWS * ROWB) + boff_l[jj];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int m = 0; m < 3 * TM * TN; ++m) {
      const int pass = m / (TM * TN), i = (m % (TM * TN)) / TN, jj = m % TN;
      if (pass == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][jj]) : "v"(wl[C][jj]), "v"(xh[C][i]));
      else if (pass == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][jj]) : "v"(wh[C][jj]), "v"(xl[C][i]));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i][jj]) : "v"(wh[C][jj]), "v"(xh[C][i]));
      // TN = 1 has six MFMAs for six reads; TN = 2 twelve for eight
      if (m < NR) {
        const int r = m;
        if (r < 2 * TM) {
          if (r & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(xl[N][r >> 1]) : "v"(ra[r]) : "memory");
          else asm volatile("ds_read_b128 %0, %1" : "=v"(xh[N][r >> 1]) : "v"(ra[r]) : "memory");
        } else {
          if (r & 1) asm volatile("ds_read_b128 %0, %1" : "=v"(wl[N][(r - 2 * TM) >> 1]) : "v"(ra[r]) : "memory");
          else asm volatile("ds_read_b128 %0, %1" : "=v"(wh[N][(r - 2 * TM) >> 1]) : "v"(ra[r]) : "memory");
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // stage in the buffer at byte offset sb (the next one at sbn), starting on fragment set P0 (three taps: the sets swap per stage)
  auto stage = [&](unsigned sb, unsigned sbn, auto p0_c) {
    constexpr int P0 = decltype(p0_c)::value;
    using S0 = std::integral_constant<int, P0>;
    using S1 = std::integral_constant<int, 1 - P0>;
    tap(sb, K1{}, S0{});                                     // tap 0, reads tap 1
    tap(sb, K2{}, S1{});                                     // tap 1, reads tap 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every read of this stage has returned
    C3_SBARRIER();                                           // BARRIER(s + 1); after the last stage: the ring becomes epilogue patches
    tap(sbn, K0{}, S0{});                                    // tap 2, reads tap 0 of the next stage (after the last stage: stale bytes, never used)
  };
  C3_SBARRIER();                                             // BARRIER(0): stage 0 has landed
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(xh[0][i]) : "v"(aoff_h[0][i]) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(xl[0][i]) : "v"(aoff_l[0][i]) : "memory");
  }
#pragma unroll
  for (int jj = 0; jj < TN; ++jj) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(wh[0][jj]) : "v"(boff_h[jj]) : "memory");
    asm volatile("ds_read_b128 %0, %1" : "=v"(wl[0][jj]) : "v"(boff_l[jj]) : "memory");
  }
  int bcur = 0;
  for (int s = 0; s < nk; s += 2) {                            // two stages per trip
    const int b1 = bcur == 2 ? 0 : bcur + 1, b2 = b1 == 2 ? 0 : b1 + 1;
    stage((unsigned)(bcur * T::STAGE), (unsigned)(b1 * T::STAGE), K0{});
    if (s + 1 < nk) stage((unsigned)(b1 * T::STAGE), (unsigned)(b2 * T::STAGE), K1{});
    bcur = b2;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the reads issued beside the last tap (unused) have returned

  const int mw = m0 + group * 128 + wm * 64, nw = n0 + wn * 32 * TN;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  if (ksplits > 1) {
    WdConvGemm pr = p;                                 // raw partial sums, plain rows [m][n] of this split
    pr.bias = nullptr; pr.res = nullptr; pr.c2 = nullptr; pr.range_flag = nullptr;
    pr.c = ws + (size_t)ks * p.m * p.n; pr.ldc = p.n;
    pr.out_mode = WD_OUT_ROWS; pr.c_batch_stride = 0; pr.seg_rows = 0; pr.sigmoid = 0; pr.out_scale = 1.0f; pr.out_bias = 0.0f;
    pr.act = WD_ACT_NONE;
    EpiOctOperands<TM, TN, false> ops;
    ops.load(pr, mw, nw, lane);
    EpiOctWalk<0, TM, TN, WD_ACT_NONE, false, false, false>::run(pr, 1.0f, mw, nw, lane, acc, patch, ops);
    return;
  }
  epi_oct_all<TM, TN, CSPLIT, TN == 1>(p, unscale, mw, nw, lane, acc, patch);
#ifdef C3_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (trc && lane == 0) trc[4095] = __builtin_readcyclecounter();
#endif
}

template <int TN, bool CSPLIT>
int launch_conv3w(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, int ksplits = 1, float* ws = nullptr) {
  using T = C3<TN, 3>;
  constexpr int BN = 64 * TN;
  const int nbm = (p.m + C3_BM - 1) / C3_BM, nbn = (p.n + BN - 1) / BN;
  const long long nblk = (long long)nbm * nbn * ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffLL || (long long)p.m + 2 * C3_BM >= 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  auto k = split_conv3w_kernel<TN, CSPLIT>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), T::LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(768), T::LDS, st, p, static_cast<const unsigned char*>(wsp), zero, k16, unscale, nbn,
                 ksplits, ws);
  return wd_launch_status();
}

template <int TN, bool CSPLIT, int NB = 3, bool STAG = false>
int launch_conv3(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, int ksplits = 1, float* ws = nullptr) {
  using T = C3<TN, NB>;
  constexpr int BN = 64 * TN;
  const int nbm = (p.m + C3_BM - 1) / C3_BM, nbn = (p.n + BN - 1) / BN;
  const long long nblk = (long long)nbm * nbn * ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  auto k = split_conv3_kernel<TN, CSPLIT, NB, STAG>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), T::LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(512), T::LDS, st, p, static_cast<const unsigned char*>(wsp), zero, k16, unscale, nbn,
                 ksplits, ws);
  return wd_launch_status();
}

}  // namespace

// 3 x 3 / stride 1 / pad 1 on pre-split activations, no split-K: the row-sharing kernel applies
bool wd_conv3_ok(const WdConvGemm& p, int flags) {
  if (!(flags & WD_SPLIT_A)) return false;
  if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.hout != p.hin || p.wout != p.win) return false;
  if (p.cin % 16 || p.k != 9 * p.cin || p.lda % 8 || p.n % 8 || p.wout < 2) return false;
  if ((unsigned long long)p.m * p.lda * 4 >= (1ull << 40)) return false;
  return true;
}

int wd_launch_oct_reduce(const WdConvGemm& p, const float* ws, int splits, float unscale, bool csplit, hipStream_t st);   // split_gemm_conv.hip

// the staggered K loop (STAG) is the production form of the ring-of-three kernels; $WEDETECT_CONV3_STAG=0 keeps the in-step loop (A/B runs)
static bool conv3_stag_default() {
  static const bool on = [] { const char* e = getenv("WEDETECT_CONV3_STAG"); return !(e && e[0] == '0'); }();
  return on;
}

// $WEDETECT_CONV3_WS=0 keeps the eight-wave forms (A/B runs); default: the twelve-wave producer / consumer kernel
static bool conv3_ws_default() {
  static const bool on = [] { const char* e = getenv("WEDETECT_CONV3_WS"); return !(e && e[0] == '0'); }();
  return on;
}

// variant: 0 = production choice, 2 / 3 = ring depth of the narrow (BN = 64) form forced (A/B runs), 7 / 8 = eight waves, staggered /
// in-step K loop forced, 9 = the twelve-wave producer / consumer kernel forced
int wd_launch_conv3(const WdConvGemm& p, const void* w, float unscale, int flags, hipStream_t st, int variant, int ksplits,
                    float* ws, long long ws_floats) {
  if (!wd_conv3_ok(p, flags)) return WD_ERR_UNSUPPORTED;
  const bool csplit = (flags & WD_SPLIT_C) != 0;
  const bool w12 = variant == 9 || (variant == 0 && conv3_ws_default());
  const bool stag = variant == 7 || (variant != 8 && conv3_stag_default());
  const bool narrow = (p.n % 128) != 0 && ((p.n + 63) / 64) * 64 < ((p.n + 127) / 128) * 128;
  if (ksplits < 1) ksplits = 1;
#ifdef C3_TRACE
  float* tws = ws;
#else
  float* tws = nullptr;
#endif
  if (ksplits > 1) {
    if (!ws || (long long)ksplits * p.m * p.n > ws_floats) return WD_ERR_WORKSPACE;
    if (ksplits > 3 * (p.cin >> 4)) ksplits = 3 * (p.cin >> 4);
    int rc;
    if (w12) {
      if (narrow) rc = csplit ? launch_conv3w<1, true>(p, w, unscale, st, ksplits, ws) : launch_conv3w<1, false>(p, w, unscale, st, ksplits, ws);
      else rc = csplit ? launch_conv3w<2, true>(p, w, unscale, st, ksplits, ws) : launch_conv3w<2, false>(p, w, unscale, st, ksplits, ws);
    } else if (stag) {
      if (narrow) rc = csplit ? launch_conv3<1, true, 3, true>(p, w, unscale, st, ksplits, ws) : launch_conv3<1, false, 3, true>(p, w, unscale, st, ksplits, ws);
      else rc = csplit ? launch_conv3<2, true, 3, true>(p, w, unscale, st, ksplits, ws) : launch_conv3<2, false, 3, true>(p, w, unscale, st, ksplits, ws);
    } else {
      if (narrow) rc = csplit ? launch_conv3<1, true>(p, w, unscale, st, ksplits, ws) : launch_conv3<1, false>(p, w, unscale, st, ksplits, ws);
      else rc = csplit ? launch_conv3<2, true>(p, w, unscale, st, ksplits, ws) : launch_conv3<2, false>(p, w, unscale, st, ksplits, ws);
    }
    if (rc != WD_OK) return rc;
    return wd_launch_oct_reduce(p, ws, ksplits, unscale, csplit, st);
  }
  if (narrow) {
    // BN = 64: a stage is 30 KB.  With more tiles than CUs a ring of TWO stages (60 KB) lets two eight-wave workgroups share a CU —
    // one's prologue / epilogue under the other's K loop: 32 x 80 x 80 64 -> 64 85.9 -> 72.2 us, 128 -> 64 133 -> 114 us
    // (profiles/r04_conv3.txt; the twelve-wave kernel, one workgroup per CU: 86 / 128 us); with at most one tile per CU the
    // deeper ring wins
    const long long tiles = (long long)((p.m + C3_BM - 1) / C3_BM) * ((p.n + 63) / 64);
    const bool ring2 = variant == 2 || ((variant == 0 || variant == 8) && tiles > 256);
    if (variant == 9) return csplit ? launch_conv3w<1, true>(p, w, unscale, st, 1, tws) : launch_conv3w<1, false>(p, w, unscale, st, 1, tws);
    if (ring2) return csplit ? launch_conv3<1, true, 2>(p, w, unscale, st) : launch_conv3<1, false, 2>(p, w, unscale, st);
    if (w12) return csplit ? launch_conv3w<1, true>(p, w, unscale, st, 1, tws) : launch_conv3w<1, false>(p, w, unscale, st, 1, tws);
    if (stag) return csplit ? launch_conv3<1, true, 3, true>(p, w, unscale, st) : launch_conv3<1, false, 3, true>(p, w, unscale, st);
    return csplit ? launch_conv3<1, true>(p, w, unscale, st) : launch_conv3<1, false>(p, w, unscale, st);
  }
  if (w12) return csplit ? launch_conv3w<2, true>(p, w, unscale, st, 1, tws) : launch_conv3w<2, false>(p, w, unscale, st, 1, tws);
  if (stag) return csplit ? launch_conv3<2, true, 3, true>(p, w, unscale, st, 1, tws) : launch_conv3<2, false, 3, true>(p, w, unscale, st, 1, tws);
  return csplit ? launch_conv3<2, true>(p, w, unscale, st) : launch_conv3<2, false>(p, w, unscale, st);
}

// the kernel form wd_launch_conv3 picks for an m x n layer without split-K (bench.py's gemm_kernels table)
const char* wd_conv3_config_name(int m, int n) {
  const bool narrow = (n % 128) != 0 && ((n + 63) / 64) * 64 < ((n + 127) / 128) * 128;
  const bool w12 = conv3_ws_default();
  if (narrow) {
    const long long tiles = (long long)((m + C3_BM - 1) / C3_BM) * ((n + 63) / 64);
    if (tiles > 256) return "fp16x3 256x64x16/8w/dma3/ring2";
    return w12 ? "fp16x3 256x64x16/12w/dma3" : "fp16x3 256x64x16/8w/dma3";
  }
  return w12 ? "fp16x3 256x128x16/12w/dma3" : "fp16x3 256x128x16/8w/dma3";
}
