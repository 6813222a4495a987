// split_gemm_p8.hip — fp16x3 GEMM for the big plain layers with pre-split operands (the ConvNeXt
// pointwise MLPs, mm_backbone.py:115-124): 256 x 256 output tile, K tiles of 32, eight waves in two
// row groups running half a phase apart, four phases per K tile.
//
// Why another structure.  The 128 x 128 direct-to-LDS kernel (split_gemm_pre.hip) pays one
// workgroup barrier and one full vmcnt(0) drain per 12 MFMAs and reads 8 ds_read_b128 per 12
// MFMAs; its matrix pipe is busy a third of the time.  Here a wave owns 128 x 64 outputs (eight
// 32 x 32 accumulators = 128 registers), so a K tile of 32 costs 20 ds_read_b128 per 48 MFMAs,
// and global -> LDS traffic per MFMA halves.  The K loop never drains the DMA queue:
//
//   * LDS holds two K tiles of 64 KB, each cut into four 16 KB regions (128 rows x 128 B):
//       A0 / A1 = activation rows {0-63, 128-191} / {64-127, 192-255} of the tile (the first / second
//                 64 rows of each row group),  W0 / W1 = weight rows {64 wn + 0-31} / {64 wn + 32-63}.
//   * a K tile is consumed in four phases, one 64 x 32 output quadrant of the wave per phase
//     (12 MFMAs = 2 row blocks x 2 k16 steps x {lo.hi, hi.lo, hi.hi}):
//       phase 1  read A0 (8 x ds_read_b128) + W0 (4)   quadrant (rows 0-63,  cols 0-31)
//       phase 2  read W1 (4)                           quadrant (rows 0-63,  cols 32-63)
//       phase 3  read A1 (8)                           quadrant (rows 64-127, cols 32-63)
//       phase 4  -  (W0 still in registers)            quadrant (rows 64-127, cols 0-31)
//     The per-accumulator MFMA sequence (k ascending, lo.hi -> hi.lo -> hi.hi inside a k16 step) is the
//     one of every other fp16x3 kernel: results are bit-identical to theirs.
//   * every phase also re-fills one region that went dead two phases earlier with the data of two K
//     tiles ahead (phase 1: W1 of tile t+1, 2: A1 of t+1, 3: A0 of t+2, 4: W0 of t+2): 16 one-KB
//     global_load_lds_dwordx4 per region, two per wave, issued from inline asm so that hipcc does
//     not serialise them against the ds_reads.  A region is read five or six phases after it was
//     requested; each wave retires its own requests with a COUNTED s_waitcnt vmcnt(8) (the four
//     newest regions stay in flight across the barriers) and never waits vmcnt(0) in steady state.
//   * phase = [ds_reads, DMA issue, vmcnt] s_barrier [12 MFMAs] s_barrier.  The two row groups
//     (waves 0-3 / 4-7; waves w and w + 4 share a SIMD) are offset by one barrier, so a SIMD's
//     matrix pipe is fed by one wave while its partner reads and requests.  A region is re-filled
//     two phases after its last read: the other group's reads of it (issued up to one barrier later,
//     completed before its MFMAs) are over before the DMA can be issued.
//   * LDS rows are unpadded 128 B with the 16-byte slot XOR-swizzled by (row / 2) & 7 — applied on the
//     GLOBAL side of the DMA and on the ds_read address (LDS-DMA writes lane-linear); out-of-range
//     rows are clamped to the last valid row (their outputs are never stored), so no zero page.
#include "split_gemm_impl.h"

namespace {

constexpr int P8_ROWB = 128, P8_REGION = 128 * P8_ROWB, P8_TILE = 4 * P8_REGION, P8_LDS = 2 * P8_TILE;
constexpr int P8_A0 = 0, P8_W0 = P8_REGION, P8_W1 = 2 * P8_REGION, P8_A1 = 3 * P8_REGION;

// one 1 KB LDS-DMA: 64 lanes x 16 B from base + voff[lane] to LDS [lds_addr, +1024)
__device__ __forceinline__ void p8_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

#define P8_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

template <int VAR>
__global__ void __launch_bounds__(512, 2)
split_gemm_p8_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, int k16, float unscale, int nbn,
                     int vec_c, int vec_res, int vec_bias, int ngrp, int nbm) {
  constexpr int TM = 4, TN = 2, BM = 256, BN = 256, ROWB = P8_ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int group = wave >> 2, wn = wave & 3;
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
  const int nk = p.k >> 5;

  // ---- DMA sources: instruction j (0 / 1) of this wave fills slots [(2 wave + j) * 8, +8) of a region
  const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.a);
  unsigned va0[2], va1[2], vw0[2], vw1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int slot = (wave * 2 + j) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((slot >> 1) & 7);
    const int memchunk = (logical & ~3) | ((logical & 1) << 1) | ((logical >> 1) & 1);
    int ar = m0 + (slot >> 6) * 128 + (slot & 63);              // A0 row; A1 = + 64
    int wr = n0 + (slot >> 5) * 64 + (slot & 31);               // W0 row; W1 = + 32
    const int ar1 = ar + 64 < p.m ? ar + 64 : p.m - 1, wr1 = wr + 32 < p.n ? wr + 32 : p.n - 1;
    ar = ar < p.m ? ar : p.m - 1;
    wr = wr < p.n ? wr : p.n - 1;
    va0[j] = (unsigned)((size_t)ar * p.lda * 4 + memchunk * 16);
    va1[j] = (unsigned)((size_t)ar1 * p.lda * 4 + memchunk * 16);
    vw0[j] = (unsigned)((size_t)wr * k16 * 4 + memchunk * 16);
    vw1[j] = (unsigned)((size_t)wr1 * k16 * 4 + memchunk * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const unsigned dma_dst = lds0 + wave * 2048;                   // + buffer + region + j * 1024
  // request region `reg` of K tile kt into buffer kt & 1
  auto stage = [&](int kt, int reg, unsigned (&voff)[2], const unsigned char* base) {
    const unsigned dst = dma_dst + (kt & 1) * P8_TILE + reg;
    const unsigned char* kbase = base + (size_t)kt * ROWB;       // scalar: the K offset rides on the SGPR base
    p8_dma(dst, voff[0], kbase);
    p8_dma(dst + 1024, voff[1], kbase);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- fragment addresses (byte offsets inside a region): slot * 128 + ((logical ^ f(slot)) << 4)
  const int hsel = lane >> 5;
  int aoff[2][2][2], woff[2][2];                                  // [row block][k16 step][hi / lo], [k16 step][hi / lo]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = group * 64 + i * 32 + (lane & 31), f = (slot >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aoff[i][ks][0] = slot * ROWB + (((ks * 4 + hsel) ^ f) << 4);
      aoff[i][ks][1] = slot * ROWB + (((ks * 4 + 2 + hsel) ^ f) << 4);
    }
  }
  {
    const int slot = wn * 32 + (lane & 31), f = (slot >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      woff[ks][0] = slot * ROWB + (((ks * 4 + hsel) ^ f) << 4);
      woff[ks][1] = slot * ROWB + (((ks * 4 + 2 + hsel) ^ f) << 4);
    }
  }
  h8 xh[2][2], xl[2][2], w0h[2], w0l[2], w1h[2], w1l[2];         // activation fragments [row block][k16 step]; weights [k16 step]
  auto read_a = [&](const unsigned char* region) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        xh[i][ks] = *reinterpret_cast<const h8*>(region + aoff[i][ks][0]);
        xl[i][ks] = *reinterpret_cast<const h8*>(region + aoff[i][ks][1]);
      }
  };
  auto read_w = [&](const unsigned char* region, h8 (&wh)[2], h8 (&wl)[2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      wh[ks] = *reinterpret_cast<const h8*>(region + woff[ks][0]);
      wl[ks] = *reinterpret_cast<const h8*>(region + woff[ks][1]);
    }
  };
  // 12 MFMAs of one quadrant: accumulators acc[2 * half + i][j], i = 0, 1
  auto quadrant = [&](f32x16& c0, f32x16& c1, const h8 (&wh)[2], const h8 (&wl)[2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], xh[0][ks], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[ks], xh[1][ks], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xl[0][ks], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xl[1][ks], c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xh[0][ks], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[ks], xh[1][ks], c1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: K tiles 0 and 1 complete
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
    if (kt < nk) {
      stage(kt, P8_A0, va0, abase);
      stage(kt, P8_W0, vw0, wsp);
      stage(kt, P8_W1, vw1, wsp);
      stage(kt, P8_A1, va1, abase);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P8_BARRIER();
  if (group == 1) P8_BARRIER();                                  // the second row group runs one barrier behind

  // One K tile = four phases.  Counted waits: in steady state the four newest region requests (8 DMA instructions
  // of this wave) stay in flight; the last two K tiles request less, so less may be left outstanding
  // (TAIL 1 = tile nk - 2, TAIL 2 = tile nk - 1).  S12 / S34: whether phases 1-2 / 3-4 request a region (tile 0 does
  // not: the prologue already did; the last tiles have nothing left to request) — compile-time, so the loop
  // body is branch-free.
#define P8_WAIT(a, b, c)                                                                       \
  do {                                                                                         \
    if constexpr (TAIL == 0) asm volatile("s_waitcnt vmcnt(" #a ")" ::: "memory");             \
    else if constexpr (TAIL == 1) asm volatile("s_waitcnt vmcnt(" #b ")" ::: "memory");        \
    else asm volatile("s_waitcnt vmcnt(" #c ")" ::: "memory");                                 \
  } while (0)
  auto ktile = [&](int kt, auto tail_c, auto s12_c, auto s34_c) {
    constexpr int TAIL = decltype(tail_c)::value;
    constexpr bool S12 = decltype(s12_c)::value, S34 = decltype(s34_c)::value;
    const unsigned char* buf = smem_raw + (kt & 1) * P8_TILE;
    // phase 1
    read_a(buf + P8_A0);
    read_w(buf + P8_W0, w0h, w0l);
    if constexpr (S12) stage(kt + 1, P8_W1, vw1, wsp);
    P8_WAIT(8, 8, 2);
    P8_BARRIER();
    quadrant(acc[0][0], acc[1][0], w0h, w0l);
    P8_BARRIER();
    // phase 2
    read_w(buf + P8_W1, w1h, w1l);
    if constexpr (S12) stage(kt + 1, P8_A1, va1, abase);
    P8_WAIT(8, 8, 0);
    P8_BARRIER();
    quadrant(acc[0][1], acc[1][1], w1h, w1l);
    P8_BARRIER();
    // phase 3
    read_a(buf + P8_A1);
    if constexpr (S34) stage(kt + 2, P8_A0, va0, abase);
    P8_WAIT(8, 6, 0);
    P8_BARRIER();
    quadrant(acc[2][1], acc[3][1], w1h, w1l);
    P8_BARRIER();
    // phase 4
    if constexpr (S34) stage(kt + 2, P8_W0, vw0, wsp);
    P8_WAIT(8, 4, 0);
    P8_BARRIER();
    quadrant(acc[2][0], acc[3][0], w0h, w0l);
    P8_BARRIER();
  };
#undef P8_WAIT
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  if (nk >= 3) {
    ktile(0, I0{}, std::false_type{}, std::true_type{});
    for (int kt = 1; kt + 2 < nk; ++kt) ktile(kt, I0{}, std::true_type{}, std::true_type{});
    ktile(nk - 2, I1{}, std::true_type{}, std::false_type{});
    ktile(nk - 1, I2{}, std::false_type{}, std::false_type{});
  } else {                                                       // nk == 2: everything came with the prologue
    ktile(0, I1{}, std::false_type{}, std::false_type{});
    ktile(1, I2{}, std::false_type{}, std::false_type{});
  }
  if (group == 0) P8_BARRIER();                                  // pairs with the extra barrier of group 1
  P8_BARRIER();                                                  // every read of the operand tiles is over: LDS becomes epilogue patches

  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  const int mw = m0 + group * 128, nw = n0 + wn * 64;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  static_assert(P8_LDS >= 8 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
  if (VAR & SVAR_CSPLIT) {
    switch (p.act) {
      case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane, acc, patch); break;
      default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane, acc, patch); break;
    }
  } else {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue_lds<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_SILU: split_epilogue_lds<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane, acc, patch); break;
      case WD_ACT_GELU: split_epilogue_lds<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane, acc, patch); break;
      default: split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane, acc, patch); break;
    }
  }
}

template <int VAR>
int launch_p8(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st) {
  const int nbm = (p.m + 255) / 256, nbn = (p.n + 255) / 256;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  // 32-bit DMA offsets from the operand bases
  if ((unsigned long long)p.m * p.lda * 4 >= (1ull << 32) || (unsigned long long)p.n * k16 * 4 >= (1ull << 32))
    return WD_ERR_UNSUPPORTED;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c);
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  int ngrp = 2;                                                   // column tiles walked in pairs: 512 weight rows live per group
  if (ngrp > nbn || nbn % ngrp) ngrp = nbn;
  auto k = split_gemm_p8_kernel<VAR>;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS) !=
        hipSuccess) return WD_ERR_LAUNCH;
    attr = true;
  }
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(512), P8_LDS, st, p, static_cast<const unsigned char*>(wsp), k16, unscale, nbn,
                 vec_c, vec_res, vec_bias, ngrp, nbm);
  return wd_launch_status();
}

}  // namespace

// cfg 64: plain 1x1 layer, both operands pre-split, K % 32 == 0.  csplit: output written as fp16 hi/lo groups.
int wd_launch_p8(const WdConvGemm& p, const void* w, float unscale, bool csplit, hipStream_t st) {
  if (p.k % 32 || p.k < 64 || p.lda % 8 || !wd_aligned16(p.a) || !wd_aligned16(w)) return WD_ERR_UNSUPPORTED;
  if (csplit) {
    if (p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c)) return WD_ERR_BAD_ARG;
    return launch_p8<SVAR_CSPLIT>(p, w, unscale, st);
  }
  return launch_p8<0>(p, w, unscale, st);
}
