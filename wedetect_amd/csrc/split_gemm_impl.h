// split_gemm_impl.h — kernel template of the fp16x3 GEMM (see split_gemm.hip for the description);
// included by split_gemm.hip (activations split by the loader) and split_gemm_pre.hip (operands
// already stored as fp16 hi/lo groups by their producers).
#pragma once
#include <type_traits>
#include "common.h"
#include "gemm_loader.h"

namespace {

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int SVAR_XCD = 256;      // XCD-aware tile order (as in conv_gemm.hip)
constexpr int SVAR_PIN = 2;        // sched_barrier fences: global loads, then MFMAs, then split + LDS store
constexpr int SVAR_PF2 = 128;      // global loads run two K stages ahead (two register sets)
constexpr int SVAR_LDSEPI = 512;   // epilogue through LDS: every global access of C / residual / bias is a full 128-byte row segment
constexpr int SVAR_ASPLIT = 1024;  // the activation operand is already stored as fp16 hi/lo groups (same layout as split weights)
constexpr int SVAR_CSPLIT = 2048;  // the output is written as fp16 hi/lo groups for a consuming ASPLIT layer (needs SVAR_LDSEPI)
// timing-only ablations (WRONG results by construction) for on-device diagnosis; no shipped configuration sets them
constexpr int SABL_NOLOAD = 4, SABL_NOBAR = 8, SABL_NOEPI = 16, SABL_NOLDS = 32, SABL_NOSPLIT = 64;

template <int TM, int TN, int WM, int WN, int BK_>
struct STile {
  static constexpr int BK = BK_;                          // k per LDS stage: 16 or 32
  static constexpr int KS = BK / 16;                      // MFMA k-steps per stage
  static constexpr int ROWB = BK * 4 + 16;                // bytes per LDS row
  static constexpr int KCH = BK / 4;                      // 16-byte global chunks per row per stage
  static constexpr int BM = 32 * TM * WM;
  static constexpr int BN = 32 * TN * WN;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int A_PT = (BM * KCH) / NT;
  static constexpr int B_PT = (BN * KCH) / NT;
  static constexpr int RSTEP = NT / KCH;
  static constexpr int LDS_BYTES = 2 * (BM + BN) * ROWB;
  static_assert(BK == 16 || BK == 32, "stage depth");
  static_assert((BM * KCH) % NT == 0 && (BN * KCH) % NT == 0, "tiles must split evenly over the threads");
};

// x -> (hi, lo) halves of 4 consecutive k, packed 2 per dword
__device__ __forceinline__ void split4(const f32x4 v, u32x2& hi, u32x2& lo) {
  // no contraction: the residual must be taken from the ROUNDED fp32 value (fusing a producer's
  // final multiply into this subtraction would split a different number than the one an fp32
  // buffer would have held, and pre-split layers would stop being bit-identical to the others)
#pragma clang fp contract(off)
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  hi = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  lo = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

// ---------------------------------------------------------------------------------------
// epilogue pieces (same semantics as conv_gemm.hip's epilogue; one accumulator quad = 4
// consecutive channels n..n+3 of pixel row m)
// ---------------------------------------------------------------------------------------
struct EpiRow { size_t crow; int hw2; float oscale, obias; };

template <bool SPECIAL>
__device__ __forceinline__ EpiRow epi_row(const WdConvGemm& p, int m) {
  EpiRow er{(size_t)m, 0, 1.0f, 0.0f};
  if (SPECIAL) {
    if (p.out_mode == WD_OUT_DECONV2X2) {
      const int wq = m % p.wout;
      const int q = m / p.wout;
      const int hq = q % p.hout;
      const int b = q / p.hout;
      er.crow = ((size_t)(b * 2 * p.hout + 2 * hq) * (2 * p.wout) + 2 * wq);
      er.hw2 = 2 * p.wout;
    } else if (p.c_batch_stride > 0) {
      const int hw = p.hout * p.wout;
      const int b = m / hw;
      er.crow = (size_t)b * p.c_batch_stride + (size_t)(m - b * hw);
    }
    er.oscale = p.out_scale; er.obias = p.out_bias;
    if (p.seg_rows > 0) {
      const int pos = m % p.seg_rows;
      const int lvl = (pos >= p.seg_end0) + (pos >= p.seg_end1);
      er.oscale = lvl == 0 ? p.seg_scale[0] : lvl == 1 ? p.seg_scale[1] : p.seg_scale[2];
      er.obias = lvl == 0 ? p.seg_bias[0] : lvl == 1 ? p.seg_bias[1] : p.seg_bias[2];
    }
  }
  return er;
}

template <int ACT>
__device__ __forceinline__ float sact(float v) {
  if (ACT == WD_ACT_RELU) return fmaxf(v, 0.0f);
  if (ACT == WD_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
  if (ACT == WD_ACT_GELU) return wd_gelu(v);
  return v;
}

struct EpiVec { int c, res, bias; float unscale; };

template <int ACT, bool SPECIAL>
__device__ __forceinline__ void epi_quad(const WdConvGemm& p, const EpiRow& er, const EpiVec& ev, int m, int n,
                                         const f32x4 v) {
  if (n >= p.n) return;
  if (p.range_flag) {          // an fp16 operand half that overflowed to inf shows up as an inf / NaN accumulator
    if (wd_any_nonfinite4(v[0], v[1], v[2], v[3])) *p.range_flag = 1u;
  }
  const bool full = n + 3 < p.n;
  f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    if (full && ev.bias) {
      b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n + r < p.n) b4[r] = p.bias[n + r];
    }
  }
  float o[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x = sact<ACT>(fmaf(v[r], ev.unscale, b4[r]));     // unscale is a power of two: exact
    if (SPECIAL) {
      x = fmaf(x, er.oscale, er.obias);           // explicit fma: the same bits in every kernel's epilogue
      if (p.sigmoid) x = wd_sigmoid_fast(x);
    }
    o[r] = x;
  }
  if (p.res != nullptr) {
    const float* rp = p.res + (size_t)m * p.ldres + n;
    if (full && ev.res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = fmaf(p.res_alpha, rv[r], o[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n + r < p.n) o[r] = fmaf(p.res_alpha, rp[r], o[r]);
    }
  }
  float* cp;
  if (SPECIAL && p.out_mode == WD_OUT_DECONV2X2) {
    const int ncq = p.n >> 2;
    const int tap = n / ncq, co = n - tap * ncq;
    cp = p.c + (er.crow + (size_t)(tap >> 1) * er.hw2 + (tap & 1)) * p.ldc + co;
  } else {
    cp = p.c + er.crow * p.ldc + n;
  }
  if (full && ev.c) {
    *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) if (n + r < p.n) cp[r] = o[r];
  }
}

// accumulator tile (32 pixels x 32 channels): lane holds pixel (lane & 31), channels
// 8g + 4(lane >> 5) + 0..3 in registers 4g..4g+3
template <int TM, int TN, int ACT, bool SPECIAL>
__device__ __forceinline__ void split_epilogue(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                               const f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = mw + i * 32 + (lane & 31);
    if (m >= p.m) continue;
    const EpiRow er = epi_row<SPECIAL>(p, m);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw + j * 32 + 8 * g + 4 * (lane >> 5);
        const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        epi_quad<ACT, SPECIAL>(p, er, ev, m, n, v);
      }
    }
  }
}

// Same epilogue through LDS: each wave transposes its 32 x 32 accumulator tiles in a private
// 32 x 36-float patch, so that a lane ends with 4 consecutive channels of a row and 8 lanes cover
// a full 128-byte row segment: C stores, residual and bias loads are whole cache lines (the
// direct epilogue writes 32-byte pieces of 32 different rows per instruction).
constexpr int EPI_LDT = 36;

template <int I, int J, int TM, int TN, int ACT, bool SPECIAL>
__device__ __forceinline__ void epi_lds_tile(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN], float* patch) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(patch + (lane & 31) * EPI_LDT + 8 * g + 4 * (lane >> 5)) =
        f32x4{acc[I][J][4 * g], acc[I][J][4 * g + 1], acc[I][J][4 * g + 2], acc[I][J][4 * g + 3]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 8 + (lane >> 3);
    const f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * EPI_LDT + 4 * (lane & 7));
    const int m = mw + I * 32 + row;
    const int n = nw + J * 32 + 4 * (lane & 7);
    if (m < p.m) {
      const EpiRow er = epi_row<SPECIAL>(p, m);
      epi_quad<ACT, SPECIAL>(p, er, ev, m, n, v);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The same tile written as fp16 hi/lo groups ([hi x8 | lo x8] per 8 channels, the layout an
// ASPLIT consumer and wd_split_weights use): a lane takes 8 consecutive channels of a row, 4 lanes
// cover the 128 bytes the fp32 row segment would occupy.  Plain row-major outputs only.
// EABL (timing-only, -DWD_DEBUG_ABLATIONS builds): 32 = all the arithmetic, no stores; 64 = the stores, no arithmetic
template <int I, int J, int TM, int TN, int ACT, int EABL = 0>
__device__ __forceinline__ void epi_lds_tile_csplit(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                                    const f32x16 (&acc)[TM][TN], float* patch) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<f32x4*>(patch + (lane & 31) * EPI_LDT + 8 * g + 4 * (lane >> 5)) =
        f32x4{acc[I][J][4 * g], acc[I][J][4 * g + 1], acc[I][J][4 * g + 2], acc[I][J][4 * g + 3]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = ps * 16 + (lane >> 2);
    const int m = mw + I * 32 + row;
    const int n = nw + J * 32 + 8 * (lane & 3);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + row * EPI_LDT + 8 * (lane & 3));
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(patch + row * EPI_LDT + 8 * (lane & 3) + 4);
    if (m < p.m && n < p.n) {
      if (p.range_flag) {
        if (wd_any_nonfinite4(v0[0] + v0[1], v0[2] + v0[3], v1[0] + v1[1], v1[2] + v1[3])) *p.range_flag = 1u;
      }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
        b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
      }
      f32x4 o0, o1;
      if constexpr ((EABL & 64) != 0) {
        const u32x4 r0 = __builtin_bit_cast(u32x4, v0), r1 = __builtin_bit_cast(u32x4, v1);
        unsigned char* cq = reinterpret_cast<unsigned char*>(p.c + (size_t)m * p.ldc) + (size_t)(n >> 3) * 32;
        __builtin_nontemporal_store(r0, reinterpret_cast<u32x4*>(cq));
        __builtin_nontemporal_store(r1, reinterpret_cast<u32x4*>(cq + 16));
        continue;
      }
      if (p.ln_stats) {                                            // LayerNorm folded into this GEMM (WdConvGemm.ln_stats): wave-uniform
        const f32x2 st = *reinterpret_cast<const f32x2*>(p.ln_stats + 2 * (size_t)m);
        const f32x4 u0 = *reinterpret_cast<const f32x4*>(p.ln_u + n), u1 = *reinterpret_cast<const f32x4*>(p.ln_u + n + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[r] = sact<ACT>(fmaf(st[1], fmaf(-st[0], u0[r], v0[r] * ev.unscale), b0[r]));
          o1[r] = sact<ACT>(fmaf(st[1], fmaf(-st[0], u1[r], v1[r] * ev.unscale), b1[r]));
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o0[r] = sact<ACT>(fmaf(v0[r], ev.unscale, b0[r]));
          o1[r] = sact<ACT>(fmaf(v1[r], ev.unscale, b1[r]));
        }
      }
      u32x2 h0, l0, h1, l1;
      // c_split_scale: a power of two (exact); 0 = none.  Always multiplied — x * 1.0f is x, bit for bit, so the unscaled path
      // stays identical to the loader-split kernels — because hipcc turned the "only when it is not 1" branch of rounds 1-3
      // into a multiply PLUS a v_cndmask per value (one extra VALU instruction in 23 of this epilogue)
      const float cs = p.c_split_scale != 0.f ? p.c_split_scale : 1.0f;
      o0 = o0 * cs;
      o1 = o1 * cs;
      split4(o0, h0, l0);
      split4(o1, h1, l1);
      unsigned char* cp = reinterpret_cast<unsigned char*>(p.c + (size_t)m * p.ldc) + (size_t)(n >> 3) * 32;
      if constexpr ((EABL & 32) != 0) {                       // keep the arithmetic alive, store (almost) nothing
        if ((h0[0] ^ l0[0] ^ h1[1] ^ l1[1]) == 0x12345678u) *reinterpret_cast<u32x4*>(cp) = u32x4{h0[0], h0[1], h1[0], h1[1]};
        continue;
      }
      // non-temporal: the hi/lo output of a C-split layer is a write-once stream (1.7 GB per stage-1 pwconv1) that its consumer
      // re-reads only after this launch is over: keep it from displacing the operand panels in L2 (-0.3 ms per step, same box)
#ifdef WD_CSPLIT_PLAIN_STORE
      *reinterpret_cast<u32x4*>(cp) = u32x4{h0[0], h0[1], h1[0], h1[1]};
      *reinterpret_cast<u32x4*>(cp + 16) = u32x4{l0[0], l0[1], l1[0], l1[1]};
#else
      __builtin_nontemporal_store(u32x4{h0[0], h0[1], h1[0], h1[1]}, reinterpret_cast<u32x4*>(cp));
      __builtin_nontemporal_store(u32x4{l0[0], l0[1], l1[0], l1[1]}, reinterpret_cast<u32x4*>(cp + 16));
#endif
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Round 5 experiment ($WEDETECT_CSPLIT_DIRECT=1, 256 x 256 kernel only): the same hi/lo output WITHOUT the LDS transpose.  In the
// accumulator a lane (row m, half h) holds channels 8 g + 4 h + 0..3 of every 8-channel group g, its partner lane (m, 1 - h) the
// other four: two v_permlane32_swap per group hand the low half the eight hi values and the high half the eight lo values —
// 16 bytes each, adjacent in memory — and a group is stored with one instruction (32 rows x 32 contiguous bytes).  Same
// arithmetic per element as epi_lds_tile_csplit: bit-identical.  The through-LDS form exists because whole 128-byte row segments
// were worth 13-27 % on the round-1 kernels; round 5 measured that full-line fp32 rows and 16-byte pieces at a 32-byte stride cost
// this kernel the same (profiles/r05_gelu_ab.txt), so the transpose (4 ds_write_b128 + 4 ds_read_b128 + two wave barriers per
// 32 x 32 tile) might have been pure cost.  It is not: MEASURED 516 vs 336 us on the stage-3 pwconv1 launch, 2 088 vs 695 us at stage 1,
// the step 782 vs 881 images/s (profiles/r05_csplit_direct.txt) — a store instruction that touches 32 lines with 32 bytes each costs
// about three times one that touches 16 with 64; from 16 lines to 8 whole ones nothing more is gained.  Reachable in
// -DWD_DEBUG_ABLATIONS builds only ($WEDETECT_CSPLIT_DIRECT=1).  (The round-2 "compute-first" form did the swaps AND the transpose.)
template <int I, int J, int TM, int TN, int ACT>
__device__ __forceinline__ void epi_direct_tile_csplit(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                                       const f32x16 (&acc)[TM][TN]) {
  const int m = mw + I * 32 + (lane & 31);
  const int h = lane >> 5;
  const float cs = p.c_split_scale != 0.f ? p.c_split_scale : 1.0f;
  unsigned char* rowp = reinterpret_cast<unsigned char*>(p.c + (size_t)(m < p.m ? m : 0) * p.ldc) + 16 * h;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n = nw + J * 32 + 8 * g;                              // wave-uniform: the swaps below need every lane
    if (n >= p.n) continue;
    const f32x4 v = {acc[I][J][4 * g], acc[I][J][4 * g + 1], acc[I][J][4 * g + 2], acc[I][J][4 * g + 3]};
    if (p.range_flag) {
      if (wd_any_nonfinite4(v[0], v[1], v[2], v[3])) *p.range_flag = 1u;
    }
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b = *reinterpret_cast<const f32x4*>(p.bias + n + 4 * h);
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = sact<ACT>(fmaf(v[r], ev.unscale, b[r]));
    o = o * cs;
    u32x2 hh, ll;
    split4(o, hh, ll);
    u32x4 out;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto sw = __builtin_amdgcn_permlane32_swap(hh[d], ll[d], false, false);   // low half: [hi own | hi partner]; high half: [lo partner | lo own]
      out[d] = sw[0];
      out[2 + d] = sw[1];
    }
    if (m < p.m) __builtin_nontemporal_store(out, reinterpret_cast<u32x4*>(rowp + (size_t)(n >> 3) * 32));
  }
}
template <int IJ, int TM, int TN, int ACT>
struct EpiCsplitDirectWalk {
  static __device__ __forceinline__ void run(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN]) {
    epi_direct_tile_csplit<IJ / TN, IJ % TN, TM, TN, ACT>(p, ev, mw, nw, lane, acc);
    EpiCsplitDirectWalk<IJ + 1, TM, TN, ACT>::run(p, ev, mw, nw, lane, acc);
  }
};
template <int TM, int TN, int ACT>
struct EpiCsplitDirectWalk<TM * TN, TM, TN, ACT> {
  static __device__ __forceinline__ void run(const WdConvGemm&, const EpiVec&, int, int, int, const f32x16 (&)[TM][TN]) {}
};

// Tried and dropped (round 2, profiles/r02_p8_ablations.txt): a compute-first form of this epilogue — bias + GELU +
// fp16 split on the accumulators where they lie, 8-byte pieces swapped between the lane halves with
// v_permlane32_swap so that every lane owns a final 16-byte [hi x 8] / [lo x 8] chunk, and only then the LDS
// transpose — was 5-10 % SLOWER on the 51200 x 2048 x 512 pwconv1 launch (450-470 us against 404-426 us): beside 128
// live accumulators the 16-value GELU batches spill, and the transposed form already overlaps its VALU work with the
// stores of the previous 32 x 32 tile.
template <int IJ, int TM, int TN, int ACT, int EABL = 0>
struct EpiCsplitWalk {
  static __device__ __forceinline__ void run(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN], float* patch) {
    epi_lds_tile_csplit<IJ / TN, IJ % TN, TM, TN, ACT, EABL>(p, ev, mw, nw, lane, acc, patch);
    EpiCsplitWalk<IJ + 1, TM, TN, ACT, EABL>::run(p, ev, mw, nw, lane, acc, patch);
  }
};
template <int TM, int TN, int ACT, int EABL>
struct EpiCsplitWalk<TM * TN, TM, TN, ACT, EABL> {
  static __device__ __forceinline__ void run(const WdConvGemm&, const EpiVec&, int, int, int,
                                             const f32x16 (&)[TM][TN], float*) {}
};

// compile-time walk over the TM x TN accumulator tiles (a #pragma unroll loop around the fences is
// not always honoured, and a rolled loop would index the accumulators through scratch)
template <int IJ, int TM, int TN, int ACT, bool SPECIAL>
struct EpiLdsWalk {
  static __device__ __forceinline__ void run(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN], float* patch) {
    epi_lds_tile<IJ / TN, IJ % TN, TM, TN, ACT, SPECIAL>(p, ev, mw, nw, lane, acc, patch);
    EpiLdsWalk<IJ + 1, TM, TN, ACT, SPECIAL>::run(p, ev, mw, nw, lane, acc, patch);
  }
};
template <int TM, int TN, int ACT, bool SPECIAL>
struct EpiLdsWalk<TM * TN, TM, TN, ACT, SPECIAL> {
  static __device__ __forceinline__ void run(const WdConvGemm&, const EpiVec&, int, int, int,
                                             const f32x16 (&)[TM][TN], float*) {}
};

// Residual epilogue with the residual rows PREFETCHED (plain row-major fp32 output, no activation, vector-aligned C /
// residual / bias, whole column tiles — the ConvNeXt pwconv2 case).  The walk above loads a tile's residual quads,
// waits, adds, stores, and only then starts the next tile: with the output written in place over the residual the
// loads of tile b + 1 sit behind the stores of tile b, eight dependent HBM round trips per wave, and the epilogue
// of a 256 x 256 tile is latency-bound (a quarter of the pwconv2 launch).  Here the residual quads of the next D
// accumulator tiles are already in flight while one is transposed and stored.  Same arithmetic, same order per
// element as epi_quad: bit-identical output.
template <int IJ, int TM, int TN>
__device__ __forceinline__ void epi_res_load(const WdConvGemm& p, int mw, int nw, int lane, f32x4 (&rv)[4]) {
  constexpr int I = IJ / TN, J = IJ % TN;
  const int n = nw + J * 32 + 4 * (lane & 7);
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    int m = mw + I * 32 + ps * 8 + (lane >> 3);
    m = m < p.m ? m : p.m - 1;                                   // rows past the end: any valid address, never stored
    rv[ps] = *reinterpret_cast<const f32x4*>(p.res + (size_t)m * p.ldres + n);
  }
}

template <int IJ, int TM, int TN, int D>
struct EpiResWalk {
  static __device__ __forceinline__ void run(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN], float* patch, f32x4 (&rv)[D][4]) {
    constexpr int I = IJ / TN, J = IJ % TN;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(patch + (lane & 31) * EPI_LDT + 8 * g + 4 * (lane >> 5)) =
          f32x4{acc[I][J][4 * g], acc[I][J][4 * g + 1], acc[I][J][4 * g + 2], acc[I][J][4 * g + 3]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x4 v[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps)
      v[ps] = *reinterpret_cast<const f32x4*>(patch + (ps * 8 + (lane >> 3)) * EPI_LDT + 4 * (lane & 7));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int n = nw + J * 32 + 4 * (lane & 7);
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int m = mw + I * 32 + ps * 8 + (lane >> 3);
      if (p.range_flag) {
        if (wd_any_nonfinite4(v[ps][0], v[ps][1], v[ps][2], v[ps][3])) *p.range_flag = 1u;
      }
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = fmaf(v[ps][r], ev.unscale, b4[r]);
        o[r] = fmaf(p.res_alpha, rv[IJ % D][ps][r], o[r]);
      }
      if (m < p.m) *reinterpret_cast<f32x4*>(p.c + (size_t)m * p.ldc + n) = f32x4{o[0], o[1], o[2], o[3]};
    }
    if constexpr (IJ + D < TM * TN) epi_res_load<IJ + D, TM, TN>(p, mw, nw, lane, rv[IJ % D]);
    EpiResWalk<IJ + 1, TM, TN, D>::run(p, ev, mw, nw, lane, acc, patch, rv);
  }
};
template <int TM, int TN, int D>
struct EpiResWalk<TM * TN, TM, TN, D> {
  static __device__ __forceinline__ void run(const WdConvGemm&, const EpiVec&, int, int, int, const f32x16 (&)[TM][TN],
                                             float*, f32x4 (&)[D][4]) {}
};
template <int IJ, int TM, int TN, int D>
struct EpiResPrime {
  static __device__ __forceinline__ void run(const WdConvGemm& p, int mw, int nw, int lane, f32x4 (&rv)[D][4]) {
    epi_res_load<IJ, TM, TN>(p, mw, nw, lane, rv[IJ]);
    EpiResPrime<IJ + 1, TM, TN, D>::run(p, mw, nw, lane, rv);
  }
};
template <int TM, int TN, int D>
struct EpiResPrime<D, TM, TN, D> {
  static __device__ __forceinline__ void run(const WdConvGemm&, int, int, int, f32x4 (&)[D][4]) {}
};

// the conditions under which the prefetching residual epilogue applies to a wave's tile (all wave-uniform)
__device__ __forceinline__ bool epi_res_prefetch_ok(const WdConvGemm& p, const EpiVec& ev, int nw, int width) {
  return p.res != nullptr && p.act == WD_ACT_NONE && ev.res && ev.c && (!p.bias || ev.bias) && nw + width <= p.n && (p.n & 3) == 0 &&
         p.out_mode == WD_OUT_ROWS;
}

template <int TM, int TN, int D>
__device__ __forceinline__ void split_epilogue_res_prefetch(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                                            const f32x16 (&acc)[TM][TN], float* patch) {
  f32x4 rv[D][4];
  EpiResPrime<0, TM, TN, D>::run(p, mw, nw, lane, rv);
  EpiResWalk<0, TM, TN, D>::run(p, ev, mw, nw, lane, acc, patch, rv);
}

template <int TM, int TN, int ACT, bool SPECIAL>
__device__ __forceinline__ void split_epilogue_lds(const WdConvGemm& p, const EpiVec& ev, int mw, int nw, int lane,
                                                   const f32x16 (&acc)[TM][TN], float* patch) {
  EpiLdsWalk<0, TM, TN, ACT, SPECIAL>::run(p, ev, mw, nw, lane, acc, patch);
}

constexpr int split_waves_per_simd(int waves, int lds_bytes, int acc_tiles, bool two_sets) {
  // resident workgroups are LDS-limited (160 KB per CU); ask for the register budget that fits
  // them, but never squeeze a 128-register accumulator set (64 x 128 wave tiles) below 256 registers
  const int wgs = (160 * 1024) / lds_bytes;
  const int w = (wgs * waves) / 4;
  // a second staging register set (prefetch distance 2) does not fit 128 registers: trade one
  // resident wave per SIMD for it
  const int cap = acc_tiles > 4 ? 2 : (two_sets ? 3 : 4);
  return w > cap ? cap : (w < 1 ? 1 : w);
}

template <int TM, int TN, int WM, int WN, int BKT, bool CONV, int VAR>
__global__ void __launch_bounds__(64 * WM * WN, split_waves_per_simd(WM * WN, STile<TM, TN, WM, WN, BKT>::LDS_BYTES, TM * TN, (VAR & SVAR_PF2) != 0))
split_gemm_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, const float* __restrict__ zero,
                  int k16, float unscale, int nbn, int vec_c, int vec_res, int vec_bias, int ksplits, float* __restrict__ ws) {
  using T = STile<TM, TN, WM, WN, BKT>;
  constexpr int BK = T::BK, KS = T::KS, ROWB = T::ROWB, KCH = T::KCH, BM = T::BM, BN = T::BN;
  constexpr int A_PT = T::A_PT, B_PT = T::B_PT, RSTEP = T::RSTEP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned char* As = smem_raw;                         // activations: [2][BM][ROWB]
  unsigned char* Bs = smem_raw + 2 * BM * ROWB;         // weights:     [2][BN][ROWB]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile = blockIdx.x;
  if (VAR & SVAR_XCD) {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // split-K (ksplits > 1): workgroup (tile, ks) accumulates K stages [s_begin, s_end) and writes raw partial
  // sums to ws[ks][m][n]; splitk_reduce_kernel adds them in a fixed order and applies the epilogue
  const int ntiles = gridDim.x / ksplits;
  const int ks = tile / ntiles;
  tile -= ks * ntiles;
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kc = t % KCH, r0 = t / KCH;
  const int K = p.k;
  const float a_scale = p.a_scale != 0.f ? p.a_scale : 1.0f;
  const int nk_all = (K + BK - 1) / BK;
  const int per_split = (nk_all + ksplits - 1) / ksplits;
  const int s_begin = ks * per_split;
  const int nk = (s_begin + per_split < nk_all ? s_begin + per_split : nk_all) - s_begin;   // stages of this workgroup (may be <= 0)
  const int kofs = s_begin * BK;

  ALoader<A_PT, RSTEP, CONV> al;
  al.init(p, m0, r0, zero);

  // weight chunk kc of a stage: 8-k group g = kc / 2, part = kc & 1 (0 = hi, 1 = lo)
  const unsigned char* wrow[B_PT];
  bool wok[B_PT];
#pragma unroll
  for (int j = 0; j < B_PT; ++j) {
    const int n = n0 + r0 + j * RSTEP;
    wok[j] = n < p.n;
    wrow[j] = wsp + (size_t)(wok[j] ? n : 0) * k16 * 4 + kc * 16;
  }
  // LDS store offsets inside a row
  const int a_off = (kc >> 2) * 64 + (kc & 3) * 8;                              // hi; lo at +32
  const int b_off = (kc >> 2) * 64 + (kc & 1) * 32 + ((kc >> 1) & 1) * 16;

  constexpr int NSET = (VAR & SVAR_PF2) ? 2 : 1;
  f32x4 areg[NSET][A_PT];
  u32x4 breg[NSET][B_PT];
  // stage at kbase -> register set S (k beyond K reads the zero block: over-running loads are harmless)
  auto load = [&](int kbase, auto set) {
    constexpr int S = decltype(set)::value;
    al.template load<BK>(kbase, kc * 4, areg[S]);
    const bool kok = kbase + (kc >> 1) * 8 < k16;
#pragma unroll
    for (int j = 0; j < B_PT; ++j)
      breg[S][j] = *reinterpret_cast<const u32x4*>((wok[j] && kok) ? wrow[j] + (size_t)kbase * 4
                                                                   : reinterpret_cast<const unsigned char*>(zero));
  };
  auto store = [&](int buf, auto set) {
    constexpr int S = decltype(set)::value;
    if (VAR & SVAR_ASPLIT) {          // chunks are [hi x8 | lo x8] groups already: plain copies, like the weights
      unsigned char* ad = As + (buf * BM + r0) * ROWB + b_off;
#pragma unroll
      for (int i = 0; i < A_PT; ++i)
        *reinterpret_cast<f32x4*>(ad + i * RSTEP * ROWB) = areg[S][i];
    }
    unsigned char* ad = As + (buf * BM + r0) * ROWB + a_off;
#pragma unroll
    for (int i = 0; i < ((VAR & SVAR_ASPLIT) ? 0 : A_PT); ++i) {
      u32x2 hi, lo;
      if (VAR & SABL_NOSPLIT) {
        hi = u32x2{__builtin_bit_cast(unsigned, areg[S][i][0]), __builtin_bit_cast(unsigned, areg[S][i][1])};
        lo = u32x2{__builtin_bit_cast(unsigned, areg[S][i][2]), __builtin_bit_cast(unsigned, areg[S][i][3])};
      } else {
        f32x4 av = areg[S][i];
        if (a_scale != 1.0f) av = av * a_scale;    // power of two (exact); wave-uniform branch: scale 1 = the unscaled instruction stream
        split4(av, hi, lo);
      }
      *reinterpret_cast<u32x2*>(ad + i * RSTEP * ROWB) = hi;
      *reinterpret_cast<u32x2*>(ad + i * RSTEP * ROWB + 32) = lo;
    }
    unsigned char* bd = Bs + (buf * BN + r0) * ROWB + b_off;
#pragma unroll
    for (int j = 0; j < B_PT; ++j) *reinterpret_cast<u32x4*>(bd + j * RSTEP * ROWB) = breg[S][j];
  };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  auto compute = [&](int buf) {
    const unsigned char* as = As + (buf * BM + wm * TM * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
    const unsigned char* bs = Bs + (buf * BN + wn * TN * 32 + (lane & 31)) * ROWB + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      h8 xh[TM], xl[TM], wh[TN], wl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        xh[i] = *reinterpret_cast<const h8*>(as + i * 32 * ROWB + ks * 64);
        xl[i] = *reinterpret_cast<const h8*>(as + i * 32 * ROWB + ks * 64 + 32);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        wh[j] = *reinterpret_cast<const h8*>(bs + j * 32 * ROWB + ks * 64);
        wl[j] = *reinterpret_cast<const h8*>(bs + j * 32 * ROWB + ks * 64 + 32);
      }
      // small terms first; consecutive MFMAs on one accumulator are TM*TN instructions apart
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
    }
  };

  load(kofs, Set0{});
  store(0, Set0{});
  __syncthreads();
  if (VAR & SVAR_PF2) {
    // stage s is computed from LDS buffer s & 1 while stage s+1 waits in registers and the loads
    // of stage s+2 are issued: a global load has a whole K stage of MFMAs to land
    load(kofs + BK, Set1{});
    for (int s = 0; s < nk; s += 2) {
      load(kofs + (s + 2) * BK, Set0{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(0);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      store(1, Set1{});
      __syncthreads();
      if (s + 1 >= nk) break;
      load(kofs + (s + 3) * BK, Set1{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(1);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      store(0, Set0{});
      __syncthreads();
    }
  } else {
    int cur = 0;
    for (int kt = 1; kt < nk; ++kt) {
      if (!(VAR & SABL_NOLOAD)) load(kofs + kt * BK, Set0{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      if (!(VAR & SABL_NOLDS)) store(cur ^ 1, Set0{});
      if (!(VAR & SABL_NOBAR)) __syncthreads();
      cur ^= 1;
    }
    if (nk > 0) compute(cur);
  }

  if (VAR & SABL_NOEPI) {            // keep the accumulators live, store one value per lane
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (m0 + (int)(t % BM) < p.m) p.c[(size_t)(m0 + t % BM) * p.ldc + n0] = sacc;
    return;
  }
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
  const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
  const int mw = m0 + wm * TM * 32, nw = n0 + wn * TN * 32;
  if (VAR & SVAR_LDSEPI) {
    static_assert(T::LDS_BYTES >= WM * WN * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
    __syncthreads();                                   // every wave is done reading the last operand stage
    float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
    if (ksplits > 1) {
      WdConvGemm pr = p;                               // raw partial sums, plain rows [m][n] of this split
      pr.bias = nullptr; pr.res = nullptr; pr.c = ws + (size_t)ks * p.m * p.n; pr.ldc = p.n;
      const EpiVec er{(p.n & 3) == 0, 0, 0, 1.0f};
      split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(pr, er, mw, nw, lane, acc, patch);
      return;
    }
    if (VAR & SVAR_CSPLIT) {          // host side guarantees: plain rows, bias 16-byte aligned, n % 8 == 0, no residual
      switch (p.act) {
        case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane, acc, patch); break;
        case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane, acc, patch); break;
        case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane, acc, patch); break;
        default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane, acc, patch); break;
      }
      return;
    }
#define WD_SPLIT_EPI(A, S) split_epilogue_lds<TM, TN, A, S>(p, ev, mw, nw, lane, acc, patch)
    if (special) {
      switch (p.act) {
        case WD_ACT_RELU: WD_SPLIT_EPI(WD_ACT_RELU, true); break;
        case WD_ACT_SILU: WD_SPLIT_EPI(WD_ACT_SILU, true); break;
        case WD_ACT_GELU: WD_SPLIT_EPI(WD_ACT_GELU, true); break;
        default: WD_SPLIT_EPI(WD_ACT_NONE, true); break;
      }
    } else {
      switch (p.act) {
        case WD_ACT_RELU: WD_SPLIT_EPI(WD_ACT_RELU, false); break;
        case WD_ACT_SILU: WD_SPLIT_EPI(WD_ACT_SILU, false); break;
        case WD_ACT_GELU: WD_SPLIT_EPI(WD_ACT_GELU, false); break;
        default: WD_SPLIT_EPI(WD_ACT_NONE, false); break;
      }
    }
#undef WD_SPLIT_EPI
    return;
  }
  if (special) {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue<TM, TN, WD_ACT_RELU, true>(p, ev, mw, nw, lane, acc); break;
      case WD_ACT_SILU: split_epilogue<TM, TN, WD_ACT_SILU, true>(p, ev, mw, nw, lane, acc); break;
      case WD_ACT_GELU: split_epilogue<TM, TN, WD_ACT_GELU, true>(p, ev, mw, nw, lane, acc); break;
      default: split_epilogue<TM, TN, WD_ACT_NONE, true>(p, ev, mw, nw, lane, acc); break;
    }
  } else {
    switch (p.act) {
      case WD_ACT_RELU: split_epilogue<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane, acc); break;
      case WD_ACT_SILU: split_epilogue<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane, acc); break;
      case WD_ACT_GELU: split_epilogue<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane, acc); break;
      default: split_epilogue<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane, acc); break;
    }
  }
}

// WHICH: 0 = instantiate the plain (1x1) and the implicit-im2col kernel, 1 = plain only, 2 = conv only
template <int TM, int TN, int WM, int WN, int BKT, int VAR, int WHICH = 0>
int launch_split(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, int ksplits = 1, float* ws = nullptr) {
  using T = STile<TM, TN, WM, WN, BKT>;
  const bool conv = !(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0);
  if ((WHICH == 1 && conv) || (WHICH == 2 && !conv)) return WD_ERR_UNSUPPORTED;
  const int nbm = (p.m + T::BM - 1) / T::BM, nbn = (p.n + T::BN - 1) / T::BN;
  if (ksplits < 1 || (ksplits > 1 && (!ws || !(VAR & SVAR_LDSEPI) || (VAR & (SVAR_PF2 | SVAR_CSPLIT))))) return WD_ERR_BAD_ARG;
  const long long nblk = (long long)nbm * nbn * ksplits;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c) && (p.out_mode == WD_OUT_ROWS || (p.n % 16 == 0));
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  const unsigned char* w8 = static_cast<const unsigned char*>(wsp);
  const float* zero = wd_zero_block();
  if (!zero) return WD_ERR_LAUNCH;
  static WdAttrOnce attr_plain, attr_conv;
  if constexpr (WHICH != 1) if (conv) {
    auto k = split_gemm_kernel<TM, TN, WM, WN, BKT, true, VAR>;
    if (wd_set_max_lds(attr_conv, reinterpret_cast<const void*>(k), T::LDS_BYTES) != WD_OK) return WD_ERR_LAUNCH;
    WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, w8, zero, k16, unscale, nbn,
                       vec_c, vec_res, vec_bias, ksplits, ws);
    return wd_launch_status();
  }
  if constexpr (WHICH != 2) if (!conv) {
    auto k = split_gemm_kernel<TM, TN, WM, WN, BKT, false, VAR>;
    if (wd_set_max_lds(attr_plain, reinterpret_cast<const void*>(k), T::LDS_BYTES) != WD_OK) return WD_ERR_LAUNCH;
    WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, w8, zero, k16, unscale, nbn,
                       vec_c, vec_res, vec_bias, ksplits, ws);
  }
  return wd_launch_status();
}

int check_split_args(const WdConvGemm& p, const void* wsp, float unscale) {
  if (!p.a || !wsp || !p.c) return WD_ERR_BAD_ARG;
  if (p.m <= 0 || p.n <= 0 || p.k <= 0) return WD_ERR_BAD_ARG;
  if (p.cin <= 0 || p.cin % 4 || p.lda % 4 || p.lda < p.cin) return WD_ERR_BAD_ARG;
  if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0) return WD_ERR_BAD_ARG;
  if (p.k != p.kh * p.kw * p.cin) return WD_ERR_BAD_ARG;
  if ((long long)p.batch * p.hout * p.wout != (long long)p.m) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(p.a) || !wd_aligned16(wsp)) return WD_ERR_BAD_ARG;
  if (p.act < WD_ACT_NONE || p.act > WD_ACT_GELU) return WD_ERR_BAD_ARG;
  if (!(unscale > 0.0f)) return WD_ERR_BAD_ARG;
  if (p.a_scale < 0.f || p.c_split_scale < 0.f || !(p.a_scale == p.a_scale) || !(p.c_split_scale == p.c_split_scale)) return WD_ERR_BAD_ARG;
  if (p.out_mode == WD_OUT_DECONV2X2) {
    if (p.n % 16 || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.res) return WD_ERR_BAD_ARG;
    if (p.ldc < p.n / 4) return WD_ERR_BAD_ARG;
  } else if (p.out_mode == WD_OUT_ROWS) {
    if (p.ldc < p.n) return WD_ERR_BAD_ARG;
  } else {
    return WD_ERR_BAD_ARG;
  }
  if (p.res && p.ldres < p.n) return WD_ERR_BAD_ARG;
  if (p.c_batch_stride < 0 || (p.c_batch_stride > 0 && (p.out_mode != WD_OUT_ROWS || p.c_batch_stride < p.hout * p.wout)))
    return WD_ERR_BAD_ARG;
  if (p.seg_rows < 0 || (p.seg_rows > 0 && !(0 <= p.seg_end0 && p.seg_end0 <= p.seg_end1 && p.seg_end1 <= p.seg_rows)))
    return WD_ERR_BAD_ARG;
  return WD_OK;
}

}  // namespace

// defined in split_gemm_pre.hip: launch with pre-split activations (flags: WD_SPLIT_A / WD_SPLIT_C)
int wd_launch_presplit(const WdConvGemm& p, const void* w_split, float w_unscale, int cfg, int flags, hipStream_t st,
                       int ksplits = 1, float* ws = nullptr, long long ws_floats = 0);
