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
// timing-only ablations (WRONG results by construction) for on-device diagnosis
constexpr int SVAR_PF2 = 128;      // global loads run two K stages ahead (two register sets)
constexpr int SVAR_LDSEPI = 512;   // epilogue through LDS: every global access of C / residual / bias is a full 128-byte row segment
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
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  hi = u32x2{__builtin_bit_cast(unsigned, ha), __builtin_bit_cast(unsigned, hb)};
  lo = u32x2{__builtin_bit_cast(unsigned, la), __builtin_bit_cast(unsigned, lb)};
}

// ---------------------------------------------------------------------------------------
// weight preparation: fp32 [n][k] -> per row, per 8 k: [8 x fp16 hi | 8 x fp16 lo] of w*scale,
// rows zero-padded to a multiple of 16 k.
// ---------------------------------------------------------------------------------------
__global__ void split_weights_kernel(const float* __restrict__ w, int n, int k, int k16, float scale,
                                     unsigned short* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * k16) return;
  const int row = (int)(idx / k16), kk = (int)(idx % k16);
  const float x = kk < k ? w[(size_t)row * k + kk] * scale : 0.0f;
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)(x - (float)hi);
  unsigned short* o = out + (size_t)row * 2 * k16 + (kk >> 3) * 16 + (kk & 7);
  o[0] = __builtin_bit_cast(unsigned short, hi);
  o[8] = __builtin_bit_cast(unsigned short, lo);
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
      x = x * er.oscale + er.obias;
      if (p.sigmoid) x = wd_sigmoid(x);
    }
    o[r] = x;
  }
  if (p.res != nullptr) {
    const float* rp = p.res + (size_t)m * p.ldres + n;
    if (full && ev.res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += p.res_alpha * rv[r];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n + r < p.n) o[r] += p.res_alpha * rp[r];
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
                  int k16, float unscale, int nbn, int vec_c, int vec_res, int vec_bias) {
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
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int kc = t % KCH, r0 = t / KCH;
  const int K = p.k;
  const int nk = (K + BK - 1) / BK;

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
    unsigned char* ad = As + (buf * BM + r0) * ROWB + a_off;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      u32x2 hi, lo;
      if (VAR & SABL_NOSPLIT) {
        hi = u32x2{__builtin_bit_cast(unsigned, areg[S][i][0]), __builtin_bit_cast(unsigned, areg[S][i][1])};
        lo = u32x2{__builtin_bit_cast(unsigned, areg[S][i][2]), __builtin_bit_cast(unsigned, areg[S][i][3])};
      } else {
        split4(areg[S][i], hi, lo);
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

  load(0, Set0{});
  store(0, Set0{});
  __syncthreads();
  if (VAR & SVAR_PF2) {
    // stage s is computed from LDS buffer s & 1 while stage s+1 waits in registers and the loads
    // of stage s+2 are issued: a global load has a whole K stage of MFMAs to land
    load(BK, Set1{});
    for (int s = 0; s < nk; s += 2) {
      load((s + 2) * BK, Set0{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(0);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      store(1, Set1{});
      __syncthreads();
      if (s + 1 >= nk) break;
      load((s + 3) * BK, Set1{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(1);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      store(0, Set0{});
      __syncthreads();
    }
  } else {
    int cur = 0;
    for (int kt = 1; kt < nk; ++kt) {
      if (!(VAR & SABL_NOLOAD)) load(kt * BK, Set0{});
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      compute(cur);
      if (VAR & SVAR_PIN) __builtin_amdgcn_sched_barrier(0);
      if (!(VAR & SABL_NOLDS)) store(cur ^ 1, Set0{});
      if (!(VAR & SABL_NOBAR)) __syncthreads();
      cur ^= 1;
    }
    compute(cur);
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

template <int TM, int TN, int WM, int WN, int BKT, int VAR>
int launch_split(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st) {
  using T = STile<TM, TN, WM, WN, BKT>;
  const bool conv = !(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0);
  const int nbm = (p.m + T::BM - 1) / T::BM, nbn = (p.n + T::BN - 1) / T::BN;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c) && (p.out_mode == WD_OUT_ROWS || (p.n % 16 == 0));
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  const unsigned char* w8 = static_cast<const unsigned char*>(wsp);
  static const float* zero = nullptr;
  if (!zero) {
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero4)) != hipSuccess || !zp) return WD_ERR_LAUNCH;
    zero = static_cast<const float*>(zp);
  }
  static bool attr_plain = false, attr_conv = false;
  if (conv) {
    auto k = split_gemm_kernel<TM, TN, WM, WN, BKT, true, VAR>;
    if (!attr_conv) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_BYTES) != hipSuccess) return WD_ERR_LAUNCH;
      attr_conv = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, w8, zero, k16, unscale, nbn,
                       vec_c, vec_res, vec_bias);
  } else {
    auto k = split_gemm_kernel<TM, TN, WM, WN, BKT, false, VAR>;
    if (!attr_plain) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_BYTES) != hipSuccess) return WD_ERR_LAUNCH;
      attr_plain = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, w8, zero, k16, unscale, nbn,
                       vec_c, vec_res, vec_bias);
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

extern "C" int64_t wd_split_weights_bytes(int32_t n, int32_t k) {
  if (n <= 0 || k <= 0) return 0;
  return (int64_t)n * ((k + 15) / 16 * 16) * 4;
}

extern "C" int wd_split_weights(const float* w, int32_t n, int32_t k, float scale, void* out, void* stream) {
  if (!w || !out || n <= 0 || k <= 0 || !(scale > 0.0f) || !wd_aligned16(out)) return WD_ERR_BAD_ARG;
  const int k16 = (k + 15) / 16 * 16;
  const long long total = (long long)n * k16;
  hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), w, n, k, k16, scale, static_cast<unsigned short*>(out));
  return wd_launch_status();
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

extern "C" const char* wd_conv_gemm_split_config(int32_t m, int32_t n, int32_t k, int32_t is_conv) {
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

// cfg < 0: production choice.  Other values select a tile for on-device A/B runs.
extern "C" int wd_conv_gemm_split(const WdConvGemm* pp, const void* w_split, float w_unscale, int32_t cfg,
                                  void* stream) {
  if (!pp) return WD_ERR_BAD_ARG;
  const WdConvGemm& p = *pp;
  const int rc = check_split_args(p, w_split, w_unscale);
  if (rc != WD_OK) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (cfg < 0) cfg = pick_split_cfg(p);
  switch (cfg) {
    case 41: return launch_split<2, 2, 2, 4, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st);  // 128 x 256 x 16, 8 waves
    case 50: return launch_split<2, 2, 2, 2, 32, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st);  // 128 x 128 x 32, 4 waves
    case 51: return launch_split<2, 2, 2, 2, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st);  // 128 x 128 x 16, 4 waves
    case 52: return launch_split<2, 2, 4, 1, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st);  // 256 x 64 x 16, 4 waves
    case 53: return launch_split<2, 2, 2, 1, 16, SVAR_XCD | SVAR_PIN | SVAR_LDSEPI>(p, w_split, w_unscale, st);  // 128 x 64 x 16, 2 waves
    // prefetch distance 2 (two staging register sets; only the BK = 32 plain kernel holds them without spilling)
    case 55: return launch_split<2, 2, 2, 2, 32, SVAR_XCD | SVAR_PF2 | SVAR_LDSEPI>(p, w_split, w_unscale, st);
    default: return WD_ERR_UNSUPPORTED;
  }
}
