// conv_gemm.hip — fp32 implicit-GEMM convolution / linear / similarity kernels for gfx950.
//
// One LDS-tiled MFMA template serves every dense contraction of the WeDetect image tower:
// ConvNeXt pointwise MLPs, patchify convs, neck/head 1x1 and 3x3 convs, the 2x2 transposed
// conv, the region x text similarity GEMM and the retrieval similarity with fused
// sigmoid + max-over-regions epilogue.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 — f32 inputs, f32 accumulate, bit-equal to an fmaf
// chain (the reference computes in fp32; no reduced precision anywhere).  MFMA roof for
// this dtype is 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Tiling (per workgroup of WM x WN waves, wave = 64 lanes):
//   block tile  BM = 16*TM*WM rows (m)  x  BN = 16*TN*WN cols (n),  K step BK (16 in production)
//   wave tile   (16*TM) x (16*TN), TM*TN accumulators of 4 VGPRs
//   production: 128 x 128 x 16 with 8 waves (32 x 64 wave tiles) for n % 128 == 0, and
//   128 x {96,80,64,48} x 16 with 8 waves stacked along m otherwise; ~76 VGPRs and 49 KB of LDS,
//   so three workgroups (6 waves per SIMD) stay resident per CU — on device that occupancy, not
//   fragment reuse, is what keeps the (slow, 32-cycle) fp32 MFMA pipe fed.
//   operands are staged global -> VGPR (16-byte loads, im2col gather; tails and conv padding are
//   READ from a zero block so the K loop is branch-free) -> LDS [rows][BK+8], double buffered:
//   the loads of tile t+1 are issued before the MFMAs of tile t and written to the other LDS
//   buffer after them (one barrier per K step).
//   The MFMA "A" operand carries WEIGHT rows and the "B" operand ACTIVATION rows, so a lane
//   ends up holding 4 consecutive output channels of one pixel: epilogue loads/stores are
//   16-byte vectors along n.
//   K is consumed 16 at a time: lane (i = lane&15, g = lane>>4) reads the float4
//   [k0+4g, k0+4g+4) of row i with one ds_read_b128 and feeds component r to the r-th of 4
//   MFMAs — a permutation of k that is identical for both operands, hence harmless; every tile
//   configuration therefore sums k in the same order and results are bit-identical across them.
//   Workgroups are renumbered so that each XCD (b % 8) walks a contiguous range of tiles: the
//   column tiles that share an A row panel hit the same L2.
#include "common.h"
#include "gemm_loader.h"

namespace {

// Template int VAR — variants selectable through wd_conv_gemm_tuned for on-device A/B runs
// (profiles/r01_gemm_ab.txt records what was measured).
constexpr int VAR_PRIO = 1;        // s_setprio(1) over the MFMA cluster (measured: null)
constexpr int VAR_PIN = 2;         // sched_barrier fences: loads first, MFMAs, then wait + LDS store
                                   // (measured: +8 % on small low-occupancy problems, -4..-10 % on saturated ones)
constexpr int VAR_XCD = 256;       // XCD-aware tile order: each XCD's L2 sees whole A row panels (+0..5 %)
// timing-only ablations (WRONG results by construction): skip the in-loop global loads / LDS
// stores / barrier / the epilogue
constexpr int ABL_NOLOAD = 4, ABL_NOBAR = 8, ABL_NOEPI = 16, ABL_NOLDS = 32;
// Tried and dropped in round 1 (all bit-exact, none faster): prefetch two K steps ahead with two
// register sets (-6..-13 %), hoisting / prefetching the epilogue operands (-9 %), staggering the
// co-resident workgroups (0 %), sched_group_barrier interleave (0 %), v_mfma_f32_32x32x2_f32
// tiles (0..-6 %), 4-wave 64x64 wave tiles and K step 32/64 (-10..-30 %), 6/10/12-wave workgroups
// (96/160/192-row tiles, -10..-25 %: waves no longer spread evenly over the 4 SIMDs).


template <int TM, int TN, int WM, int WN, int BK_ = 32>
struct Tile {
  static constexpr int BK = BK_;
  // Row pitch in floats.  BK + 8 (10 / 6 sixteen-byte slots for BK = 32 / 16), not BK + 4: ds_read_b128 is served in four
  // groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 — over a 256-byte bank row of sixteen
  // 16-byte slots; a fragment read puts lane (i = lane & 15, g = lane >> 4) at slot (pitch * i + g) mod 16.  With a pitch of
  // 9 (or 5) slots seven slots of every group are hit twice (two LDS cycles per group: SQ_LDS_BANK_CONFLICT was 35 % of
  // SQ_LDS_IDX_ACTIVE on the similarity GEMM, round-3 review); 10 and 6 are the smallest pitches with 16 distinct slots in
  // all four groups.  The staging ds_write_b128 (8 consecutive lanes = one row's 8 consecutive chunks) is conflict-free at
  // any pitch.  Same arithmetic: results are bit-identical.
  static constexpr int LD = BK + 8;
  static constexpr int KCH = BK / 4;                     // float4 chunks per row per K step
  static constexpr int BM = 16 * TM * WM;
  static constexpr int BN = 16 * TN * WN;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int A_PT = (BM * KCH) / NT;           // float4 chunks of A per thread per K step
  static constexpr int B_PT = (BN * KCH + NT - 1) / NT;
  static constexpr int RSTEP = NT / KCH;
  static constexpr int BN_LDS = B_PT * RSTEP;            // >= BN: every thread stores unconditionally
  static constexpr int LDS_BYTES = 2 * (BM + BN_LDS) * LD * 4;
  static_assert((BM * KCH) % NT == 0, "A tile must split evenly over the threads");
  static_assert(NT % KCH == 0, "threads must tile the K chunks");
};


// Rows of one image's region-embedding block for wd_retrieval_max.
template <int A_PT, int RSTEP>
struct RegionLoader {
  const float* base;
  int dim, rows;
  int row[A_PT];
  __device__ __forceinline__ void init(const float* e, int img, int rows_per_img, int dim_, int r0) {
    base = e + (size_t)img * rows_per_img * dim_;
    dim = dim_; rows = rows_per_img;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) row[i] = r0 + i * RSTEP;
  }
  template <int BKS>
  __device__ __forceinline__ void load(int kbase, int kc4, f32x4 (&reg)[A_PT]) const {
    const int k = kbase + kc4;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const bool valid = row[i] < rows && k < dim;
      reg[i] = *reinterpret_cast<const f32x4*>(valid ? base + (size_t)row[i] * dim + k : g_zero4);
    }
  }
};

// ---------------------------------------------------------------------------------------
// Main loop shared by all kernels: acc[tm][tn] (+)= X[m, :] . W[n, :]
// ---------------------------------------------------------------------------------------
template <class T, int TM, int TN, int WN, int VAR, class AL, class PreLast>
__device__ __forceinline__ void gemm_mainloop(const AL& al, const float* __restrict__ w, int n0, int N, int K,
                                              f32x4 (&acc)[TM][TN], float* smem, PreLast&& pre_last) {
  constexpr int BK = T::BK, LD = T::LD, KCH = T::KCH;
  constexpr int BM = T::BM, A_PT = T::A_PT, B_PT = T::B_PT, RSTEP = T::RSTEP, BN_LDS = T::BN_LDS;
  float* As = smem;
  float* Bs = smem + 2 * BM * LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kc = t % KCH, r0 = t / KCH;
  const int nk = (K + BK - 1) / BK;

  f32x4 areg[A_PT], breg[B_PT];
  size_t boff[B_PT];
  bool bok[B_PT];
#pragma unroll
  for (int j = 0; j < B_PT; ++j) {
    const int n = n0 + r0 + j * RSTEP;
    bok[j] = (r0 + j * RSTEP) < T::BN && n < N;
    boff[j] = bok[j] ? (size_t)n * K : 0;
  }

  auto load_b = [&](int k) {
    const bool kok = k < K;
    const int kk = kok ? k : 0;
#pragma unroll
    for (int j = 0; j < B_PT; ++j)
      breg[j] = *reinterpret_cast<const f32x4*>((bok[j] && kok) ? w + boff[j] + kk : g_zero4);
  };
  auto store = [&](int buf) {
    float* ad = As + buf * BM * LD + r0 * LD + kc * 4;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) *reinterpret_cast<f32x4*>(ad + i * RSTEP * LD) = areg[i];
    float* bd = Bs + buf * BN_LDS * LD + r0 * LD + kc * 4;
#pragma unroll
    for (int j = 0; j < B_PT; ++j) *reinterpret_cast<f32x4*>(bd + j * RSTEP * LD) = breg[j];
  };
  auto compute = [&](int buf) {
    const float* as = As + buf * BM * LD + (wm * TM * 16 + (lane & 15)) * LD + 4 * (lane >> 4);
    const float* bs = Bs + buf * BN_LDS * LD + (wn * TN * 16 + (lane & 15)) * LD + 4 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f32x4 xf[TM], wf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const f32x4*>(as + tm * 16 * LD + ks * 16);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const f32x4*>(bs + tn * 16 * LD + ks * 16);
      if (VAR & VAR_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn][r], xf[tm][r], acc[tm][tn], 0, 0, 0);
      if (VAR & VAR_PRIO) __builtin_amdgcn_s_setprio(0);
    }
  };

#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  al.template load<BK>(0, kc * 4, areg);
  load_b(kc * 4);
  store(0);
  __syncthreads();

  int cur = 0;
  for (int kt = 1; kt < nk; ++kt) {
    const int k = kt * BK + kc * 4;
    if (!(VAR & ABL_NOLOAD)) {
      al.template load<BK>(kt * BK, kc * 4, areg);   // tile kt: global -> VGPR, in flight during the MFMAs of tile kt-1
      load_b(k);
    }
    // Without these fences hipcc sinks the two global loads down to their only consumer (the
    // ds_write at the end of the step) and waits for them on the spot, exposing the whole
    // memory round trip every K step; pinned, the loads fly during this step's MFMAs.
    if (VAR & VAR_PIN) __builtin_amdgcn_sched_barrier(0);
    compute(cur);
    if (VAR & VAR_PIN) __builtin_amdgcn_sched_barrier(0);
    if (!(VAR & ABL_NOLDS)) store(cur ^ 1);
    if (!(VAR & ABL_NOBAR)) __syncthreads();
    cur ^= 1;
  }
  pre_last();          // epilogue operand loads (bias, residual) fly during the last K step's MFMAs
  compute(cur);
}

// ---------------------------------------------------------------------------------------
// Epilogue.  Lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] per 16x16 tile.
// SPECIAL = false: row-major output, no scale/bias/sigmoid (every MLP / conv layer);
// SPECIAL = true : deconv scatter, batch-strided rows, per-level affine, sigmoid.
// ---------------------------------------------------------------------------------------
struct EpiCtx { int m0, n0, wm, wn, lane, vec_c, vec_res, vec_bias; };

template <int ACT>
__device__ __forceinline__ float act_fast(float v) {
  if (ACT == WD_ACT_RELU) return fmaxf(v, 0.0f);
  if (ACT == WD_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));   // <= 2 ulp
  if (ACT == WD_ACT_GELU) return wd_gelu(v);
  return v;
}

template <int TM, int TN, int ACT, bool SPECIAL>
__device__ __forceinline__ void epilogue(const WdConvGemm& p, const EpiCtx& ec, const f32x4 (&acc)[TM][TN]) {
  const int lane = ec.lane;
  const int ncq = p.n >> 2;   // deconv: channels per tap
  const bool has_res = p.res != nullptr;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = ec.m0 + (ec.wm * TM + tm) * 16 + (lane & 15);
    if (m >= p.m) continue;
    size_t crow = (size_t)m;
    int hw2 = 0;
    float oscale = 1.0f, obias = 0.0f;
    if (SPECIAL) {
      if (p.out_mode == WD_OUT_DECONV2X2) {
        const int wq = m % p.wout;
        const int q = m / p.wout;
        const int hq = q % p.hout;
        const int b = q / p.hout;
        crow = ((size_t)(b * 2 * p.hout + 2 * hq) * (2 * p.wout) + 2 * wq);   // pixel of tap (0,0)
        hw2 = 2 * p.wout;
      } else if (p.c_batch_stride > 0) {
        const int hw = p.hout * p.wout;
        const int b = m / hw;
        crow = (size_t)b * p.c_batch_stride + (size_t)(m - b * hw);
      }
      oscale = p.out_scale; obias = p.out_bias;
      if (p.seg_rows > 0) {
        const int pos = m % p.seg_rows;
        const int lvl = (pos >= p.seg_end0) + (pos >= p.seg_end1);
        oscale = lvl == 0 ? p.seg_scale[0] : lvl == 1 ? p.seg_scale[1] : p.seg_scale[2];
        obias = lvl == 0 ? p.seg_bias[0] : lvl == 1 ? p.seg_bias[1] : p.seg_bias[2];
      }
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = ec.n0 + (ec.wn * TN + tn) * 16 + 4 * (lane >> 4);
      if (n >= p.n) continue;
      const f32x4 v = acc[tm][tn];
      const bool full = n + 3 < p.n;
      f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        if (full && ec.vec_bias) {
          b4 = *reinterpret_cast<const f32x4*>(p.bias + n);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.n) b4[r] = p.bias[n + r];
        }
      }
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = act_fast<ACT>(v[r] + b4[r]);
        if (SPECIAL) {
          x = x * oscale + obias;
          if (p.sigmoid) x = wd_sigmoid_fast(x);
        }
        o[r] = x;
      }
      if (has_res) {
        const float* rp = p.res + (size_t)m * p.ldres + n;
        if (full && ec.vec_res) {
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
        const int tap = n / ncq, co = n - tap * ncq;
        cp = p.c + (crow + (size_t)(tap >> 1) * hw2 + (tap & 1)) * p.ldc + co;
      } else {
        cp = p.c + crow * p.ldc + n;
      }
      if (full && ec.vec_c) {
        *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.n) cp[r] = o[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// conv / linear / similarity kernel
// ---------------------------------------------------------------------------------------
// Register budget: ask for the occupancy the tile was tuned at (the second __launch_bounds__
// argument is waves per SIMD).  Without it hipcc spends AGPRs on the 4-wave kernels (137
// registers, 3 waves per SIMD) and the 64x128 tile loses 8 %.
constexpr int min_waves_per_simd(int waves, int acc_tiles) {
  return (acc_tiles > 8) ? 1 : (waves == 4) ? 5 : (waves == 8) ? 6 : 1;
}

template <int TM, int TN, int WM, int WN, bool CONV, int BKT = 32, int VAR = 0>
__global__ void __launch_bounds__(64 * WM * WN, min_waves_per_simd(WM * WN, TM * TN)) conv_gemm_kernel(const WdConvGemm p, int nbn, int vec_c, int vec_res, int vec_bias) {
  using T = Tile<TM, TN, WM, WN, BKT>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  int tile = blockIdx.x;
  if (VAR & VAR_XCD) {
    // workgroup b runs on XCD b % 8 (observed; only locality depends on it).  Give XCD x the
    // contiguous tile range [start(x), start(x+1)): bijective for any grid size.
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int bn = tile % nbn, bm = tile / nbn;
  const int m0 = bm * T::BM, n0 = bn * T::BN;

  ALoader<T::A_PT, T::RSTEP, CONV> al;
  al.init(p, m0, t / T::KCH);
  f32x4 acc[TM][TN];
  gemm_mainloop<T, TM, TN, WN, VAR>(al, p.w, n0, p.n, p.k, acc, smem, [] {});

  if (VAR & ABL_NOEPI) {           // keep the accumulators live, store one value per lane
    f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) s4 += acc[tm][tn];
    if (m0 + (int)(t % T::BM) < p.m) p.c[(size_t)(m0 + t % T::BM) * p.ldc + n0] = s4[0] + s4[1] + s4[2] + s4[3];
    return;
  }
  // ---- epilogue, specialised at compile time on (activation, residual, plain/special output)
  // so that the executed path is straight-line code: with the switches inside the unrolled
  // element loops the epilogue was ~6000 instructions and 13 % of the short-K GEMMs.
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
  EpiCtx ec{m0, n0, wm, wn, lane, vec_c, vec_res, vec_bias};
  if (special) {
    switch (p.act) {
      case WD_ACT_RELU: epilogue<TM, TN, WD_ACT_RELU, true>(p, ec, acc); break;
      case WD_ACT_SILU: epilogue<TM, TN, WD_ACT_SILU, true>(p, ec, acc); break;
      case WD_ACT_GELU: epilogue<TM, TN, WD_ACT_GELU, true>(p, ec, acc); break;
      default: epilogue<TM, TN, WD_ACT_NONE, true>(p, ec, acc); break;
    }
  } else {
    switch (p.act) {
      case WD_ACT_RELU: epilogue<TM, TN, WD_ACT_RELU, false>(p, ec, acc); break;
      case WD_ACT_SILU: epilogue<TM, TN, WD_ACT_SILU, false>(p, ec, acc); break;
      case WD_ACT_GELU: epilogue<TM, TN, WD_ACT_GELU, false>(p, ec, acc); break;
      default: epilogue<TM, TN, WD_ACT_NONE, false>(p, ec, acc); break;
    }
  }
}

// ---------------------------------------------------------------------------------------
// retrieval similarity: one workgroup = (64 classes, one image of <= 320 region rows)
// ---------------------------------------------------------------------------------------
constexpr int RT_TM = 5, RT_TN = 2, RT_WM = 4, RT_WN = 2;   // 320 rows x 64 classes, 8 waves of 80 x 32

__global__ void __launch_bounds__(64 * RT_WM * RT_WN) retrieval_max_kernel(const float* __restrict__ e, const float* __restrict__ tb,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ bias,
                                                            const int* __restrict__ count, float* __restrict__ out,
                                                            int rows_per_img, int n_cls, int dim) {
  using T = Tile<RT_TM, RT_TN, RT_WM, RT_WN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / RT_WN, wn = wave % RT_WN;
  const int img = blockIdx.y;
  const int n0 = blockIdx.x * T::BN;

  RegionLoader<T::A_PT, T::RSTEP> al;
  al.init(e, img, rows_per_img, dim, t / T::KCH);
  f32x4 acc[RT_TM][RT_TN];
  gemm_mainloop<T, RT_TM, RT_TN, RT_WN, 0>(al, tb, n0, n_cls, dim, acc, smem, [] {});

  const int cnt = count[img];
  float cmax[RT_TN][4];
#pragma unroll
  for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) cmax[tn][r] = 0.f;   // sigmoid > 0: 0 is the identity of max here
#pragma unroll
  for (int tm = 0; tm < RT_TM; ++tm) {
    const int row = (wm * RT_TM + tm) * 16 + (lane & 15);
    if (row < cnt && row < rows_per_img) {
      const float s = expf(scale[(size_t)img * rows_per_img + row]);
      const float b = bias[(size_t)img * rows_per_img + row];
#pragma unroll
      for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmax[tn][r] = fmaxf(cmax[tn][r], wd_sigmoid(acc[tm][tn][r] * s + b));
    }
  }
  // max over the 16 rows held by lanes with equal (lane>>4)
#pragma unroll
  for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = cmax[tn][r];
      v = fmaxf(v, __shfl_xor(v, 1, 64));
      v = fmaxf(v, __shfl_xor(v, 2, 64));
      v = fmaxf(v, __shfl_xor(v, 4, 64));
      v = fmaxf(v, __shfl_xor(v, 8, 64));
      cmax[tn][r] = v;
    }
  __syncthreads();                 // all waves are done with the operand tiles in LDS
  float* red = smem;               // [RT_WM row groups][64 classes]
  if ((lane & 15) == 0) {
#pragma unroll
    for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wm * 64 + (wn * RT_TN + tn) * 16 + 4 * (lane >> 4) + r] = cmax[tn][r];
  }
  __syncthreads();
  if (t < 64) {
    const int n = n0 + t;
    if (n < n_cls) {
      const float v = fmaxf(fmaxf(red[t], red[64 + t]), fmaxf(red[128 + t], red[192 + t]));
      out[(size_t)img * n_cls + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN, int BKT = 32, int VAR = 0>
int launch_cfg(const WdConvGemm& p, hipStream_t st) {
  using T = Tile<TM, TN, WM, WN, BKT>;
  const bool conv = !(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0);
  const int nbm = (p.m + T::BM - 1) / T::BM, nbn = (p.n + T::BN - 1) / T::BN;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c) && (p.out_mode == WD_OUT_ROWS || (p.n % 16 == 0));
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  static WdAttrOnce attr_plain, attr_conv;
  if (conv) {
    auto k = conv_gemm_kernel<TM, TN, WM, WN, true, BKT, VAR>;
    if (wd_set_max_lds(attr_conv, reinterpret_cast<const void*>(k), T::LDS_BYTES) != WD_OK) return WD_ERR_LAUNCH;
    WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, nbn, vec_c, vec_res, vec_bias);
  } else {
    auto k = conv_gemm_kernel<TM, TN, WM, WN, false, BKT, VAR>;
    if (wd_set_max_lds(attr_plain, reinterpret_cast<const void*>(k), T::LDS_BYTES) != WD_OK) return WD_ERR_LAUNCH;
    WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, nbn, vec_c, vec_res, vec_bias);
  }
  return wd_launch_status();
}

// tile widths available along n, widest first; pick the least padded, ties -> widest
constexpr int kBnChoices[5] = {128, 96, 80, 64, 48};

int pick_bn(int n) {
  int best = 128;
  long long best_pad = -1;
  for (int i = 0; i < 5; ++i) {
    const int bn = kBnChoices[i];
    const long long padded = (long long)((n + bn - 1) / bn) * bn;
    if (best_pad < 0 || padded < best_pad) { best_pad = padded; best = bn; }
  }
  return best;
}

}  // namespace

// Production tile table (picked on device with scripts/gemm_bench.py, profiles/r01_gemm_ab.txt):
// K step 16 and 8 waves per workgroup keep 4-5 waves per SIMD resident, which is what keeps
// the fp32 MFMA pipe busy; results are bit-identical across configurations (same k order).
// n % 128 == 0 problems with fewer than this many 128x128 tiles run 64x128 tiles (4 waves) instead:
// twice the workgroups and five resident per CU cut the tile-quantisation tail (s3 pwconv2
// 114 -> 120 TF, s4 pwconv2 101 -> 112 TF); above it the 8-wave 128x128 tile is faster.
constexpr long long kSmallProblemTiles = 2048;

static inline bool use_small_tile(int m, int n) {
  return (long long)((m + 127) / 128) * ((n + 127) / 128) < kSmallProblemTiles;
}

extern "C" const char* wd_conv_gemm_config(int32_t m, int32_t n, int32_t k) {
  (void)k;
  switch (pick_bn(n)) {
    case 128: return use_small_tile(m, n) ? "64x128x16/4w" : "128x128x16/8w";
    case 96: return "128x96x16/8w";
    case 80: return "64x80x32/4w";
    case 64: return "128x64x16/8w";
    default: return "128x48x16/8w";
  }
}

extern "C" int wd_conv_gemm(const WdConvGemm* pp, void* stream) {
  if (!pp) return WD_ERR_BAD_ARG;
  const WdConvGemm& p = *pp;
  if (!p.a || !p.w || !p.c) return WD_ERR_BAD_ARG;
  if (p.ln_stats || p.ln_u) return WD_ERR_UNSUPPORTED;              // the LayerNorm fold exists in the fp16x3 C-split epilogue only
  if (p.m <= 0 || p.n <= 0 || p.k <= 0) return WD_ERR_BAD_ARG;
  if (p.cin <= 0 || p.cin % 4 || p.lda % 4 || p.lda < p.cin) return WD_ERR_BAD_ARG;
  if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0) return WD_ERR_BAD_ARG;
  if (p.k != p.kh * p.kw * p.cin) return WD_ERR_BAD_ARG;
  if ((long long)p.batch * p.hout * p.wout != (long long)p.m) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(p.a) || !wd_aligned16(p.w)) return WD_ERR_BAD_ARG;
  if (p.act < WD_ACT_NONE || p.act > WD_ACT_GELU) return WD_ERR_BAD_ARG;
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
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pick_bn(p.n)) {
    case 128:
      return use_small_tile(p.m, p.n) ? launch_cfg<2, 4, 2, 2, 16, VAR_XCD>(p, st)
                                      : launch_cfg<2, 4, 4, 2, 16, VAR_XCD>(p, st);
    case 96: return launch_cfg<1, 6, 8, 1, 16, VAR_XCD>(p, st);
    // the similarity GEMM: pinned order +2..5 %, K step 32 +3 %, 64-row tiles of 4 waves (4200 instead of 2100
    // tiles at B = 32: a shorter tail) +2 % inside the step (profiles/r01_gemm_ab.txt)
    case 80: return launch_cfg<1, 5, 4, 1, 32, VAR_XCD | VAR_PIN>(p, st);
    case 64: return launch_cfg<1, 4, 8, 1, 16, VAR_XCD>(p, st);
    default: return launch_cfg<1, 3, 8, 1, 16, VAR_XCD>(p, st);
  }
}

// Experimental tile configurations for on-device A/B runs (scripts/gemm_bench.py); not used
// by the product path until a winner is promoted into wd_conv_gemm's table.
extern "C" int wd_conv_gemm_tuned(const WdConvGemm* pp, int32_t cfg, void* stream) {
  if (!pp) return WD_ERR_BAD_ARG;
  const WdConvGemm& p = *pp;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (cfg) {
    case 0: return launch_cfg<4, 4, 2, 2, 32, 0>(p, st);                  // 128x128x32, 4 waves (first cut)
    case 1: return launch_cfg<2, 4, 4, 2, 32, 0>(p, st);                  // 128x128x32, 8 waves
    case 2: return launch_cfg<2, 4, 4, 2, 16, 0>(p, st);                  // 128x128x16, 8 waves
    case 3: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD>(p, st);            // production tile for n % 128 == 0
    case 4: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | VAR_PRIO>(p, st);
    case 5: return launch_cfg<2, 2, 4, 4, 32, VAR_XCD>(p, st);            // 128x128x32, 16 waves
    case 6: return launch_cfg<4, 4, 2, 2, 16, VAR_XCD>(p, st);            // 128x128x16, 4 waves
    case 7: return launch_cfg<1, 8, 4, 1, 16, VAR_XCD>(p, st);            // 64x128x16, 4 waves
    case 8: return launch_cfg<1, 4, 4, 1, 16, VAR_XCD>(p, st);            // 64x64x16, 4 waves
    case 9: return launch_cfg<1, 4, 8, 1, 16, VAR_XCD>(p, st);            // 128x64x16, 8 waves (production, n = 64)
    case 10: return launch_cfg<1, 5, 8, 1, 16, VAR_XCD>(p, st);           // 128x80x16, 8 waves
    case 11: return launch_cfg<4, 5, 4, 1, 32, 0>(p, st);                 // 256x80x32, 4 waves (first cut)
    case 12: return launch_cfg<2, 5, 8, 1, 16, VAR_XCD>(p, st);           // 256x80x16, 8 waves
    case 13: return launch_cfg<2, 4, 2, 2, 16, VAR_XCD>(p, st);           // 64x128x16, 4 waves (production, small problems)
    case 14: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | VAR_PIN>(p, st);
    case 15: return launch_cfg<2, 4, 2, 2, 16, VAR_XCD | VAR_PIN>(p, st);
    case 16: return launch_cfg<1, 5, 8, 1, 16, VAR_XCD | VAR_PIN>(p, st);
    case 17: return launch_cfg<1, 4, 8, 1, 16, VAR_XCD | VAR_PIN>(p, st);
    case 19: return launch_cfg<1, 5, 8, 1, 32, VAR_XCD | VAR_PIN>(p, st);  // 128x80x32, 8 waves
    case 25: return launch_cfg<1, 5, 4, 1, 32, VAR_XCD | VAR_PIN>(p, st);  // 64x80x32, 4 waves (production, n = 80)
    case 26: return launch_cfg<2, 5, 4, 1, 32, VAR_XCD | VAR_PIN>(p, st);  // 128x80x32, 4 waves of 32 rows
    case 20: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | ABL_NOLOAD>(p, st);
    case 21: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | ABL_NOBAR>(p, st);
    case 22: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | ABL_NOEPI>(p, st);
    case 24: return launch_cfg<2, 4, 4, 2, 16, VAR_XCD | ABL_NOLOAD | ABL_NOLDS | ABL_NOBAR | ABL_NOEPI>(p, st);
    default: return WD_ERR_UNSUPPORTED;
  }
}

extern "C" int wd_retrieval_max(const float* e, const float* t, const float* scale, const float* bias,
                                const int32_t* count, float* out, int32_t n_img, int32_t rows_per_img,
                                int32_t n_cls, int32_t dim, void* stream) {
  using T = Tile<RT_TM, RT_TN, RT_WM, RT_WN>;
  if (!e || !t || !scale || !bias || !count || !out) return WD_ERR_BAD_ARG;
  if (n_img <= 0 || n_cls <= 0 || dim <= 0 || dim % 4) return WD_ERR_BAD_ARG;
  if (rows_per_img <= 0 || rows_per_img > T::BM) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(e) || !wd_aligned16(t)) return WD_ERR_BAD_ARG;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(retrieval_max_kernel), T::LDS_BYTES) != WD_OK) return WD_ERR_LAUNCH;
  const int nbn = (n_cls + T::BN - 1) / T::BN;
  hipLaunchKernelGGL(retrieval_max_kernel, dim3(nbn, n_img), dim3(T::NT), T::LDS_BYTES,
                     static_cast<hipStream_t>(stream), e, t, scale, bias, count, out, rows_per_img, n_cls, dim);
  return wd_launch_status();
}
