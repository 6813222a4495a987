// split_epi_oct.h — the (row, 8 channels) epilogue of the pre-split kernels that write fp16 hi/lo groups, fp32, or both
// (split_gemm_conv.hip, split_gemm_mlp.hip).
#pragma once
#include "split_gemm_impl.h"

namespace {

// ---------------------------------------------------------------------------------------
// Epilogue of one (row, 8 consecutive channels) piece — the unit of the fp16 hi/lo output format.  Same
// arithmetic, same order per element as epi_quad (split_gemm_impl.h).  Outputs, any combination the host asks for:
//   CSPLIT : p.c receives [hi x8 | lo x8] groups (rows / batch-stride rows / the 2x2 deconv scatter)
//   !CSPLIT: p.c receives fp32
//   p.c2   : an fp32 copy in plain rows (ldc2; with c_batch_stride: the rows of c) next to a CSPLIT output — for consumers
//            that read fp32 (the BottleRep "+ alpha x" residual, yolo_world_pafpn.py:602-605; round 6: the region embeddings)
// Host-side contract: n % 8 == 0, bias / res / c / c2 16-byte aligned, ldres % 4 == 0, ldc2 % 4 == 0,
// ldc % 8 == 0 (CSPLIT) or % 4 == 0.
// ---------------------------------------------------------------------------------------
// b0 / b1: the bias quads of channels n .. n + 7 (zeros without a bias); r0 / r1: the residual quads (read only if p.res).
template <int ACT, bool SPECIAL, bool CSPLIT>
__device__ __forceinline__ void epi_oct_core(const WdConvGemm& p, float unscale, int m, int n, const f32x4 v0, const f32x4 v1,
                                             const f32x4 b0, const f32x4 b1, const f32x4 r0, const f32x4 r1) {
  if (p.range_flag) {
    if (wd_any_nonfinite4(v0[0] + v0[1], v0[2] + v0[3], v1[0] + v1[1], v1[2] + v1[3])) *p.range_flag = 1u;
  }
  const EpiRow er = epi_row<SPECIAL>(p, m);
  f32x4 o0, o1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float x0 = sact<ACT>(fmaf(v0[r], unscale, b0[r]));
    float x1 = sact<ACT>(fmaf(v1[r], unscale, b1[r]));
    if (SPECIAL) {
      x0 = fmaf(x0, er.oscale, er.obias);
      x1 = fmaf(x1, er.oscale, er.obias);
      if (p.sigmoid) { x0 = wd_sigmoid_fast(x0); x1 = wd_sigmoid_fast(x1); }
    }
    o0[r] = x0;
    o1[r] = x1;
  }
  if (p.res != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o0[r] = fmaf(p.res_alpha, r0[r], o0[r]);
      o1[r] = fmaf(p.res_alpha, r1[r], o1[r]);
    }
  }
  if (p.c2 != nullptr) {
    // round 6: with c_batch_stride the fp32 twin takes the SAME per-image row mapping as c (the head's embedding conv writes
    // [B, anchors, 768] twice: hi/lo groups for the fp16x3 similarity GEMM, fp32 for the gather and the callers)
    float* qp = p.c2 + (SPECIAL && p.c_batch_stride > 0 ? er.crow : (size_t)m) * p.ldc2 + n;
    *reinterpret_cast<f32x4*>(qp) = o0;
    *reinterpret_cast<f32x4*>(qp + 4) = o1;
  }
  float* rowp;
  int col = n;
  if (SPECIAL && p.out_mode == WD_OUT_DECONV2X2) {
    const int ncq = p.n >> 2;
    const int tap = n / ncq;
    col = n - tap * ncq;
    rowp = p.c + (er.crow + (size_t)(tap >> 1) * er.hw2 + (tap & 1)) * p.ldc;
  } else {
    rowp = p.c + er.crow * p.ldc;
  }
  if (CSPLIT) {
    u32x2 h0, l0, h1, l1;
    f32x4 s0 = o0, s1 = o1;
    if (p.c_split_scale != 0.f && p.c_split_scale != 1.0f) {       // power of two (exact); wave-uniform branch, see epi_lds_tile_csplit
      s0 = o0 * p.c_split_scale;
      s1 = o1 * p.c_split_scale;
    }
    split4(s0, h0, l0);
    split4(s1, h1, l1);
    unsigned char* cp = reinterpret_cast<unsigned char*>(rowp) + (size_t)(col >> 3) * 32;
    *reinterpret_cast<u32x4*>(cp) = u32x4{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<u32x4*>(cp + 16) = u32x4{l0[0], l0[1], l1[0], l1[1]};
  } else {
    *reinterpret_cast<f32x4*>(rowp + col) = o0;
    *reinterpret_cast<f32x4*>(rowp + col + 4) = o1;
  }
}

// one piece with its own operand loads (the split-K second pass: one piece per thread)
template <int ACT, bool SPECIAL, bool CSPLIT>
__device__ __forceinline__ void epi_oct(const WdConvGemm& p, float unscale, int m, int n, const f32x4 v0, const f32x4 v1) {
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f}, r0 = b0, r1 = b0;
  if (p.bias) {
    b0 = *reinterpret_cast<const f32x4*>(p.bias + n);
    b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
  }
  if (p.res != nullptr) {
    const float* rp = p.res + (size_t)m * p.ldres + n;
    r0 = *reinterpret_cast<const f32x4*>(rp);
    r1 = *reinterpret_cast<const f32x4*>(rp + 4);
  }
  epi_oct_core<ACT, SPECIAL, CSPLIT>(p, unscale, m, n, v0, v1, b0, b1, r0, r1);
}

// Bias and residual operands of a whole wave tile, requested BEFORE the first output store.  vmcnt counts stores with
// loads and both retire in order: a load issued after a piece's stores (as the walk below did, one piece at a time) cannot
// be consumed until those stores have drained — every one of the 2 TM TN passes paid a store round trip.
// PRE = false (kernels held to 128 registers, which these 8 TN + 16 TM TN more would spill): nothing is preloaded, every
// piece reads its own operands as before.
template <int TM, int TN, bool PRE = true>
struct EpiOctOperands {
  static constexpr int NB = PRE ? TN : 1, NR = PRE ? TM * TN * 2 : 1;
  f32x4 b0[NB], b1[NB], r0[NR], r1[NR];
  __device__ __forceinline__ void load(const WdConvGemm& p, int mw, int nw, int lane) {
    if (!PRE) return;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = nw + j * 32 + 8 * (lane & 3);
      b0[j] = b1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.n) {
        b0[j] = *reinterpret_cast<const f32x4*>(p.bias + n);
        b1[j] = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
      }
    }
#pragma unroll
    for (int ij = 0; ij < TM * TN; ++ij)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int m = mw + (ij / TN) * 32 + ps * 16 + (lane >> 2), n = nw + (ij % TN) * 32 + 8 * (lane & 3);
        r0[ij * 2 + ps] = r1[ij * 2 + ps] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.res != nullptr && m < p.m && n < p.n) {
          const float* rp = p.res + (size_t)m * p.ldres + n;
          r0[ij * 2 + ps] = *reinterpret_cast<const f32x4*>(rp);
          r1[ij * 2 + ps] = *reinterpret_cast<const f32x4*>(rp + 4);
        }
      }
  }
};

// a wave's TM x TN accumulator tiles through its private LDS patch: a lane ends with 8 consecutive channels of a
// row, 4 lanes cover a 128-byte row segment
template <int IJ, int TM, int TN, int ACT, bool SPECIAL, bool CSPLIT, bool PRE = true>
struct EpiOctWalk {
  static __device__ __forceinline__ void run(const WdConvGemm& p, float unscale, int mw, int nw, int lane,
                                             const f32x16 (&acc)[TM][TN], float* patch, const EpiOctOperands<TM, TN, PRE>& ops) {
    constexpr int I = IJ / TN, J = IJ % TN;
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
        if (PRE)
          epi_oct_core<ACT, SPECIAL, CSPLIT>(p, unscale, m, n, v0, v1, ops.b0[PRE ? J : 0], ops.b1[PRE ? J : 0],
                                             ops.r0[PRE ? IJ * 2 + ps : 0], ops.r1[PRE ? IJ * 2 + ps : 0]);
        else
          epi_oct<ACT, SPECIAL, CSPLIT>(p, unscale, m, n, v0, v1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    EpiOctWalk<IJ + 1, TM, TN, ACT, SPECIAL, CSPLIT, PRE>::run(p, unscale, mw, nw, lane, acc, patch, ops);
  }
};
template <int TM, int TN, int ACT, bool SPECIAL, bool CSPLIT, bool PRE>
struct EpiOctWalk<TM * TN, TM, TN, ACT, SPECIAL, CSPLIT, PRE> {
  static __device__ __forceinline__ void run(const WdConvGemm&, float, int, int, int, const f32x16 (&)[TM][TN], float*,
                                             const EpiOctOperands<TM, TN, PRE>&) {}
};

template <int TM, int TN, bool CSPLIT, bool PRE = true>
__device__ __forceinline__ void epi_oct_all(const WdConvGemm& p, float unscale, int mw, int nw, int lane,
                                            const f32x16 (&acc)[TM][TN], float* patch) {
  const bool special = p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.seg_rows > 0 || p.sigmoid ||
                       p.out_scale != 1.0f || p.out_bias != 0.0f;
  EpiOctOperands<TM, TN, PRE> ops;
  ops.load(p, mw, nw, lane);
#define WD_OCT(A, S) EpiOctWalk<0, TM, TN, A, S, CSPLIT, PRE>::run(p, unscale, mw, nw, lane, acc, patch, ops)
  if (special) {
    switch (p.act) {
      case WD_ACT_RELU: WD_OCT(WD_ACT_RELU, true); break;
      case WD_ACT_SILU: WD_OCT(WD_ACT_SILU, true); break;
      case WD_ACT_GELU: WD_OCT(WD_ACT_GELU, true); break;
      default: WD_OCT(WD_ACT_NONE, true); break;
    }
  } else {
    switch (p.act) {
      case WD_ACT_RELU: WD_OCT(WD_ACT_RELU, false); break;
      case WD_ACT_SILU: WD_OCT(WD_ACT_SILU, false); break;
      case WD_ACT_GELU: WD_OCT(WD_ACT_GELU, false); break;
      default: WD_OCT(WD_ACT_NONE, false); break;
    }
  }
#undef WD_OCT
}

}  // namespace
