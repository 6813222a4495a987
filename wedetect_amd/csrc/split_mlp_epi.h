// split_mlp_epi.h — the hidden-activation epilogue of the fused ConvNeXt block MLP kernels (split_gemm_mlp.hip: the
// 128-channel stage, hidden chunk in registers; split_gemm_mlpw.hip: the 256 / 512-channel stages, hidden chunk through
// LDS): bias + GELU + range scale + fp16 hi/lo split on a GEMM-1 accumulator where it lies, and the lane exchange that
// turns the accumulator layout into GEMM 2's operand layout.  Element for element the arithmetic of the two-kernel
// form's WD_SPLIT_C epilogue (epi_lds_tile_csplit, split_gemm_impl.h): the fused kernels are bit-identical to it.
#pragma once
#include "split_gemm_impl.h"

namespace {

// one group of the epilogue: 4 channels of a pixel (accumulator registers 4 g .. 4 g + 3): bias, GELU, range scale, hi/lo
// split.  One value at a time, scalar instructions only — v_pk_{fma,mul,add}_f32 do not overlap with MFMAs at all
// (scripts/issue_probe.py: an MFMA hides six v_fma_f32 behind it, four v_pk_fma_f32 cost their full 18 cycles on top).  No
// range check here: a non-finite accumulator or a hidden value beyond the fp16 range becomes an inf / NaN half, then a NaN
// output accumulator, which the final epilogue reports (GELU maps every non-finite input to inf / NaN).
__device__ __forceinline__ void mc_epi_group(const f32x16& hid, int g, const f32x4 bq, float unscale1, float hid_scale, u32x2& hi, u32x2& lo) {
#pragma clang fp contract(off)
  typedef _Float16 ph2 __attribute__((ext_vector_type(2)));
  _Float16 h[4], l[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float o = wd_gelu(fmaf(hid[4 * g + r], unscale1, bq[r])) * hid_scale;   // scale: a power of two, 1 = none — exact either
    asm("" : "+v"(o));                               // way, and no branch in the region; opaque to the SLP vectoriser
    h[r] = (_Float16)o;
    const float res = o - (float)h[r];               // the residual of the ROUNDED fp32 value, as split4 takes it
    l[r] = (_Float16)res;
  }
  hi = u32x2{__builtin_bit_cast(unsigned, ph2{h[0], h[1]}), __builtin_bit_cast(unsigned, ph2{h[2], h[3]})};
  lo = u32x2{__builtin_bit_cast(unsigned, ph2{l[0], l[1]}), __builtin_bit_cast(unsigned, ph2{l[2], l[3]})};
}
// Groups 2 s and 2 s + 1 -> the operand fragments of k16 step s, without touching LDS.  The accumulator gives a lane
// (pixel p, half h) channels 8 g + 4 h + 0..3; the operand of k16 step s wants k = 16 s + 8 h + 0..7: lane (p, 0) keeps its
// group 2 s and takes lane (p, 1)'s group 2 s; lane (p, 1) takes lane (p, 0)'s group 2 s + 1 and keeps its own —
// v_permlane32_swap(X = group 2 s, Y = group 2 s + 1) swaps X's upper 32 lanes with Y's lower 32: fragment = [X' | Y'].
__device__ __forceinline__ void mc_epi_swap(const u32x2 (&hi)[4], const u32x2 (&lo)[4], int s, h8& fh8, h8& fl8) {
  u32x4 fh, fl;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const auto a = __builtin_amdgcn_permlane32_swap(hi[2 * s][d], hi[2 * s + 1][d], false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(lo[2 * s][d], lo[2 * s + 1][d], false, false);
    fh[d] = a[0]; fh[2 + d] = a[1];
    fl[d] = b[0]; fl[2 + d] = b[1];
  }
  fh8 = __builtin_bit_cast(h8, fh);
  fl8 = __builtin_bit_cast(h8, fl);
}

}  // namespace
