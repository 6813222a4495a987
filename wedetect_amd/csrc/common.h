// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "wedetect_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WD_WAVE 64

static inline int wd_launch_status() {
  return hipGetLastError() == hipSuccess ? WD_OK : WD_ERR_LAUNCH;
}

// wd_time_next_gemm (abi.hip): events the next GEMM launch of this thread stamps with its own begin / end
struct WdLaunchTiming { hipEvent_t start, stop; };
extern thread_local WdLaunchTiming wd_launch_timing;

// every GEMM kernel launch goes through this: a plain launch, or — once, after wd_time_next_gemm — one whose
// dispatch carries the two events (hipExtLaunchKernelGGL: kernel begin / end timestamps, no barrier packets)
#define WD_LAUNCH_GEMM(kern, grid, block, lds, st, ...)                                      \
  do {                                                                                       \
    const WdLaunchTiming wd_t_ = wd_launch_timing;                                           \
    if (wd_t_.start || wd_t_.stop) {                                                         \
      wd_launch_timing = WdLaunchTiming{nullptr, nullptr};                                   \
      hipExtLaunchKernelGGL(kern, grid, block, lds, st, wd_t_.start, wd_t_.stop, 0, __VA_ARGS__); \
    } else {                                                                                 \
      hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                           \
    }                                                                                        \
  } while (0)

// 1 iff any of the values is inf or NaN: x * 0 is NaN exactly then, and a NaN survives the sum (two packed FMAs and
// one compare per four values instead of an and + compare per value)
__device__ __forceinline__ bool wd_any_nonfinite4(float a, float b, float c, float d) {
  float z = 0.0f;
  z = fmaf(a, 0.0f, z); z = fmaf(b, 0.0f, z); z = fmaf(c, 0.0f, z); z = fmaf(d, 0.0f, z);
  return z != z;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE setting of a kernel: remember it per device id, so a process
// that drives a second GPU sets it there too (one bit per device; a lost race only repeats an idempotent call)
struct WdAttrOnce { unsigned long long done = 0; };
static inline int wd_set_max_lds(WdAttrOnce& once, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return WD_ERR_LAUNCH;
  const unsigned long long bit = 1ull << (dev & 63);
  if (once.done & bit) return WD_OK;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return WD_ERR_LAUNCH;
  once.done |= bit;
  return WD_OK;
}

static inline bool wd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// GEMM-epilogue variant: hardware exp2 / rcp (<= 2 ulp, i.e. <= 1.2e-7 absolute on a score), a third of the
// instructions of the exact form — the similarity GEMM's epilogue is 20 sigmoids per lane
__device__ __forceinline__ float wd_sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// Exact-erf GELU, 0.5*x*(1+erf(x/sqrt2)), with erfc from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute on erfc,
// i.e. <= 1e-7*|x| on GELU — three orders below the 1e-3 parity budget) instead of the ~3x longer libdevice erff: the GELU
// epilogue was 11 % of the pwconv1 GEMMs.  The reciprocal is the hardware v_rcp_f32 (1 ulp), the exponential v_exp_f32.
// Round 4: 13 instructions instead of 19 (the fused 128-channel block MLP is VALU-bound on this function, the pwconv1 launches
// spend ~60 us of 370 in it).  With E = erfc(|x| / sqrt2) / 2 the two branches of the old form — 0.5 x (2 - 2E) for x >= 0,
// 0.5 x (2E) below — are ONE expression:  GELU(x) = max(x, 0) - |x| E,  so the compare / select / subtract go; the 0.5 is
// folded into the polynomial's coefficients, 1/sqrt2 into p, and exp(-x^2 / 2) = exp2(-(k x)^2) with k = sqrt(log2(e) / 2)
// costs two multiplies.  Same approximation, same accuracy (max |error| 3.3e-7 against 4.2e-7 before, on 4 M points of
// [-12, 12]; the two forms differ by at most 4.8e-7, i.e. by their last bits).  Non-finite inputs stay non-finite (+inf now
// gives NaN instead of inf: the fp16x3 range guard tests "not finite").
#ifdef WD_GELU_R3     // the 19-instruction form of rounds 2-3, for A/B builds only (scripts: profiles/r04_gelu_ab.txt)
__device__ __forceinline__ float wd_gelu(float x) {
#pragma clang fp contract(off)
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  const float erfc_abs = y * t * __expf(-ax * ax);
  const float one_plus_erf = x >= 0.f ? 2.0f - erfc_abs : erfc_abs;
  return 0.5f * x * one_plus_erf;
}
#elif defined(WD_GELU_R4)   // the 13-instruction Abramowitz-Stegun form of round 4, for A/B builds only (profiles/r05_gelu_ab.txt)
__device__ __forceinline__ float wd_gelu(float x) {
  // Every fused multiply-add of this function is written as fmaf; nothing else may be contracted.  The GEMM epilogues of
  // different kernels must produce the SAME bits for the same accumulator (pre-split and loader-split paths are compared bit
  // for bit), and an optional contraction is decided per inlining context by the compiler.
#pragma clang fp contract(off)
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, ax, 1.0f));          // 1 / (1 + p |x| / sqrt2), p = 0.3275911
  float y = fmaf(0.5307027145f, t, -0.7265760135f);                             // A&S coefficients a5 .. a1, halved
  y = fmaf(y, t, 0.7107068705f);
  y = fmaf(y, t, -0.142248368f);
  y = fmaf(y, t, 0.127414796f);
  const float xs = x * 0.84932180f;                                             // sqrt(log2(e) / 2)
  const float e = __builtin_amdgcn_exp2f(-(xs * xs));                           // exp(-x^2 / 2)
  const float half_erfc = (y * t) * e;                                          // erfc(|x| / sqrt2) / 2
  return fmaf(-ax, half_erfc, fmaxf(x, 0.0f));
}
#else
// Round 5: 8 instructions, ONE transcendental.  GELU(x) = max(x, 0) - |x| Phi(-|x|), and log2 Phi(-a) is so nearly a parabola
// that a degree-5 polynomial in a reproduces a Phi(-a) to 5.5e-7 ABSOLUTE over the whole line (weighted minimax fit,
// scripts/fit_gelu.py; the A-S form: 3.3e-7; the parity budget: 1e-3):  Phi(-a) = exp2(P5(a)),  P5(0) = -1 exactly, so
// GELU(x) -> x / 2 with a RELATIVE error below 1e-6 as x -> 0.  Five FMAs, v_exp_f32, v_max, one FMA — the reciprocal, the second
// polynomial and four multiplies of the round-4 form are gone (19 -> 11 issue slots per value with the transcendentals at
// quarter rate; the GELU + split epilogue of a pwconv1 tile is VALU-bound, the fused block MLP of stage 1 too).  The leading
// coefficient is negative: P5 -> -inf for large |x|, exp2 -> 0, and GELU(x) = max(x, 0) exactly there; x = +-inf gives NaN
// (inf * 0), NaN stays NaN: non-finite inputs stay non-finite for the fp16x3 range guard, as before.
__device__ __forceinline__ float wd_gelu(float x) {
  // every operation written out (fmaf, no optional contraction): all kernels must produce the SAME bits for the same accumulator
#pragma clang fp contract(off)
  const float ax = fabsf(x);
  float p = fmaf(-4.8865197459e-04f, ax, 7.2031705640e-03f);
  p = fmaf(p, ax, -5.2158899605e-02f);
  p = fmaf(p, ax, -4.5958217978e-01f);
  p = fmaf(p, ax, -1.1510057449e+00f);
  p = fmaf(p, ax, -1.0f);
  const float phi = __builtin_amdgcn_exp2f(p);                                  // Phi(-|x|)
  return fmaf(-ax, phi, fmaxf(x, 0.0f));
}
#endif

__device__ __forceinline__ float wd_act(float v, int act) {
  switch (act) {
    case WD_ACT_RELU: return fmaxf(v, 0.0f);
    case WD_ACT_SILU: return v / (1.0f + expf(-v));
    case WD_ACT_GELU: return wd_gelu(v);
    default: return v;
  }
}

// xor-butterfly all-reduce over a power-of-two lane group (width <= 64)
__device__ __forceinline__ float wd_group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wd_group_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
