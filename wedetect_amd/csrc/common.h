// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wedetect_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WD_WAVE 64

static inline int wd_launch_status() {
  return hipGetLastError() == hipSuccess ? WD_OK : WD_ERR_LAUNCH;
}

static inline bool wd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wd_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wd_act(float v, int act) {
  switch (act) {
    case WD_ACT_RELU: return fmaxf(v, 0.0f);
    case WD_ACT_SILU: return v / (1.0f + expf(-v));
    case WD_ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
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
