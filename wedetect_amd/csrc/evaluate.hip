// evaluate.hip — proposal-recall matching on the device (eval_recall/recall.py:
// bbox_overlaps 6-67, _recalls 70-100).  Per image and per proposal budget k: the fp32 IoU matrix
// gts x first-k proposals, then the greedy one-to-one assignment of the reference — repeatedly take
// the largest remaining IoU (ties: lowest gt index, then lowest proposal index, which is what the
// reference's two chained argmax calls select), record it for that gt, strike its row and column.
// One workgroup per (image, budget); the matrix lives in a global scratch slice.  IoU arithmetic is
// done with the reference's operation order and without contraction, so results are bit-identical.
#include "common.h"

namespace {

__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ void __launch_bounds__(256) recall_match_kernel(const float* __restrict__ gts, const int* __restrict__ gt_off,
                                                           const float* __restrict__ props, const int* __restrict__ prop_off,
                                                           const int* __restrict__ budgets, int n_budget,
                                                           float* __restrict__ scratch, long long scratch_stride,
                                                           float* __restrict__ out, int total_gt, float extra, float eps) {
#pragma clang fp contract(off)
  const int img = blockIdx.x / n_budget, kb = blockIdx.x % n_budget;
  const int g0 = gt_off[img], ng = gt_off[img + 1] - g0;
  const int p0 = prop_off[img], np_all = prop_off[img + 1] - p0;
  const int np = np_all < budgets[kb] ? np_all : budgets[kb];
  float* o = out + (size_t)kb * total_gt + g0;
  const int t = threadIdx.x;
  if (ng == 0) return;
  if (np == 0) {
    for (int j = t; j < ng; j += 256) o[j] = 0.0f;
    return;
  }
  float* mat = scratch + (size_t)blockIdx.x * scratch_stride;       // [ng][np]
  for (int e = t; e < ng * np; e += 256) {
    const int gi = e / np, pi = e - gi * np;
    const float* a = gts + (size_t)(g0 + gi) * 4;
    const float* b = props + (size_t)(p0 + pi) * 4;
    // plain operators under "fp contract(off)": the __f*_rn intrinsics are header functions compiled with
    // the default contraction and get fused (union = fma(-w, h, a + b)), which is not what numpy computes
    const float area_a = (a[2] - a[0] + extra) * (a[3] - a[1] + extra);
    const float area_b = (b[2] - b[0] + extra) * (b[3] - b[1] + extra);
    const float xs = fmaxf(a[0], b[0]), ys = fmaxf(a[1], b[1]), xe = fminf(a[2], b[2]), ye = fminf(a[3], b[3]);
    const float ov = fmaxf(xe - xs + extra, 0.0f) * fmaxf(ye - ys + extra, 0.0f);
    const float un = fmaxf(area_a + area_b - ov, eps);
    mat[e] = ov / un;
  }
  __syncthreads();
  __shared__ unsigned long long red[256];
  for (int it = 0; it < ng; ++it) {
    unsigned long long best = 0ull;
    for (int e = t; e < ng * np; e += 256) {
      const int gi = e / np, pi = e - gi * np;
      const unsigned long long key = ((unsigned long long)ordered_bits(mat[e]) << 32) |
                                     ((unsigned long long)(0xFFFFu - (unsigned)gi) << 16) | (0xFFFFu - (unsigned)pi);
      best = key > best ? key : best;
    }
    red[t] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) red[t] = red[t + s] > red[t] ? red[t + s] : red[t];
      __syncthreads();
    }
    const unsigned long long k = red[0];
    __syncthreads();
    const int gi = 0xFFFF - (int)((k >> 16) & 0xFFFFu), pi = 0xFFFF - (int)(k & 0xFFFFu);
    if (t == 0) o[it] = mat[gi * np + pi];
    __syncthreads();
    for (int e = t; e < np; e += 256) mat[gi * np + e] = -1.0f;
    for (int e = t; e < ng; e += 256) mat[e * np + pi] = -1.0f;
    __syncthreads();
  }
}

}  // namespace

extern "C" int64_t wd_recall_scratch_floats(int32_t max_gt, int32_t max_budget) {
  return (int64_t)max_gt * max_budget;
}

extern "C" int wd_recall_match(const float* gts, const int32_t* gt_off, const float* props, const int32_t* prop_off,
                               int32_t n_img, const int32_t* budgets, int32_t n_budget, float* scratch,
                               int64_t scratch_floats_per_block, float* out, int32_t total_gt, int32_t legacy, void* stream) {
  if (!gts || !gt_off || !props || !prop_off || !budgets || !scratch || !out) return WD_ERR_BAD_ARG;
  if (n_img <= 0 || n_budget <= 0 || total_gt < 0 || scratch_floats_per_block <= 0) return WD_ERR_BAD_ARG;
  hipLaunchKernelGGL(recall_match_kernel, dim3((unsigned)(n_img * n_budget)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gts, gt_off, props, prop_off, budgets, n_budget, scratch, (long long)scratch_floats_per_block, out,
                     total_gt, legacy ? 1.0f : 0.0f, 1e-6f);
  return wd_launch_status();
}
