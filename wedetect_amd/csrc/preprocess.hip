// preprocess.hip — device-side letterbox: Pillow-exact antialiased bilinear resize of a uint8 RGB
// image (two separable int32 fixed-point passes, 22 fractional bits, uint8 rounding between the
// passes as in Pillow's src/libImaging/Resample.c) + paste onto a constant-colour canvas.
// Replaces the per-image host work of generate_proposal.py:17-82 (letterbox) before the tower.
// HBM-bound byte work: one thread per output pixel, 3 channels, coalesced along x.
#include "common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;

__device__ __forceinline__ unsigned char clip8(int v) {
  v >>= kPrecisionBits;                       // arithmetic shift = Pillow's table index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// tmp[y][xx][c] = clip8(2^21 + sum_x src[y][xmin + x][c] * k[xx][x])
__global__ void __launch_bounds__(256) resample_h_kernel(const unsigned char* __restrict__ src, int h, int w,
                                                         const int* __restrict__ bounds, const int* __restrict__ kk,
                                                         int ksize, unsigned char* __restrict__ tmp, int new_w) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)h * new_w) return;
  const int y = (int)(idx / new_w), xx = (int)(idx % new_w);
  const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
  const int* k = kk + (size_t)xx * ksize;
  const unsigned char* row = src + ((size_t)y * w + xmin) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
    s0 += row[3 * x] * kv;
    s1 += row[3 * x + 1] * kv;
    s2 += row[3 * x + 2] * kv;
  }
  unsigned char* o = tmp + (size_t)idx * 3;
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

// dst[Y][X][c] = inside the pasted rectangle ? clip8(2^21 + sum_y tmp[ymin + y][X - left][c] * k[Y - top][y]) : fill
__global__ void __launch_bounds__(256) resample_v_paste_kernel(const unsigned char* __restrict__ tmp, int new_w, int new_h,
                                                               const int* __restrict__ bounds,
                                                               const int* __restrict__ kk, int ksize,
                                                               unsigned char* __restrict__ dst, int dst_h, int dst_w,
                                                               int left, int top, int f0, int f1, int f2) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)dst_h * dst_w) return;
  const int Y = (int)(idx / dst_w), X = (int)(idx % dst_w);
  const int yy = Y - top, xx = X - left;
  unsigned char* o = dst + (size_t)idx * 3;
  if ((unsigned)yy >= (unsigned)new_h || (unsigned)xx >= (unsigned)new_w) {
    o[0] = (unsigned char)f0; o[1] = (unsigned char)f1; o[2] = (unsigned char)f2;
    return;
  }
  const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
  const int* k = kk + (size_t)yy * ksize;
  const unsigned char* col = tmp + ((size_t)ymin * new_w + xx) * 3;
  int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
  for (int y = 0; y < n; ++y) {
    const int kv = k[y];
    const unsigned char* px = col + (size_t)y * new_w * 3;
    s0 += px[0] * kv;
    s1 += px[1] * kv;
    s2 += px[2] * kv;
  }
  o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
}

}  // namespace

extern "C" int wd_letterbox_u8(const uint8_t* src, int32_t h, int32_t w, const int32_t* bounds_h, const int32_t* kk_h,
                               int32_t ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int32_t ksize_v,
                               uint8_t* tmp, uint8_t* dst, int32_t dst_h, int32_t dst_w, int32_t new_w, int32_t new_h,
                               int32_t left, int32_t top, int32_t fill_r, int32_t fill_g, int32_t fill_b, void* stream) {
  if (!src || !bounds_h || !kk_h || !bounds_v || !kk_v || !tmp || !dst) return WD_ERR_BAD_ARG;
  if (h <= 0 || w <= 0 || new_w <= 0 || new_h <= 0 || dst_h <= 0 || dst_w <= 0 || ksize_h <= 0 || ksize_v <= 0)
    return WD_ERR_BAD_ARG;
  if (left < 0 || top < 0 || left + new_w > dst_w || top + new_h > dst_h) return WD_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long n1 = (long long)h * new_w, n2 = (long long)dst_h * dst_w;
  if (n1 > 0x7fffffffLL * 256 || n2 > 0x7fffffffLL * 256) return WD_ERR_BAD_ARG;
  hipLaunchKernelGGL(resample_h_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, src, h, w, bounds_h, kk_h,
                     ksize_h, tmp, new_w);
  hipLaunchKernelGGL(resample_v_paste_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, tmp, new_w, new_h,
                     bounds_v, kk_v, ksize_v, dst, dst_h, dst_w, left, top, fill_r, fill_g, fill_b);
  return wd_launch_status();
}
