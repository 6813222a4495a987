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


// ---------------------------------------------------------------------------------------------------
// OpenCV-style 8-bit resize (the mmdet test pipeline: WeDetectKeepRatioResize -> cv2.resize INTER_AREA
// when shrinking / INTER_LINEAR when enlarging, transforms.py:94-123) fused with the letter pad
// (WeDetectLetterResize, transforms.py:180-272): one thread per canvas pixel, 3 channels.
// The float accumulation of the general area path keeps OpenCV's operation order (per source row:
// buf = sum_k S*alpha left to right; sum = beta*buf, then sum += beta*buf), multiply and add rounded
// separately — no fma contraction.
// ---------------------------------------------------------------------------------------------------
struct CvResizeArgs {
  const unsigned char* src; int sh, sw;
  unsigned char* dst; int dst_h, dst_w, new_h, new_w, top, left, fill, swap_rb;
  const int* xa; const int* xidx; const float* xw;      // per-axis tables, see wd_cv_resize_paste_u8
  const int* ya; const int* yidx; const float* yw;
  int p0, p1; float p2;
};

__device__ __forceinline__ unsigned char sat_round_u8(float v) {
  const int r = __float2int_rn(v);                      // cvRound: half to even
  return (unsigned char)(r < 0 ? 0 : (r > 255 ? 255 : r));
}

template <int MODE>
__global__ void __launch_bounds__(256) cv_resize_paste_kernel(const CvResizeArgs a) {
#pragma clang fp contract(off)
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)a.dst_h * a.dst_w) return;
  const int Y = (int)(idx / a.dst_w), X = (int)(idx % a.dst_w);
  const int dy = Y - a.top, dx = X - a.left;
  unsigned char* o = a.dst + (size_t)idx * 3;
  if ((unsigned)dy >= (unsigned)a.new_h || (unsigned)dx >= (unsigned)a.new_w) {
    o[0] = o[1] = o[2] = (unsigned char)a.fill;
    return;
  }
  unsigned char r0, r1, r2;
  if (MODE == WD_CVRESIZE_COPY) {
    const unsigned char* s = a.src + ((size_t)dy * a.sw + dx) * 3;
    r0 = s[0]; r1 = s[1]; r2 = s[2];
  } else if (MODE == WD_CVRESIZE_AREA_FAST) {
    const int isx = a.p0, isy = a.p1;
    int s0 = 0, s1 = 0, s2 = 0;
    for (int y = 0; y < isy; ++y) {
      const unsigned char* s = a.src + ((size_t)(dy * isy + y) * a.sw + (size_t)dx * isx) * 3;
      for (int x = 0; x < isx; ++x) { s0 += s[3 * x]; s1 += s[3 * x + 1]; s2 += s[3 * x + 2]; }
    }
    if (isx == 2 && isy == 2) {
      r0 = (unsigned char)((s0 + 2) >> 2); r1 = (unsigned char)((s1 + 2) >> 2); r2 = (unsigned char)((s2 + 2) >> 2);
    } else {
      r0 = sat_round_u8((float)s0 * a.p2); r1 = sat_round_u8((float)s1 * a.p2); r2 = sat_round_u8((float)s2 * a.p2);
    }
  } else if (MODE == WD_CVRESIZE_AREA) {
    const int xs = a.xa[2 * dx], xn = a.xa[2 * dx + 1];
    const int ys = a.ya[2 * dy], yn = a.ya[2 * dy + 1];
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < yn; ++j) {
      const unsigned char* row = a.src + (size_t)a.yidx[ys + j] * a.sw * 3;
      const float beta = a.yw[ys + j];
      float b0 = 0.f, b1 = 0.f, b2 = 0.f;
      for (int k = 0; k < xn; ++k) {
        const unsigned char* s = row + (size_t)a.xidx[xs + k] * 3;
        const float al = a.xw[xs + k];
        b0 = b0 + (float)s[0] * al; b1 = b1 + (float)s[1] * al; b2 = b2 + (float)s[2] * al;
      }
      if (j == 0) { t0 = beta * b0; t1 = beta * b1; t2 = beta * b2; }
      else { t0 = t0 + beta * b0; t1 = t1 + beta * b1; t2 = t2 + beta * b2; }
    }
    r0 = sat_round_u8(t0); r1 = sat_round_u8(t1); r2 = sat_round_u8(t2);
  } else {                                               // WD_CVRESIZE_LINEAR: 11-bit fixed point
    const int sx = a.xidx[dx], a0 = a.xa[2 * dx], a1 = a.xa[2 * dx + 1];
    const bool two = dx < a.p0;                          // p0 = xmax: from there on single tap * 2048
    int sy0 = a.yidx[dy], sy1 = sy0 + 1;
    sy0 = sy0 < 0 ? 0 : (sy0 < a.sh ? sy0 : a.sh - 1);
    sy1 = sy1 < 0 ? 0 : (sy1 < a.sh ? sy1 : a.sh - 1);
    const int b0 = a.ya[2 * dy], b1 = a.ya[2 * dy + 1];
    const unsigned char* p = a.src + ((size_t)sy0 * a.sw + sx) * 3;
    const unsigned char* q = a.src + ((size_t)sy1 * a.sw + sx) * 3;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = two ? p[c] * a0 + p[c + 3] * a1 : p[c] * 2048;
      const int h1 = two ? q[c] * a0 + q[c + 3] * a1 : q[c] * 2048;
      v[c] = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    }
    r0 = (unsigned char)v[0]; r1 = (unsigned char)v[1]; r2 = (unsigned char)v[2];
  }
  if (a.swap_rb) { o[0] = r2; o[1] = r1; o[2] = r0; }
  else { o[0] = r0; o[1] = r1; o[2] = r2; }
}

// [B, 3, H, W] (uint8 or fp32 0..255, channel order c2 c1 c0) -> [B, H, W, 3] uint8 (c0 c1 c2):
// what DetDataPreprocessor's bgr_to_rgb + the NHWC stem need from a packed mmdet batch.
template <typename T>
__global__ void __launch_bounds__(256) chw_to_hwc_swap_kernel(const T* __restrict__ src, unsigned char* __restrict__ dst,
                                                              long long hw, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long b = idx / hw, p = idx % hw;
  const T* s = src + b * 3 * hw + p;
  unsigned char* o = dst + idx * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const T v = s[(2 - c) * hw];
    if constexpr (sizeof(T) == 1) o[c] = (unsigned char)v;
    else { const float r = rintf((float)v); o[c] = (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r)); }
  }
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

extern "C" int wd_cv_resize_paste_u8(const uint8_t* src, int32_t sh, int32_t sw, int32_t mode, const int32_t* xa,
                                     const int32_t* xidx, const float* xw, const int32_t* ya, const int32_t* yidx,
                                     const float* yw, int32_t p0, int32_t p1, float p2, uint8_t* dst, int32_t dst_h,
                                     int32_t dst_w, int32_t new_h, int32_t new_w, int32_t top, int32_t left,
                                     int32_t fill, int32_t swap_rb, void* stream) {
  if (!src || !dst || sh <= 0 || sw <= 0 || dst_h <= 0 || dst_w <= 0 || new_h <= 0 || new_w <= 0) return WD_ERR_BAD_ARG;
  if (left < 0 || top < 0 || left + new_w > dst_w || top + new_h > dst_h || fill < 0 || fill > 255) return WD_ERR_BAD_ARG;
  switch (mode) {
    case WD_CVRESIZE_COPY:
      if (new_h != sh || new_w != sw) return WD_ERR_BAD_ARG;
      break;
    case WD_CVRESIZE_AREA_FAST:
      if (p0 < 1 || p1 < 1 || (long long)new_w * p0 > sw || (long long)new_h * p1 > sh) return WD_ERR_BAD_ARG;
      break;
    case WD_CVRESIZE_AREA:
      if (!xa || !xidx || !xw || !ya || !yidx || !yw) return WD_ERR_BAD_ARG;
      break;
    case WD_CVRESIZE_LINEAR:
      if (!xa || !xidx || !ya || !yidx || p0 < 0 || p0 > new_w) return WD_ERR_BAD_ARG;
      break;
    default:
      return WD_ERR_BAD_ARG;
  }
  const long long n = (long long)dst_h * dst_w;
  if (n > 0x7fffffffLL * 256) return WD_ERR_BAD_ARG;
  CvResizeArgs a{src, sh, sw, dst, dst_h, dst_w, new_h, new_w, top, left, fill, swap_rb, xa, xidx, xw, ya, yidx, yw, p0, p1, p2};
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (mode) {
    case WD_CVRESIZE_COPY: hipLaunchKernelGGL(cv_resize_paste_kernel<WD_CVRESIZE_COPY>, grid, block, 0, st, a); break;
    case WD_CVRESIZE_AREA_FAST: hipLaunchKernelGGL(cv_resize_paste_kernel<WD_CVRESIZE_AREA_FAST>, grid, block, 0, st, a); break;
    case WD_CVRESIZE_AREA: hipLaunchKernelGGL(cv_resize_paste_kernel<WD_CVRESIZE_AREA>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(cv_resize_paste_kernel<WD_CVRESIZE_LINEAR>, grid, block, 0, st, a); break;
  }
  return wd_launch_status();
}

extern "C" int wd_chw_to_hwc_u8(const void* src, int32_t src_is_f32, uint8_t* dst, int32_t batch, int32_t h, int32_t w,
                                void* stream) {
  if (!src || !dst || batch <= 0 || h <= 0 || w <= 0) return WD_ERR_BAD_ARG;
  const long long hw = (long long)h * w, total = hw * batch;
  if (total > 0x7fffffffLL * 256) return WD_ERR_BAD_ARG;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (src_is_f32)
    hipLaunchKernelGGL(chw_to_hwc_swap_kernel<float>, grid, block, 0, st, static_cast<const float*>(src), dst, hw, total);
  else
    hipLaunchKernelGGL(chw_to_hwc_swap_kernel<unsigned char>, grid, block, 0, st, static_cast<const unsigned char*>(src), dst,
                       hw, total);
  return wd_launch_status();
}
