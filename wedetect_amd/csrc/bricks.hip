// bricks.hip — the pieces of the reference's text-guided attention bricks that are not convolutions,
// Linear layers or LayerNorms (wedetect/models/layers/yolo_bricks.py):
//   * MaxSigmoidAttnBlock.forward (214-243): per pixel and head, max over the guide (text) rows of
//     <embed, guide>, / sqrt(head_channels) + bias, sigmoid, * scale; the projected feature map is
//     multiplied by it                                        -> wd_max_sigmoid_attn
//   * ImagePoolingAttentionModule.forward (614-648): AdaptiveMaxPool2d(p x p) of each projected level
//     into one [levels * p * p, E] patch table                -> wd_adaptive_maxpool_nhwc
//     and multi-head attention of the text rows over those patches -> wd_cross_attention_small
// The WeDetect configs ship with mm_neck=False (config/wedetect_*.py:40-41), so none of this is on
// the measured path: the kernels are HBM-bound, one pass over their inputs, written for clarity.
#include "common.h"

namespace {

constexpr int MSA_GUIDE_FLOATS = 4096;   // guide rows staged per pass: 4096 / head_channels

template <int HC>
struct MsaTile {
  static constexpr int THREADS = HC <= 32 ? 256 : (HC == 64 ? 128 : 64);   // = pixels per block, one per thread
  static constexpr int LDE = HC + 4;   // padded LDS row: the per-thread float4 reads of a row are conflict-free
};

// grid (pixel tiles, heads, images); thread = one pixel of one head.  Every global access is a full
// 128-byte row segment shared by HC/4 neighbouring lanes: the embed tile goes through LDS to be
// transposed into per-thread rows, the final scaling of x walks (pixel, channel quad) pairs in memory
// order.  The head's guide rows are staged through LDS in chunks and read back as wave-uniform
// (broadcast) float4s.  Per pixel the dot product is one sequential fma chain over the head's channels.
template <int HC>
__global__ void __launch_bounds__(MsaTile<HC>::THREADS) max_sigmoid_attn_kernel(
    const float* __restrict__ embed, int ld_embed, const float* __restrict__ guide, const float* __restrict__ head_bias,
    const float* __restrict__ head_scale, float* __restrict__ x, int ld_x, int hw, int n_guide, int heads, int oc,
    float inv_sqrt_hc) {
  using T = MsaTile<HC>;
  constexpr int Q = HC / 4;
  __shared__ __attribute__((aligned(16))) float es[T::THREADS * T::LDE];
  __shared__ __attribute__((aligned(16))) float gs[MSA_GUIDE_FLOATS];
  const int img = blockIdx.z, hd = blockIdx.y, tid = threadIdx.x;
  const int pix0 = blockIdx.x * T::THREADS;
  const size_t row0 = (size_t)img * hw + pix0;
  for (int i = tid; i < T::THREADS * Q; i += T::THREADS) {
    const int px = i / Q, q = i % Q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (pix0 + px < hw) v = *reinterpret_cast<const f32x4*>(embed + (row0 + px) * ld_embed + hd * HC + q * 4);
    *reinterpret_cast<f32x4*>(es + px * T::LDE + q * 4) = v;
  }
  __syncthreads();
  float e[HC];
#pragma unroll
  for (int c = 0; c < Q; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(es + tid * T::LDE + c * 4);
    e[4 * c] = v.x; e[4 * c + 1] = v.y; e[4 * c + 2] = v.z; e[4 * c + 3] = v.w;
  }
  constexpr int CHUNK = MSA_GUIDE_FLOATS / HC;
  const float* gbase = guide + (size_t)img * n_guide * heads * HC + hd * HC;
  float mx = -3.402823466e38f;
  for (int n0 = 0; n0 < n_guide; n0 += CHUNK) {
    const int cnt = min(CHUNK, n_guide - n0);
    if (n0) __syncthreads();
    for (int i = tid; i < cnt * Q; i += T::THREADS) {
      const int n = i / Q, c = i % Q;
      reinterpret_cast<f32x4*>(gs)[i] = *reinterpret_cast<const f32x4*>(gbase + (size_t)(n0 + n) * heads * HC + c * 4);
    }
    __syncthreads();
    for (int n = 0; n < cnt; ++n) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < Q; ++c) {
        const f32x4 g = reinterpret_cast<const f32x4*>(gs)[n * Q + c];
        s = fmaf(e[4 * c], g.x, s); s = fmaf(e[4 * c + 1], g.y, s);
        s = fmaf(e[4 * c + 2], g.z, s); s = fmaf(e[4 * c + 3], g.w, s);
      }
      mx = fmaxf(mx, s);
    }
  }
  float a = wd_sigmoid(mx * inv_sqrt_hc + head_bias[hd]);
  if (head_scale) a *= head_scale[hd];
  __syncthreads();                       // everyone is done with gs: reuse it for the per-pixel factors
  gs[tid] = a;
  __syncthreads();
  const int oq = oc / 4;
  for (int i = tid; i < T::THREADS * oq; i += T::THREADS) {
    const int px = i / oq, q = i % oq;
    if (pix0 + px >= hw) break;          // pixels ascend with i
    f32x4* xp = reinterpret_cast<f32x4*>(x + (row0 + px) * ld_x + (size_t)hd * oc + q * 4);
    f32x4 v = *xp;
    v *= gs[px];
    *xp = v;
  }
}

// grid (p * p cells, images), 256 threads: channel quads x window slices (the slices of a quad are max-reduced
// through LDS).  PyTorch's adaptive windows: [floor(i * H / p), ceil((i + 1) * H / p)).
__global__ void __launch_bounds__(256) adaptive_maxpool_kernel(const float* __restrict__ x, int ld_x, float* __restrict__ out,
                                                               int ld_out, long long out_img_stride, int h, int w, int c4,
                                                               int p) {
  __shared__ f32x4 part[256];
  const int cell = blockIdx.x, img = blockIdx.y;
  const int py = cell / p, px = cell % p;
  const int y0 = (py * h) / p, y1 = ((py + 1) * h + p - 1) / p;
  const int x0 = (px * w) / p, x1 = ((px + 1) * w + p - 1) / p;
  const int ww = x1 - x0, npix = (y1 - y0) * ww;
  const int cw = c4 < 256 ? c4 : 256;            // channel quads per pass
  const int nsub = 256 / cw;                     // window slices
  const int sub = threadIdx.x / cw, cq = threadIdx.x % cw;
  for (int c0 = 0; c0 < c4; c0 += cw) {
    const int c = c0 + cq;
    f32x4 m = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
    if (sub < nsub && c < c4)
      for (int i = sub; i < npix; i += nsub) {
        const int yy = y0 + i / ww, xx = x0 + i % ww;
        const f32x4 v = reinterpret_cast<const f32x4*>(x + ((size_t)img * h * w + (size_t)yy * w + xx) * ld_x)[c];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    part[threadIdx.x] = m;
    __syncthreads();
    if (sub == 0 && c < c4) {
      for (int s2 = 1; s2 < nsub; ++s2) {
        const f32x4 v = part[s2 * cw + cq];
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
      reinterpret_cast<f32x4*>(out + (size_t)img * out_img_stride + (size_t)cell * ld_out)[c] = m;
    }
    __syncthreads();
  }
}

// grid (heads, images), 64 threads; lane = query row (strided over n_q), keys / values of the head in LDS.
template <int DH>
__global__ void __launch_bounds__(64) cross_attention_kernel(const float* __restrict__ q, int ld_q, const float* __restrict__ k,
                                                             const float* __restrict__ v, int ld_kv, float* __restrict__ out,
                                                             int ld_out, int n_q, int n_k, float scale) {
  __shared__ __attribute__((aligned(16))) float ks[64 * DH];
  __shared__ __attribute__((aligned(16))) float vs[64 * DH];
  const int hd = blockIdx.x, img = blockIdx.y, lane = threadIdx.x;
  for (int i = lane; i < n_k * (DH / 4); i += 64) {
    const int j = i / (DH / 4), c = i % (DH / 4);
    const size_t off = ((size_t)img * n_k + j) * ld_kv + hd * DH + c * 4;
    reinterpret_cast<f32x4*>(ks)[i] = *reinterpret_cast<const f32x4*>(k + off);
    reinterpret_cast<f32x4*>(vs)[i] = *reinterpret_cast<const f32x4*>(v + off);
  }
  __syncthreads();
  for (int r = lane; r < n_q; r += 64) {
    const float* qp = q + ((size_t)img * n_q + r) * ld_q + hd * DH;
    float qr[DH];
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
      const f32x4 t = reinterpret_cast<const f32x4*>(qp)[c];
      qr[4 * c] = t.x; qr[4 * c + 1] = t.y; qr[4 * c + 2] = t.z; qr[4 * c + 3] = t.w;
    }
    float mx = -3.402823466e38f;
    for (int j = 0; j < n_k; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DH; ++c) s = fmaf(qr[c], ks[j * DH + c], s);
      mx = fmaxf(mx, s * scale);
    }
    float o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = 0.f;
    float den = 0.f;
    for (int j = 0; j < n_k; ++j) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < DH; ++c) s = fmaf(qr[c], ks[j * DH + c], s);
      const float pj = expf(s * scale - mx);
      den += pj;
#pragma unroll
      for (int c = 0; c < DH; ++c) o[c] = fmaf(pj, vs[j * DH + c], o[c]);
    }
    const float inv = 1.0f / den;
    float* op = out + ((size_t)img * n_q + r) * ld_out + hd * DH;
#pragma unroll
    for (int c = 0; c < DH / 4; ++c) {
      const f32x4 t = {o[4 * c] * inv, o[4 * c + 1] * inv, o[4 * c + 2] * inv, o[4 * c + 3] * inv};
      reinterpret_cast<f32x4*>(op)[c] = t;
    }
  }
}

}  // namespace

extern "C" int wd_max_sigmoid_attn(const float* embed, int32_t ld_embed, const float* guide, const float* head_bias,
                                   const float* head_scale, float* x, int32_t ld_x, int32_t n_img, int32_t hw,
                                   int32_t n_guide, int32_t heads, int32_t head_channels, int32_t out_head_channels,
                                   void* stream) {
  if (!embed || !guide || !head_bias || !x || n_img <= 0 || hw <= 0 || n_guide <= 0 || heads <= 0) return WD_ERR_BAD_ARG;
  if (out_head_channels <= 0 || (out_head_channels & 3) || (ld_embed & 3) || (ld_x & 3)) return WD_ERR_BAD_ARG;
  if (ld_embed < heads * head_channels || ld_x < heads * out_head_channels) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(embed) || !wd_aligned16(guide) || !wd_aligned16(x)) return WD_ERR_BAD_ARG;
  if (heads > 65535 || n_img > 65535) return WD_ERR_UNSUPPORTED;
  const float inv = 1.0f / sqrtf((float)head_channels);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define WD_MSA(HC)                                                                                                    \
  case HC: {                                                                                                          \
    constexpr int PIX = MsaTile<HC>::THREADS;                                                                         \
    hipLaunchKernelGGL(max_sigmoid_attn_kernel<HC>, dim3((unsigned)((hw + PIX - 1) / PIX), (unsigned)heads, (unsigned)n_img), \
                       dim3(PIX), 0, st, embed, ld_embed, guide, head_bias, head_scale, x, ld_x, hw, n_guide, heads,   \
                       out_head_channels, inv);                                                                       \
    break;                                                                                                            \
  }
  switch (head_channels) {
    WD_MSA(8) WD_MSA(16) WD_MSA(32) WD_MSA(64) WD_MSA(128)
    default: return WD_ERR_UNSUPPORTED;
  }
#undef WD_MSA
  return wd_launch_status();
}

extern "C" int wd_adaptive_maxpool_nhwc(const float* x, int32_t ld_x, float* out, int32_t ld_out, int64_t out_img_stride,
                                        int32_t n_img, int32_t h, int32_t w, int32_t channels, int32_t pool, void* stream) {
  if (!x || !out || n_img <= 0 || h <= 0 || w <= 0 || channels <= 0 || pool <= 0) return WD_ERR_BAD_ARG;
  if ((channels & 3) || (ld_x & 3) || (ld_out & 3) || (out_img_stride & 3) || ld_x < channels || ld_out < channels)
    return WD_ERR_BAD_ARG;
  if (!wd_aligned16(x) || !wd_aligned16(out)) return WD_ERR_BAD_ARG;
  if (n_img > 65535) return WD_ERR_UNSUPPORTED;
  const int c4 = channels / 4;
  const dim3 grid((unsigned)(pool * pool), (unsigned)n_img), block(256);
  hipLaunchKernelGGL(adaptive_maxpool_kernel, grid, block, 0, static_cast<hipStream_t>(stream), x, ld_x, out, ld_out,
                     (long long)out_img_stride, h, w, c4, pool);
  return wd_launch_status();
}

extern "C" int wd_cross_attention_small(const float* q, int32_t ld_q, const float* k, const float* v, int32_t ld_kv,
                                        float* out, int32_t ld_out, int32_t n_img, int32_t n_q, int32_t n_k, int32_t heads,
                                        int32_t head_dim, void* stream) {
  if (!q || !k || !v || !out || n_img <= 0 || n_q <= 0 || n_k <= 0 || heads <= 0) return WD_ERR_BAD_ARG;
  if (n_k > 64) return WD_ERR_UNSUPPORTED;
  if ((ld_q & 3) || (ld_kv & 3) || (ld_out & 3) || ld_q < heads * head_dim || ld_kv < heads * head_dim ||
      ld_out < heads * head_dim)
    return WD_ERR_BAD_ARG;
  if (!wd_aligned16(q) || !wd_aligned16(k) || !wd_aligned16(v) || !wd_aligned16(out)) return WD_ERR_BAD_ARG;
  if (heads > 65535 || n_img > 65535) return WD_ERR_UNSUPPORTED;
  const float scale = 1.0f / sqrtf((float)head_dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)heads, (unsigned)n_img), block(64);
#define WD_XA(DH)                                                                                                     \
  case DH:                                                                                                            \
    hipLaunchKernelGGL(cross_attention_kernel<DH>, grid, block, 0, st, q, ld_q, k, v, ld_kv, out, ld_out, n_q, n_k, scale); \
    break;
  switch (head_dim) {
    WD_XA(8) WD_XA(16) WD_XA(32) WD_XA(64)
    default: return WD_ERR_UNSUPPORTED;
  }
#undef WD_XA
  return wd_launch_status();
}
