// text.hip — the two kernels of the XLM-RoBERTa text tower that are not GEMMs or LayerNorms:
// the embedding sum and the (short-sequence) self-attention.  The tower turns class-name token ids
// into the [K, 768] bank the similarity GEMM consumes (mm_backbone.py:341-390: XLMRobertaModel ->
// last_hidden_state[:, 0] -> Linear head -> L2 normalise); it runs once per vocabulary, so these
// kernels are written for clarity, not for the roofline.  Dense layers reuse wd_conv_gemm(_split),
// LayerNorm wd_layernorm_rows, the final normalisation wd_l2norm_rows.
#include "common.h"

namespace {

// out[t, :] = word[ids[t], :] + pos[pos_ids[t], :] + type0[:]          (RobertaEmbeddings before its LayerNorm)
__global__ void __launch_bounds__(256) text_embed_kernel(const int* __restrict__ ids, const int* __restrict__ pos_ids,
                                                         const float* __restrict__ word, const float* __restrict__ pos,
                                                         const float* __restrict__ type0, float* __restrict__ out,
                                                         long long total4, int dim4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const long long t = i / dim4;
  const int c = (int)(i - t * dim4);
  const f32x4 a = reinterpret_cast<const f32x4*>(word)[(long long)ids[t] * dim4 + c];
  const f32x4 b = reinterpret_cast<const f32x4*>(pos)[(long long)pos_ids[t] * dim4 + c];
  const f32x4 d = reinterpret_cast<const f32x4*>(type0)[c];
  reinterpret_cast<f32x4*>(out)[i] = (a + b) + d;
}

// One wave per (sequence, head), lane = query position (L <= 64, head dim DH <= 64, DH % 4 == 0).
// scores = q . k / sqrt(DH), keys with mask 0 excluded, softmax over keys, out = p . v.
template <int DH>
__global__ void __launch_bounds__(64) attention_small_kernel(const float* __restrict__ qkv, const int* __restrict__ mask,
                                                             float* __restrict__ out, int L, int heads, int ld_qkv,
                                                             int ld_out, float scale) {
  __shared__ __attribute__((aligned(16))) float ks[64 * DH];
  __shared__ __attribute__((aligned(16))) float vs[64 * DH];
  __shared__ int ms[64];
  const int lane = threadIdx.x;
  const int seq = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int hidden = heads * DH;
  const float* base = qkv + (size_t)seq * L * ld_qkv + hd * DH;
  for (int e = lane; e < L * (DH / 4); e += 64) {
    const int j = e / (DH / 4), c = e % (DH / 4);
    reinterpret_cast<f32x4*>(ks)[j * (DH / 4) + c] = *reinterpret_cast<const f32x4*>(base + (size_t)j * ld_qkv + hidden + c * 4);
    reinterpret_cast<f32x4*>(vs)[j * (DH / 4) + c] = *reinterpret_cast<const f32x4*>(base + (size_t)j * ld_qkv + 2 * hidden + c * 4);
  }
  if (lane < L) ms[lane] = mask[seq * L + lane];
  __syncthreads();
  if (lane >= L) return;
  float q[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) q[c] = base[(size_t)lane * ld_qkv + c];
  float mx = -3.0e38f;
  for (int j = 0; j < L; ++j) {
    if (!ms[j]) continue;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) s = fmaf(q[c], ks[j * DH + c], s);
    mx = fmaxf(mx, s * scale);
  }
  float o[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) o[c] = 0.f;
  float den = 0.f;
  for (int j = 0; j < L; ++j) {
    if (!ms[j]) continue;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) s = fmaf(q[c], ks[j * DH + c], s);
    const float pj = expf(s * scale - mx);
    den += pj;
#pragma unroll
    for (int c = 0; c < DH; ++c) o[c] = fmaf(pj, vs[j * DH + c], o[c]);
  }
  const float inv = den > 0.f ? 1.0f / den : 0.f;
  float* op = out + ((size_t)seq * L + lane) * ld_out + hd * DH;
#pragma unroll
  for (int c = 0; c < DH; ++c) op[c] = o[c] * inv;
}

}  // namespace

extern "C" int wd_text_embed(const int32_t* ids, const int32_t* pos_ids, const float* word, const float* pos,
                             const float* type0, float* out, int64_t n_tok, int32_t dim, void* stream) {
  if (!ids || !pos_ids || !word || !pos || !type0 || !out || n_tok <= 0 || dim <= 0 || (dim & 3)) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(word) || !wd_aligned16(pos) || !wd_aligned16(type0) || !wd_aligned16(out)) return WD_ERR_BAD_ARG;
  const long long total4 = (long long)n_tok * (dim / 4);
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), ids, pos_ids, word, pos, type0, out, total4, dim / 4);
  return wd_launch_status();
}

extern "C" int wd_attention_small(const float* qkv, const int32_t* mask, float* out, int32_t n_seq, int32_t seq_len,
                                  int32_t heads, int32_t head_dim, int32_t ld_qkv, int32_t ld_out, void* stream) {
  if (!qkv || !mask || !out || n_seq <= 0 || heads <= 0) return WD_ERR_BAD_ARG;
  if (seq_len <= 0 || seq_len > 64) return WD_ERR_UNSUPPORTED;
  if (ld_qkv < 3 * heads * head_dim || ld_out < heads * head_dim || (ld_qkv & 3) || !wd_aligned16(qkv)) return WD_ERR_BAD_ARG;
  const float scale = 1.0f / sqrtf((float)head_dim);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(n_seq * heads)), block(64);
  switch (head_dim) {
    case 16: hipLaunchKernelGGL(attention_small_kernel<16>, grid, block, 0, st, qkv, mask, out, seq_len, heads, ld_qkv, ld_out, scale); break;
    case 32: hipLaunchKernelGGL(attention_small_kernel<32>, grid, block, 0, st, qkv, mask, out, seq_len, heads, ld_qkv, ld_out, scale); break;
    case 64: hipLaunchKernelGGL(attention_small_kernel<64>, grid, block, 0, st, qkv, mask, out, seq_len, heads, ld_qkv, ld_out, scale); break;
    default: return WD_ERR_UNSUPPORTED;
  }
  return wd_launch_status();
}
