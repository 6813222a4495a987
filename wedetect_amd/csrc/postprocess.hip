// postprocess.hip — score filter / top-k / sort / class-aware NMS / gather for gfx950.
//
// Integer and index work with a bit-exact contract: given identical fp32 scores and boxes
// these kernels reproduce the oracle's candidate order (score desc, flat index asc), its
// nms_pre truncation and its greedy keep decisions exactly.
//
// top-k: 3-level radix select on the raw fp32 bit pattern (scores are positive, so the
// unsigned bit pattern is monotone) -> exact threshold key T, then an ORDERED compaction
// (elements > T, plus the lowest-index elements == T) into 64-bit composite keys
// (~bits << 32 | flat_index), then a bitonic sort of <= 32768 unique keys per image, run as
// LDS-resident 8192-key chunks (one workgroup each) plus three global-stride stages.
// NMS: one wave per image; a 64-candidate chunk is first tested against the kept list
// (LDS), then resolved sequentially inside the wave with ballots.  It stops at max_out
// kept boxes, which equals slicing the full NMS result (greedy NMS is prefix-consistent).
// Three forms (WD_NMS_*): label-test vanilla, torchvision.ops.batched_nms, mmcv.ops.batched_nms —
// the latter two with the libraries' fp32 coordinate offsets and their candidate-count branches.
#include "common.h"

#pragma clang fp contract(off)

namespace {

constexpr int HB = 2048;                 // histogram bins per radix level
constexpr int CH = 4096;                 // elements per workgroup chunk (256 threads x 16)
constexpr int SORT_CHUNK = 8192;          // keys sorted per LDS visit (64 KB)

struct TopkState {                       // one per image, 64 bytes
  unsigned total_valid;                  // elements with score > thr
  unsigned want;                         // min(nms_pre, total_valid)
  unsigned prefix;                       // selected bits so far (pass 0: 11, pass 1: 22, pass 2: 32)
  unsigned remaining;                    // still to take inside the selected bin
  unsigned count_gt;                     // elements with key > T (after pass 2)
  unsigned need_eq;                      // elements == T to take, lowest index first
  unsigned nonfinite;                    // != 0: the score row holds NaN / inf (an upstream overflow: the fp16x3 GEMMs'
                                         // operands must stay below 65504) -> the candidate count is reported as -1
  unsigned pad[9];
};

__device__ __forceinline__ int bin_of(unsigned key, int pass) {
  return pass == 0 ? (int)(key >> 21) : pass == 1 ? (int)((key >> 10) & 0x7FFu) : (int)(key & 0x3FFu);
}
__device__ __forceinline__ bool in_prefix(unsigned key, unsigned prefix, int pass) {
  return pass == 0 ? true : pass == 1 ? (key >> 21) == prefix : (key >> 10) == prefix;
}

// 32-bit fill (own kernel instead of hipMemsetAsync: identical behaviour eagerly and inside a
// captured hipGraph)
__global__ void __launch_bounds__(256) fill_u32_kernel(unsigned* __restrict__ p, unsigned v, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}

// ---- level histogram ------------------------------------------------------------------
template <int PASS>
__global__ void __launch_bounds__(256) topk_hist_kernel(const float* __restrict__ scores, long long n, float thr,
                                                        TopkState* __restrict__ state,
                                                        unsigned* __restrict__ hist) {
  __shared__ unsigned lh[HB];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < HB; i += 256) lh[i] = 0;
  __syncthreads();
  const unsigned prefix = PASS == 0 ? 0u : state[b].prefix;
  if (PASS > 0 && state[b].want == 0) return;
  const float* s = scores + (long long)b * n;
  const long long base = (long long)blockIdx.x * CH;
#pragma unroll 4
  for (int e = 0; e < CH / 256; ++e) {
    const long long i = base + e * 256 + threadIdx.x;
    if (i < n) {
      const float v = s[i];
      const unsigned key = __float_as_uint(v);
      if (PASS == 0 && (key & 0x7F800000u) == 0x7F800000u) state[b].nonfinite = 1u;     // NaN or inf: sticky, benign race
      if (v > thr && in_prefix(key, prefix, PASS)) atomicAdd(&lh[bin_of(key, PASS)], 1u);
    }
  }
  __syncthreads();
  unsigned* gh = hist + ((size_t)b * 3 + PASS) * HB;
  for (int i = threadIdx.x; i < HB; i += 256)
    if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// ---- pick the bin in which the want-th largest element lies ---------------------------
template <int PASS>
__global__ void __launch_bounds__(256) topk_pick_kernel(TopkState* __restrict__ state, const unsigned* __restrict__ hist,
                                                        int nms_pre) {
  __shared__ unsigned part[256];
  __shared__ unsigned tot;
  const int b = blockIdx.x, t = threadIdx.x;
  const unsigned* h = hist + ((size_t)b * 3 + PASS) * HB;
  // thread t owns bins [HB-8t-8, HB-8t) walked from the top
  unsigned loc[8], sum = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { loc[j] = h[HB - 1 - (8 * t + j)]; sum += loc[j]; }
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    unsigned run = 0;
    for (int i = 0; i < 256; ++i) { const unsigned v = part[i]; part[i] = run; run += v; }   // exclusive, top-down
    tot = run;
  }
  __syncthreads();
  TopkState& st = state[b];
  unsigned want;
  if (PASS == 0) {
    want = tot < (unsigned)nms_pre ? tot : (unsigned)nms_pre;
    if (t == 0) { st.total_valid = tot; st.want = want; st.count_gt = 0; }
  } else {
    want = st.remaining;
  }
  __syncthreads();
  if (PASS == 0 ? want == 0 : st.want == 0) {
    if (t == 0) { st.remaining = 0; st.need_eq = 0; st.prefix = 0; }
    return;
  }
  unsigned above = part[t];            // elements in bins above this thread's first bin
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned c = loc[j];
    if (above < want && want <= above + c) {      // exactly one (t, j) satisfies this
      const unsigned bin = (unsigned)(HB - 1 - (8 * t + j));
      const unsigned old_prefix = PASS == 0 ? 0u : st.prefix;
      const unsigned newp = PASS == 0 ? bin : PASS == 1 ? ((old_prefix << 11) | bin) : ((old_prefix << 10) | bin);
      const unsigned cg = (PASS == 0 ? 0u : st.count_gt) + above;
      st.prefix = newp;
      st.remaining = want - above;
      st.count_gt = cg;
      if (PASS == 2) st.need_eq = want - above;
    }
    above += c;
  }
}

// ---- ordered compaction ---------------------------------------------------------------
__global__ void __launch_bounds__(256) topk_count_kernel(const float* __restrict__ scores, long long n, float thr,
                                                         const TopkState* __restrict__ state,
                                                         unsigned* __restrict__ blk_cnt, int nblk) {
  __shared__ unsigned cg, ce;
  const int b = blockIdx.y;
  if (threadIdx.x == 0) { cg = 0; ce = 0; }
  __syncthreads();
  if (state[b].want != 0) {
    const unsigned T = state[b].prefix;
    const float* s = scores + (long long)b * n;
    const long long base = (long long)blockIdx.x * CH + threadIdx.x * 16;
    unsigned g = 0, e = 0;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      const long long i = base + j;
      if (i < n) {
        const float v = s[i];
        const unsigned key = __float_as_uint(v);
        if (v > thr) { g += key > T; e += key == T; }
      }
    }
    if (g) atomicAdd(&cg, g);
    if (e) atomicAdd(&ce, e);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    blk_cnt[((size_t)b * nblk + blockIdx.x) * 2 + 0] = cg;
    blk_cnt[((size_t)b * nblk + blockIdx.x) * 2 + 1] = ce;
  }
}

// exclusive scan of the per-chunk (gt, eq) counts, one workgroup per image
__global__ void __launch_bounds__(256) topk_scan_kernel(unsigned* __restrict__ blk_cnt, int nblk) {
  __shared__ unsigned sg[256], se[256];
  __shared__ unsigned carry_g, carry_e;
  const int b = blockIdx.x, t = threadIdx.x;
  unsigned* c = blk_cnt + (size_t)b * nblk * 2;
  if (t == 0) { carry_g = 0; carry_e = 0; }
  __syncthreads();
  for (int base = 0; base < nblk; base += 256) {
    const int i = base + t;
    const unsigned g = i < nblk ? c[2 * i] : 0u, e = i < nblk ? c[2 * i + 1] : 0u;
    sg[t] = g; se[t] = e;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 entries
    for (int o = 1; o < 256; o <<= 1) {
      const unsigned ag = t >= o ? sg[t - o] : 0u, ae = t >= o ? se[t - o] : 0u;
      __syncthreads();
      sg[t] += ag; se[t] += ae;
      __syncthreads();
    }
    if (i < nblk) { c[2 * i] = carry_g + sg[t] - g; c[2 * i + 1] = carry_e + se[t] - e; }
    __syncthreads();
    if (t == 255) { carry_g += sg[255]; carry_e += se[255]; }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) topk_scatter_kernel(const float* __restrict__ scores, long long n, float thr,
                                                           const TopkState* __restrict__ state,
                                                           const unsigned* __restrict__ blk_cnt, int nblk,
                                                           unsigned long long* __restrict__ keys, int cap) {
  __shared__ unsigned sg[256], se[256];
  const int b = blockIdx.y, t = threadIdx.x;
  if (state[b].want == 0) return;
  const unsigned T = state[b].prefix, count_gt = state[b].count_gt, need_eq = state[b].need_eq;
  const float* s = scores + (long long)b * n;
  const long long base = (long long)blockIdx.x * CH + t * 16;
  unsigned kv[16];
  unsigned g = 0, e = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const long long i = base + j;
    unsigned key = 0;
    bool ok = false;
    if (i < n) {
      const float v = s[i];
      key = __float_as_uint(v);
      ok = v > thr;
    }
    g += ok && key > T;
    e += ok && key == T;
    kv[j] = ok ? key : 0xFFFFFFFFu;        // marker: not a candidate (a valid score key is < 0x7F800000)
  }
  sg[t] = g; se[t] = e;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned ag = t >= o ? sg[t - o] : 0u, ae = t >= o ? se[t - o] : 0u;
    __syncthreads();
    sg[t] += ag; se[t] += ae;
    __syncthreads();
  }
  unsigned pg = blk_cnt[((size_t)b * nblk + blockIdx.x) * 2 + 0] + sg[t] - g;   // ordered rank among "> T"
  unsigned pe = blk_cnt[((size_t)b * nblk + blockIdx.x) * 2 + 1] + se[t] - e;   // ordered rank among "== T"
  unsigned long long* out = keys + (size_t)b * cap;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const unsigned key = kv[j];
    if (key == 0xFFFFFFFFu) continue;
    const unsigned long long ck = ((unsigned long long)(~key) << 32) | (unsigned long long)(unsigned)(base + j);
    if (key > T) {
      out[pg++] = ck;
    } else if (key == T) {
      if (pe < need_eq) out[count_gt + pe] = ck;
      ++pe;
    }
  }
}

// ---- per-image bitonic sort of the composite keys + unpack ----------------------------
// The network is cut into kernels so that every 8192-key chunk (64 KB of LDS) is its own
// workgroup: (1) sort each chunk locally, ascending / descending by global position;
// (2) for every merge wider than a chunk: the strides >= chunk in global memory (one launch
// each), then the strides < chunk again per chunk in LDS.  7 launches for 32768 keys,
// 4 x batch workgroups instead of one workgroup per image.  Entries [want, cap) hold ~0 and
// sort last; n2 = next power of two >= want bounds the work.
__device__ __forceinline__ int sort_n2(unsigned want) {
  int n2 = 1;
  while (n2 < (int)want) n2 <<= 1;
  return n2;
}

__device__ __forceinline__ void lds_bitonic_pass(unsigned long long* sk, int chunk, int base, int jj, int dir_bit) {
  for (int i = threadIdx.x; i < chunk; i += 1024) {
    const int p = i ^ jj;
    if (p > i) {
      const unsigned long long a = sk[i], c = sk[p];
      const bool up = ((base + i) & dir_bit) == 0;
      if ((a > c) == up) { sk[i] = c; sk[p] = a; }
    }
  }
  __syncthreads();
}

// size == 0: phase A (full local sort of the chunk); size > chunk: tail strides of that merge
__global__ void __launch_bounds__(1024) topk_sort_chunk_kernel(unsigned long long* __restrict__ keys, int cap,
                                                               const TopkState* __restrict__ state, int size) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sk[];
  const int b = blockIdx.y, t = threadIdx.x;
  const int n2 = sort_n2(state[b].want);
  const int chunk = n2 < SORT_CHUNK ? n2 : SORT_CHUNK;
  const int base = blockIdx.x * SORT_CHUNK;
  if (base >= n2 || (size != 0 && size > n2)) return;
  unsigned long long* k = keys + (size_t)b * cap + base;
  for (int i = t; i < chunk; i += 1024) sk[i] = k[i];
  __syncthreads();
  if (size == 0) {
    for (int sz = 2; sz <= chunk; sz <<= 1)
      for (int jj = sz >> 1; jj > 0; jj >>= 1) lds_bitonic_pass(sk, chunk, base, jj, sz);
  } else {
    for (int jj = chunk >> 1; jj > 0; jj >>= 1) lds_bitonic_pass(sk, chunk, base, jj, size);
  }
  for (int i = t; i < chunk; i += 1024) k[i] = sk[i];
}

// one global-memory stage (stride j >= SORT_CHUNK) of the merge of width `size`
__global__ void __launch_bounds__(256) topk_sort_global_kernel(unsigned long long* __restrict__ keys, int cap,
                                                               const TopkState* __restrict__ state, int size, int j) {
  const int b = blockIdx.y;
  const int n2 = sort_n2(state[b].want);
  if (size > n2) return;
  unsigned long long* k = keys + (size_t)b * cap;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n2) return;
  const int p = i ^ j;
  if (p > i) {
    const unsigned long long a = k[i], c = k[p];
    const bool up = (i & size) == 0;
    if ((a > c) == up) { k[i] = c; k[p] = a; }
  }
}

__global__ void __launch_bounds__(256) topk_unpack_kernel(const unsigned long long* __restrict__ keys, int cap,
                                                          const TopkState* __restrict__ state,
                                                          int* __restrict__ out_idx, float* __restrict__ out_score,
                                                          int* __restrict__ out_count) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const unsigned want = state[b].want;
  if (i < cap) {
    if (i < (int)want) {
      const unsigned long long ck = keys[(size_t)b * cap + i];
      out_idx[(size_t)b * cap + i] = (int)(unsigned)(ck & 0xFFFFFFFFull);
      out_score[(size_t)b * cap + i] = __uint_as_float(~(unsigned)(ck >> 32));
    } else {
      out_idx[(size_t)b * cap + i] = -1;
      out_score[(size_t)b * cap + i] = 0.f;
    }
  }
  if (i == 0) out_count[b] = state[b].nonfinite ? -1 : (int)want;
}

// ---------------------------------------------------------------------------------------
// NMS
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const f32x4 bi, const f32x4 bj, float thr) {
  // inter / (area_i + area_j - inter) > thr, every operation individually rounded (fp32).  Plain
  // operators under "fp contract(off)": HIP's __fmul_rn / __fsub_rn are header functions compiled with
  // the default contraction, and (area_i + area_j) - w * h written with them is fused into an fma.
#pragma clang fp contract(off)
  const float area_i = (bi[2] - bi[0]) * (bi[3] - bi[1]);
  const float area_j = (bj[2] - bj[0]) * (bj[3] - bj[1]);
  const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
  const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
  const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
  const float inter = w * h;
  const float ovr = inter / (area_i + area_j - inter);
  return ovr > thr;
}

__device__ __forceinline__ f32x4 rescale_box(f32x4 bx, const float* mt) {
  // (box - pad) / scale, the mmdet order (yolo_world_head.py:728-734) and the Uni order
  // (generate_proposal.py:1108-1113) share the arithmetic
#pragma clang fp contract(off)
  f32x4 r;
  r[0] = (bx[0] - mt[0]) / mt[3];
  r[1] = (bx[1] - mt[1]) / mt[4];
  r[2] = (bx[2] - mt[0]) / mt[3];
  r[3] = (bx[3] - mt[1]) / mt[4];
  return r;
}
__device__ __forceinline__ f32x4 clamp_box(f32x4 bx, const float* mt) {
  f32x4 r;
  r[0] = fminf(fmaxf(bx[0], 0.f), mt[5]);
  r[1] = fminf(fmaxf(bx[1], 0.f), mt[6]);
  r[2] = fminf(fmaxf(bx[2], 0.f), mt[5]);
  r[3] = fminf(fmaxf(bx[3], 0.f), mt[6]);
  return r;
}

constexpr int NMS_MAX_OUT = 1024, NMS_BUCKETS = 2048, NMS_MAX_RADIUS = 4;

// The two upstream libraries express class awareness as a coordinate offset, not as a label test
// (torchvision/ops/boxes.py _batched_nms_coordinate_trick, mmcv/ops/nms.py batched_nms):
//     boxes_for_nms = boxes + label * (boxes.max() + 1)          (fp32: quantises the boxes)
// followed by a class-AGNOSTIC greedy pass (torchvision at <= 4000 box coordinates on the CPU, mmcv below
// split_thr = 10000 candidates), or by a per-class loop on the original boxes (torchvision above that) / on the
// offset boxes (mmcv from split_thr).  Which form an image takes depends on its candidate count.
struct NmsForm { bool offset, agnostic; };
__device__ __forceinline__ NmsForm nms_form(int mode, int param, int count) {
  NmsForm f{false, false};
  if (mode == WD_NMS_MMCV) { f.offset = true; f.agnostic = count < param; }
  else if (mode == WD_NMS_TORCHVISION && 4ll * count <= (long long)param) { f.offset = true; f.agnostic = true; }
  return f;
}

// order-preserving float <-> uint map; 0 sorts below every float, so a zeroed word is the identity of atomicMax
__device__ __forceinline__ unsigned ord_enc(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_dec(unsigned e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

// boxes.max() (and -boxes.min(), for the cross-class reach below) over an image's candidates, on the boxes NMS sees
// (after the mmdet-order rescale when meta says so).  bounds[b] = {enc(max), enc(-min)}, zeroed by the caller.
__global__ void __launch_bounds__(256) nms_bounds_kernel(const int* __restrict__ cand_idx, const int* __restrict__ cand_count,
                                                         int cand_stride, const float* __restrict__ boxes, int n_anchor,
                                                         int k, const float* __restrict__ meta, int mode, int param,
                                                         unsigned* __restrict__ bounds) {
  const int b = blockIdx.y;
  const int count = cand_count[b];
  if (count <= 0 || !nms_form(mode, param, count).offset) return;
  const int base = blockIdx.x * 1024;
  if (base >= count) return;
  const float* mt = meta + (size_t)b * 8;
  const bool pre = mt[7] != 0.f;
  const int* ci = cand_idx + (size_t)b * cand_stride;
  const f32x4* bx = reinterpret_cast<const f32x4*>(boxes) + (size_t)b * n_anchor;
  float mx = -INFINITY, mn = INFINITY;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = base + u * 256 + threadIdx.x;
    if (i < count) {
      f32x4 box = bx[ci[i] / k];
      if (pre) box = rescale_box(box, mt);
      mx = fmaxf(mx, fmaxf(fmaxf(box[0], box[1]), fmaxf(box[2], box[3])));
      mn = fminf(mn, fminf(fminf(box[0], box[1]), fminf(box[2], box[3])));
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    mn = fminf(mn, __shfl_xor(mn, o, 64));
  }
  if ((threadIdx.x & 63) == 0 && mx >= mn) {
    atomicMax(bounds + 2 * b, ord_enc(mx));
    atomicMax(bounds + 2 * b + 1, ord_enc(-mn));
  }
}

__global__ void __launch_bounds__(64) nms_kernel(const int* __restrict__ cand_idx, const float* __restrict__ cand_score,
                                                 const int* __restrict__ cand_count, int cand_stride,
                                                 const float* __restrict__ boxes, int n_anchor, int k,
                                                 const float* __restrict__ meta, float iou_thr, int max_out, int mode,
                                                 int param, const unsigned* __restrict__ bounds,
                                                 float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                 int* __restrict__ out_labels, int* __restrict__ out_anchors,
                                                 int* __restrict__ out_count) {
  __shared__ f32x4 kept_box[NMS_MAX_OUT];      // the boxes NMS compares: offset boxes when the form has offsets
  __shared__ int kept_label[NMS_MAX_OUT];
  // kept boxes are chained per class bucket (label & (NMS_BUCKETS - 1)): a candidate only walks the kept boxes of its
  // own bucket — with 80 classes and 300 kept boxes about four entries instead of all 300 (the scan of the whole
  // kept list was most of this kernel when many candidates overlap a kept box and the loop runs over hundreds of chunks)
  __shared__ short kept_next[NMS_MAX_OUT];
  __shared__ short bucket_head[NMS_BUCKETS];
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < NMS_BUCKETS; i += 64) bucket_head[i] = -1;
  __syncthreads();
  const float* mt = meta + (size_t)b * 8;
  const bool pre = mt[7] != 0.f;
  const int count = cand_count[b];
  if (count < 0) {                         // non-finite scores upstream (wd_topk_candidates): nothing to trust in this image
    if (lane == 0) out_count[b] = -1;
    return;
  }
  const NmsForm form = nms_form(mode, param, count);
  // offset step S = boxes.max() + 1 (fp32, as upstream).  The agnostic forms compare a candidate with kept boxes of
  // EVERY class; two offset boxes of classes c < c' can only overlap when (c' - c) * S < (max - min) + rounding of
  // the offsets (fl is monotone: fl(x1' + o_c') < fl(x2 + o_c) needs x1' + o_c' < x2 + o_c), so the walk visits the
  // buckets of label - reach .. label + reach (reach 0 or 1 for boxes that touch the image) and falls back to the
  // whole kept list when the reach is large, S <= 0 (all boxes left of -1) or the threshold is negative.
  float S = 0.f;
  int reach = 0;
  bool whole_list = false;
  if (form.offset && count > 0) {
    const float maxc = ord_dec(bounds[2 * b]), minc = -ord_dec(bounds[2 * b + 1]);
    S = maxc + 1.0f;
    if (form.agnostic) {
      const double sd = (double)S, span = (double)maxc - (double)minc;
      if (!(sd > 0.0) || !(iou_thr >= 0.f)) whole_list = true;
      else {
        const double q = (span + 2.0 * (double)k * sd * 1.1920928955078125e-7) / sd;
        if (!(q < (double)(NMS_MAX_RADIUS + 1))) whole_list = true;
        else reach = (int)q;
      }
    }
  }
  const int* ci = cand_idx + (size_t)b * cand_stride;
  const float* cs = cand_score + (size_t)b * cand_stride;
  const f32x4* bx = reinterpret_cast<const f32x4*>(boxes) + (size_t)b * n_anchor;
  float* ob = out_boxes + (size_t)b * max_out * 4;
  float* os = out_scores + (size_t)b * max_out;
  int* ol = out_labels + (size_t)b * max_out;
  int* oa = out_anchors + (size_t)b * max_out;
  int nk = 0;
  // software-pipelined candidate fetch: the (index -> anchor -> box) dependent loads of chunk
  // c+1 are issued before chunk c is resolved, so their two memory round trips overlap with it
  int n_anchor_ = 0, n_label = -1;
  float n_score = 0.f;
  f32x4 n_box = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int base) {
    const int i = base + lane;
    n_anchor_ = 0; n_label = -1; n_score = 0.f; n_box = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i < count) {
      const int idx = ci[i];
      n_anchor_ = idx / k;
      n_label = idx - n_anchor_ * k;
      n_score = cs[i];
      n_box = bx[n_anchor_];
    }
  };
  fetch(0);
  for (int base = 0; base < count && nk < max_out; base += 64) {
    const bool valid = base + lane < count;
    const int anchor = n_anchor_, label = n_label;
    const float score = n_score;
    f32x4 box = n_box;
    if (base + 64 < count) fetch(base + 64);
    if (valid && pre) box = rescale_box(box, mt);
    f32x4 cmp = box;                       // what NMS compares
    if (form.offset) {
      const float off = (float)label * S;  // idxs.to(boxes) * (max_coordinate + 1), then boxes + offsets[:, None]
      cmp[0] = box[0] + off; cmp[1] = box[1] + off; cmp[2] = box[2] + off; cmp[3] = box[3] + off;
    }
    bool alive = valid;
    // candidates vs. the kept boxes that can suppress them (any order: a candidate dies if ANY of them overlaps it by
    // more than the threshold)
    if (alive) {
      if (!form.agnostic) {
        for (int e = bucket_head[label & (NMS_BUCKETS - 1)]; e >= 0; e = kept_next[e])
          if (kept_label[e] == label && iou_gt(kept_box[e], cmp, iou_thr)) { alive = false; break; }
      } else if (whole_list) {
        for (int e = 0; e < nk; ++e)
          if (iou_gt(kept_box[e], cmp, iou_thr)) { alive = false; break; }
      } else {
        for (int c2 = max(label - reach, 0); alive && c2 <= min(label + reach, k - 1); ++c2)
          for (int e = bucket_head[c2 & (NMS_BUCKETS - 1)]; e >= 0; e = kept_next[e])
            if (kept_label[e] == c2 && iou_gt(kept_box[e], cmp, iou_thr)) { alive = false; break; }
      }
    }
    // sequential resolution inside the chunk, visiting only the candidates that are still alive
    unsigned long long mask = __ballot(alive);
    while (mask != 0ull && nk < max_out) {
      const int p = __ffsll((long long)mask) - 1;          // lowest alive lane = next kept box
      f32x4 bp;
      bp[0] = __shfl(cmp[0], p, 64); bp[1] = __shfl(cmp[1], p, 64);
      bp[2] = __shfl(cmp[2], p, 64); bp[3] = __shfl(cmp[3], p, 64);
      const int lp = __shfl(label, p, 64);
      if (lane == p) {
        kept_box[nk] = cmp;
        kept_label[nk] = label;
        kept_next[nk] = bucket_head[label & (NMS_BUCKETS - 1)];
        bucket_head[label & (NMS_BUCKETS - 1)] = (short)nk;
        f32x4 o = box;
        if (!pre) o = rescale_box(o, mt);
        o = clamp_box(o, mt);
        *reinterpret_cast<f32x4*>(ob + (size_t)nk * 4) = o;
        os[nk] = score;
        ol[nk] = label;
        oa[nk] = anchor;
      }
      ++nk;
      if (lane > p && alive && (form.agnostic || label == lp) && iou_gt(bp, cmp, iou_thr)) alive = false;
      mask = __ballot(alive && lane > p);
    }
    __syncthreads();   // single-wave workgroup: orders the LDS writes above before the next chunk's reads
  }
  for (int s = nk + lane; s < max_out; s += 64) {
    *reinterpret_cast<f32x4*>(ob + (size_t)s * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    os[s] = 0.f; ol[s] = -1; oa[s] = -1;
  }
  if (lane == 0) out_count[b] = nk;
}

// rows of the region-embedding tensor selected by NMS; zero rows past the count
__global__ void __launch_bounds__(64) gather_embed_kernel(const float* __restrict__ embed, int n_anchor, int dim,
                                                          const int* __restrict__ out_anchors,
                                                          const int* __restrict__ out_count, int max_out,
                                                          float* __restrict__ out_embed) {
  const int b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
  float* dst = out_embed + ((size_t)b * max_out + s) * dim;
  if (s < out_count[b]) {
    const int a = out_anchors[(size_t)b * max_out + s];
    const float* src = embed + ((size_t)b * n_anchor + a) * dim;
    for (int i = lane * 4; i < dim; i += 256) *reinterpret_cast<f32x4*>(dst + i) = *reinterpret_cast<const f32x4*>(src + i);
  } else {
    for (int i = lane * 4; i < dim; i += 256) *reinterpret_cast<f32x4*>(dst + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

struct TopkLayout {
  size_t hist, state, blk, keys, total;
  int nblk, cap;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

TopkLayout topk_layout(int batch, long long n, int nms_pre) {
  TopkLayout L;
  L.cap = 1;
  while (L.cap < nms_pre) L.cap <<= 1;
  L.nblk = (int)((n + CH - 1) / CH);
  size_t off = 0;
  L.hist = off; off = align256(off + (size_t)batch * 3 * HB * 4);
  L.state = off; off = align256(off + (size_t)batch * sizeof(TopkState));
  L.blk = off; off = align256(off + (size_t)batch * L.nblk * 2 * 4);
  L.keys = off; off = align256(off + (size_t)batch * L.cap * 8);
  L.total = off;
  return L;
}

}  // namespace

extern "C" int32_t wd_topk_capacity(int32_t nms_pre) {
  int cap = 1;
  while (cap < nms_pre) cap <<= 1;
  return cap;
}

extern "C" int64_t wd_topk_workspace_bytes(int32_t batch, int64_t n_per_image, int32_t nms_pre) {
  if (batch <= 0 || n_per_image <= 0 || nms_pre <= 0) return 0;
  return (int64_t)topk_layout(batch, n_per_image, nms_pre).total;
}

extern "C" int wd_topk_candidates(const float* scores, int32_t batch, int64_t n, float thr, int32_t nms_pre,
                                  int32_t* out_idx, float* out_score, int32_t* out_count, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
  if (!scores || !out_idx || !out_score || !out_count || !workspace) return WD_ERR_BAD_ARG;
  if (batch <= 0 || batch > 65535 || n <= 0 || n > 0x7fffffffLL || nms_pre <= 0 || nms_pre > (1 << 20))
    return WD_ERR_BAD_ARG;
  if (reinterpret_cast<uintptr_t>(workspace) & 255u) return WD_ERR_BAD_ARG;
  const TopkLayout L = topk_layout(batch, n, nms_pre);
  if ((size_t)workspace_bytes < L.total) return WD_ERR_WORKSPACE;
  char* ws = static_cast<char*>(workspace);
  unsigned* hist = reinterpret_cast<unsigned*>(ws + L.hist);
  TopkState* state = reinterpret_cast<TopkState*>(ws + L.state);
  unsigned* blk = reinterpret_cast<unsigned*>(ws + L.blk);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws + L.keys);
  hipStream_t st = static_cast<hipStream_t>(stream);
  {
    const long long n0 = (long long)((L.blk - L.hist) / 4), n1 = (long long)batch * L.cap * 2;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n0 + 255) / 256 < 1024 ? (n0 + 255) / 256 : 1024)), dim3(256), 0,
                       st, reinterpret_cast<unsigned*>(ws + L.hist), 0u, n0);                      // hist + state
    hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n1 + 255) / 256 < 1024 ? (n1 + 255) / 256 : 1024)), dim3(256), 0,
                       st, reinterpret_cast<unsigned*>(keys), 0xFFFFFFFFu, n1);                    // pad keys sort last
  }
  const dim3 gchunk(L.nblk, batch), gimg(batch);
  hipLaunchKernelGGL(topk_hist_kernel<0>, gchunk, dim3(256), 0, st, scores, (long long)n, thr, state, hist);
  hipLaunchKernelGGL(topk_pick_kernel<0>, gimg, dim3(256), 0, st, state, hist, nms_pre);
  hipLaunchKernelGGL(topk_hist_kernel<1>, gchunk, dim3(256), 0, st, scores, (long long)n, thr, state, hist);
  hipLaunchKernelGGL(topk_pick_kernel<1>, gimg, dim3(256), 0, st, state, hist, nms_pre);
  hipLaunchKernelGGL(topk_hist_kernel<2>, gchunk, dim3(256), 0, st, scores, (long long)n, thr, state, hist);
  hipLaunchKernelGGL(topk_pick_kernel<2>, gimg, dim3(256), 0, st, state, hist, nms_pre);
  hipLaunchKernelGGL(topk_count_kernel, gchunk, dim3(256), 0, st, scores, (long long)n, thr, state, blk, L.nblk);
  hipLaunchKernelGGL(topk_scan_kernel, gimg, dim3(256), 0, st, blk, L.nblk);
  hipLaunchKernelGGL(topk_scatter_kernel, gchunk, dim3(256), 0, st, scores, (long long)n, thr, state, blk, L.nblk, keys,
                     L.cap);
  {
    const int nchunk = (L.cap + SORT_CHUNK - 1) / SORT_CHUNK;
    const dim3 gch(nchunk, batch), gel((L.cap + 255) / 256, batch);
    hipLaunchKernelGGL(topk_sort_chunk_kernel, gch, dim3(1024), SORT_CHUNK * 8, st, keys, L.cap, state, 0);
    for (int size = 2 * SORT_CHUNK; size <= L.cap; size <<= 1) {
      for (int j = size >> 1; j >= SORT_CHUNK; j >>= 1)
        hipLaunchKernelGGL(topk_sort_global_kernel, gel, dim3(256), 0, st, keys, L.cap, state, size, j);
      hipLaunchKernelGGL(topk_sort_chunk_kernel, gch, dim3(1024), SORT_CHUNK * 8, st, keys, L.cap, state, size);
    }
    hipLaunchKernelGGL(topk_unpack_kernel, gel, dim3(256), 0, st, keys, L.cap, state, out_idx, out_score, out_count);
  }
  return wd_launch_status();
}

extern "C" int64_t wd_nms_workspace_bytes(int32_t batch) {
  return batch <= 0 ? 0 : (int64_t)align256((size_t)batch * 2 * sizeof(unsigned));
}

extern "C" int wd_nms_gather(const int32_t* cand_idx, const float* cand_score, const int32_t* cand_count,
                             int32_t cand_stride, const float* boxes, int32_t n_anchor, int32_t k, const float* meta,
                             float iou_thr, int32_t max_out, int32_t nms_mode, int32_t mode_param, const float* embed,
                             int32_t embed_dim, float* out_boxes, float* out_scores, int32_t* out_labels,
                             int32_t* out_anchors, int32_t* out_count, float* out_embed, int32_t batch, void* workspace,
                             int64_t workspace_bytes, void* stream) {
  if (!cand_idx || !cand_score || !cand_count || !boxes || !meta || !out_boxes || !out_scores || !out_labels ||
      !out_anchors || !out_count)
    return WD_ERR_BAD_ARG;
  if (batch <= 0 || batch > 65535 || n_anchor <= 0 || k <= 0 || max_out <= 0 || max_out > NMS_MAX_OUT || cand_stride <= 0)
    return WD_ERR_BAD_ARG;
  if (nms_mode != WD_NMS_VANILLA && nms_mode != WD_NMS_TORCHVISION && nms_mode != WD_NMS_MMCV) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(boxes) || !wd_aligned16(out_boxes)) return WD_ERR_BAD_ARG;
  if (embed && (!out_embed || embed_dim <= 0 || (embed_dim & 3) || !wd_aligned16(embed) || !wd_aligned16(out_embed)))
    return WD_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  unsigned* bounds = static_cast<unsigned*>(workspace);
  if (nms_mode != WD_NMS_VANILLA) {
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 3u)) return WD_ERR_BAD_ARG;
    if (workspace_bytes < wd_nms_workspace_bytes(batch)) return WD_ERR_WORKSPACE;
    hipLaunchKernelGGL(fill_u32_kernel, dim3((2 * batch + 255) / 256), dim3(256), 0, st, bounds, 0u, (long long)2 * batch);
    hipLaunchKernelGGL(nms_bounds_kernel, dim3((cand_stride + 1023) / 1024, batch), dim3(256), 0, st, cand_idx, cand_count,
                       cand_stride, boxes, n_anchor, k, meta, nms_mode, mode_param, bounds);
  }
  hipLaunchKernelGGL(nms_kernel, dim3(batch), dim3(64), 0, st, cand_idx, cand_score, cand_count, cand_stride, boxes,
                     n_anchor, k, meta, iou_thr, max_out, nms_mode, mode_param, bounds, out_boxes, out_scores, out_labels,
                     out_anchors, out_count);
  if (embed)
    hipLaunchKernelGGL(gather_embed_kernel, dim3(max_out, batch), dim3(64), 0, st, embed, n_anchor, embed_dim,
                       out_anchors, out_count, max_out, out_embed);
  return wd_launch_status();
}
