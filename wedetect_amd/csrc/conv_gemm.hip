// conv_gemm.hip — fp32 implicit-GEMM convolution / linear / similarity kernels for gfx950.
//
// One LDS-tiled MFMA template serves every dense contraction of the WeDetect image tower:
// ConvNeXt pointwise MLPs, patchify convs, neck/head 1x1 and 3x3 convs, the 2x2 transposed
// conv, the region x text similarity GEMM and the retrieval similarity with fused
// sigmoid + max-over-regions epilogue.
//
// Arithmetic: v_mfma_f32_16x16x4_f32 — f32 inputs, f32 accumulate, bit-equal to an fmaf
// chain (the reference computes in fp32; no reduced precision anywhere).  MFMA roof for
// this dtype is 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Tiling (per workgroup of WM x WN waves, wave = 64 lanes):
//   block tile  BM = 16*TM*WM rows (m)  x  BN = 16*TN*WN cols (n),  K step BK = 32
//   wave tile   (16*TM) x (16*TN), TM*TN accumulators of 4 VGPRs
//   operands are staged global -> VGPR (16-byte loads, im2col gather + zero padding done
//   here) -> LDS [rows][BK+4], double buffered: the loads of tile t+1 are issued before
//   the 128 MFMAs of tile t and written to the other LDS buffer after them (one barrier
//   per K step).
//   The MFMA "A" operand carries WEIGHT rows and the "B" operand ACTIVATION rows, so a lane
//   ends up holding 4 consecutive output channels of one pixel: epilogue loads/stores are
//   16-byte vectors along n.
//   K is consumed 16 at a time: lane (i = lane&15, g = lane>>4) reads the float4
//   [k0+4g, k0+4g+4) of row i with one ds_read_b128 and feeds component r to the r-th of 4
//   MFMAs — a permutation of k that is identical for both operands, hence harmless.
#include "common.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;   // floats; 144-byte rows keep every float4 16-byte aligned

template <int TM, int TN, int WM, int WN>
struct Tile {
  static constexpr int BM = 16 * TM * WM;
  static constexpr int BN = 16 * TN * WN;
  static constexpr int NT = 64 * WM * WN;
  static constexpr int A_PT = (BM * 8) / NT;             // float4 chunks of A per thread per K step
  static constexpr int B_PT = (BN * 8 + NT - 1) / NT;
  static constexpr int RSTEP = NT / 8;
  static constexpr int LDS_BYTES = 2 * (BM + BN) * LDS_LD * 4;
  static_assert((BM * 8) % NT == 0, "A tile must split evenly over the threads");
};

// ---------------------------------------------------------------------------------------
// A-operand gather.  CONV=false: plain row-major [m][k] with row stride lda.
// CONV=true: NHWC implicit im2col, k = (kh, kw, ci).
// ---------------------------------------------------------------------------------------
template <int A_PT, int RSTEP, bool CONV>
struct ALoader {
  const float* a;
  int lda, hin, win, cin, kw_, K;
  int pix[A_PT];
  int hi0[A_PT];
  int wi0[A_PT];
  bool ok[A_PT];

  __device__ __forceinline__ void init(const WdConvGemm& p, int m0, int r0) {
    a = p.a; lda = p.lda; hin = p.hin; win = p.win; cin = p.cin; kw_ = p.kw; K = p.k;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      const int m = m0 + r0 + i * RSTEP;
      ok[i] = m < p.m;
      if (CONV) {
        const int wo = m % p.wout;
        const int q = m / p.wout;
        const int ho = q % p.hout;
        const int b = q / p.hout;
        hi0[i] = ho * p.stride - p.pad;
        wi0[i] = wo * p.stride - p.pad;
        pix[i] = b * p.hin * p.win;
      } else {
        pix[i] = m; hi0[i] = 0; wi0[i] = 0;
      }
    }
  }

  __device__ __forceinline__ void load(int k, f32x4 (&reg)[A_PT]) const {
    const bool kok = k < K;
    int kh = 0, kw = 0, ci = k;
    if (CONV) {
      const int tap = k / cin;
      ci = k - tap * cin;
      kh = tap / kw_;
      kw = tap - kh * kw_;
    }
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (CONV) {
        const int hi = hi0[i] + kh, wi = wi0[i] + kw;
        if (ok[i] && kok && (unsigned)hi < (unsigned)hin && (unsigned)wi < (unsigned)win)
          v = *reinterpret_cast<const f32x4*>(a + (size_t)(pix[i] + hi * win + wi) * lda + ci);
      } else {
        if (ok[i] && kok) v = *reinterpret_cast<const f32x4*>(a + (size_t)pix[i] * lda + k);
      }
      reg[i] = v;
    }
  }
};

// Rows of one image's region-embedding block for wd_retrieval_max.
template <int A_PT, int RSTEP>
struct RegionLoader {
  const float* base;
  int dim, rows;
  int row[A_PT];
  __device__ __forceinline__ void init(const float* e, int img, int rows_per_img, int dim_, int r0) {
    base = e + (size_t)img * rows_per_img * dim_;
    dim = dim_; rows = rows_per_img;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) row[i] = r0 + i * RSTEP;
  }
  __device__ __forceinline__ void load(int k, f32x4 (&reg)[A_PT]) const {
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row[i] < rows && k < dim) v = *reinterpret_cast<const f32x4*>(base + (size_t)row[i] * dim + k);
      reg[i] = v;
    }
  }
};

// ---------------------------------------------------------------------------------------
// Main loop shared by all kernels: acc[tm][tn] (+)= X[m, :] . W[n, :]
// ---------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN, class AL>
__device__ __forceinline__ void gemm_mainloop(const AL& al, const float* __restrict__ w, int n0, int N, int K,
                                              f32x4 (&acc)[TM][TN], float* smem) {
  using T = Tile<TM, TN, WM, WN>;
  constexpr int BM = T::BM, BN = T::BN, NT = T::NT, A_PT = T::A_PT, B_PT = T::B_PT, RSTEP = T::RSTEP;
  float* As = smem;
  float* Bs = smem + 2 * BM * LDS_LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int kc = t & 7, r0 = t >> 3;
  const int nk = (K + BK - 1) / BK;

  f32x4 areg[A_PT], breg[B_PT];

  auto load_b = [&](int k) {
    const bool kok = k < K;
#pragma unroll
    for (int j = 0; j < B_PT; ++j) {
      const int c = t + j * NT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c < BN * 8) {
        const int n = n0 + (c >> 3);
        if (n < N && kok) v = *reinterpret_cast<const f32x4*>(w + (size_t)n * K + k);
      }
      breg[j] = v;
    }
  };
  auto store = [&](int buf) {
    float* ad = As + buf * BM * LDS_LD + r0 * LDS_LD + kc * 4;
#pragma unroll
    for (int i = 0; i < A_PT; ++i) *reinterpret_cast<f32x4*>(ad + i * RSTEP * LDS_LD) = areg[i];
    float* bd = Bs + buf * BN * LDS_LD;
#pragma unroll
    for (int j = 0; j < B_PT; ++j) {
      const int c = t + j * NT;
      if (c < BN * 8) *reinterpret_cast<f32x4*>(bd + (c >> 3) * LDS_LD + (c & 7) * 4) = breg[j];
    }
  };

#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = f32x4{0.f, 0.f, 0.f, 0.f};

  al.load(kc * 4, areg);
  load_b(kc * 4);
  store(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nk;
    if (more) {
      const int k = (kt + 1) * BK + kc * 4;
      al.load(k, areg);
      load_b(k);
    }
    const float* as = As + cur * BM * LDS_LD + (wm * TM * 16 + (lane & 15)) * LDS_LD + 4 * (lane >> 4);
    const float* bs = Bs + cur * BN * LDS_LD + (wn * TN * 16 + (lane & 15)) * LDS_LD + 4 * (lane >> 4);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f32x4 xf[TM], wf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) xf[tm] = *reinterpret_cast<const f32x4*>(as + tm * 16 * LDS_LD + ks * 16);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) wf[tn] = *reinterpret_cast<const f32x4*>(bs + tn * 16 * LDS_LD + ks * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[tn][r], xf[tm][r], acc[tm][tn], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// conv / linear / similarity kernel
// ---------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN, bool CONV>
__global__ void __launch_bounds__(64 * WM * WN) conv_gemm_kernel(const WdConvGemm p, int nbn, int vec_c, int vec_res) {
  using T = Tile<TM, TN, WM, WN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int bn = blockIdx.x % nbn, bm = blockIdx.x / nbn;
  const int m0 = bm * T::BM, n0 = bn * T::BN;

  ALoader<T::A_PT, T::RSTEP, CONV> al;
  al.init(p, m0, t >> 3);
  f32x4 acc[TM][TN];
  gemm_mainloop<TM, TN, WM, WN>(al, p.w, n0, p.n, p.k, acc, smem);

  // ---- epilogue: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + 0..3] per tile
  const int ncq = p.n >> 2;   // deconv: channels per tap
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
    const int m = m0 + (wm * TM + tm) * 16 + (lane & 15);
    if (m >= p.m) continue;
    size_t crow;
    int hw2 = 0;
    if (p.out_mode == WD_OUT_DECONV2X2) {
      const int wq = m % p.wout;
      const int q = m / p.wout;
      const int hq = q % p.hout;
      const int b = q / p.hout;
      crow = ((size_t)(b * 2 * p.hout + 2 * hq) * (2 * p.wout) + 2 * wq);   // pixel of tap (0,0)
      hw2 = 2 * p.wout;
    } else if (p.c_batch_stride > 0) {
      const int hw = p.hout * p.wout;
      const int b = m / hw;
      crow = (size_t)b * p.c_batch_stride + (size_t)(m - b * hw);
    } else {
      crow = (size_t)m;
    }
    float oscale = p.out_scale, obias = p.out_bias;
    if (p.seg_rows > 0) {
      const int pos = m % p.seg_rows;
      const int lvl = (pos >= p.seg_end0) + (pos >= p.seg_end1);
      oscale = lvl == 0 ? p.seg_scale[0] : lvl == 1 ? p.seg_scale[1] : p.seg_scale[2];
      obias = lvl == 0 ? p.seg_bias[0] : lvl == 1 ? p.seg_bias[1] : p.seg_bias[2];
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int n = n0 + (wn * TN + tn) * 16 + 4 * (lane >> 4);
      if (n >= p.n) continue;
      f32x4 v = acc[tm][tn];
      const bool full = n + 3 < p.n;
      float vb[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.n) vb[r] = p.bias[n + r];
      }
      float o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = v[r] + vb[r];
        x = wd_act(x, p.act);
        x = x * oscale + obias;
        if (p.sigmoid) x = wd_sigmoid(x);
        o[r] = x;
      }
      if (p.res) {
        const float* rp = p.res + (size_t)m * p.ldres + n;
        if (full && vec_res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] += p.res_alpha * rv[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < p.n) o[r] += p.res_alpha * rp[r];
        }
      }
      float* cp;
      if (p.out_mode == WD_OUT_DECONV2X2) {
        const int tap = n / ncq, co = n - tap * ncq;
        cp = p.c + (crow + (size_t)(tap >> 1) * hw2 + (tap & 1)) * p.ldc + co;
      } else {
        cp = p.c + crow * p.ldc + n;
      }
      if (full && vec_c) {
        *reinterpret_cast<f32x4*>(cp) = f32x4{o[0], o[1], o[2], o[3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < p.n) cp[r] = o[r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// retrieval similarity: one workgroup = (64 classes, one image of <= 320 region rows)
// ---------------------------------------------------------------------------------------
constexpr int RT_TM = 5, RT_TN = 4, RT_WM = 4, RT_WN = 1;   // 320 rows x 64 classes

__global__ void __launch_bounds__(256) retrieval_max_kernel(const float* __restrict__ e, const float* __restrict__ tb,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ bias,
                                                            const int* __restrict__ count, float* __restrict__ out,
                                                            int rows_per_img, int n_cls, int dim) {
  using T = Tile<RT_TM, RT_TN, RT_WM, RT_WN>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int img = blockIdx.y;
  const int n0 = blockIdx.x * T::BN;

  RegionLoader<T::A_PT, T::RSTEP> al;
  al.init(e, img, rows_per_img, dim, t >> 3);
  f32x4 acc[RT_TM][RT_TN];
  gemm_mainloop<RT_TM, RT_TN, RT_WM, RT_WN>(al, tb, n0, n_cls, dim, acc, smem);

  const int cnt = count[img];
  float cmax[RT_TN][4];
#pragma unroll
  for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) cmax[tn][r] = 0.f;   // sigmoid > 0: 0 is the identity of max here
#pragma unroll
  for (int tm = 0; tm < RT_TM; ++tm) {
    const int row = (wave * RT_TM + tm) * 16 + (lane & 15);
    if (row < cnt && row < rows_per_img) {
      const float s = expf(scale[(size_t)img * rows_per_img + row]);
      const float b = bias[(size_t)img * rows_per_img + row];
#pragma unroll
      for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) cmax[tn][r] = fmaxf(cmax[tn][r], wd_sigmoid(acc[tm][tn][r] * s + b));
    }
  }
  // max over the 16 rows held by lanes with equal (lane>>4)
#pragma unroll
  for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = cmax[tn][r];
      v = fmaxf(v, __shfl_xor(v, 1, 64));
      v = fmaxf(v, __shfl_xor(v, 2, 64));
      v = fmaxf(v, __shfl_xor(v, 4, 64));
      v = fmaxf(v, __shfl_xor(v, 8, 64));
      cmax[tn][r] = v;
    }
  __syncthreads();                 // all waves are done with the operand tiles in LDS
  float* red = smem;               // [4 waves][64 classes]
  if ((lane & 15) == 0) {
#pragma unroll
    for (int tn = 0; tn < RT_TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave * 64 + tn * 16 + 4 * (lane >> 4) + r] = cmax[tn][r];
  }
  __syncthreads();
  if (t < 64) {
    const int n = n0 + t;
    if (n < n_cls) {
      const float v = fmaxf(fmaxf(red[t], red[64 + t]), fmaxf(red[128 + t], red[192 + t]));
      out[(size_t)img * n_cls + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------------------
template <int TM, int TN, int WM, int WN>
int launch_cfg(const WdConvGemm& p, hipStream_t st) {
  using T = Tile<TM, TN, WM, WN>;
  const bool conv = !(p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0);
  const int nbm = (p.m + T::BM - 1) / T::BM, nbn = (p.n + T::BN - 1) / T::BN;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c) && (p.out_mode == WD_OUT_ROWS || (p.n % 16 == 0));
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  static bool attr_plain = false, attr_conv = false;
  if (conv) {
    auto k = conv_gemm_kernel<TM, TN, WM, WN, true>;
    if (!attr_conv) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_BYTES) != hipSuccess) return WD_ERR_LAUNCH;
      attr_conv = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, nbn, vec_c, vec_res);
  } else {
    auto k = conv_gemm_kernel<TM, TN, WM, WN, false>;
    if (!attr_plain) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              T::LDS_BYTES) != hipSuccess) return WD_ERR_LAUNCH;
      attr_plain = true;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(T::NT), T::LDS_BYTES, st, p, nbn, vec_c, vec_res);
  }
  return wd_launch_status();
}

// tile widths available along n, widest first; pick the least padded, ties -> widest
constexpr int kBnChoices[5] = {128, 96, 80, 64, 48};

int pick_bn(int n) {
  int best = 128;
  long long best_pad = -1;
  for (int i = 0; i < 5; ++i) {
    const int bn = kBnChoices[i];
    const long long padded = (long long)((n + bn - 1) / bn) * bn;
    if (best_pad < 0 || padded < best_pad) { best_pad = padded; best = bn; }
  }
  return best;
}

}  // namespace

extern "C" const char* wd_conv_gemm_config(int32_t m, int32_t n, int32_t k) {
  (void)m; (void)k;
  switch (pick_bn(n)) {
    case 128: return "128x128x32/4w";
    case 96: return "256x96x32/4w";
    case 80: return "256x80x32/4w";
    case 64: return "256x64x32/4w";
    default: return "256x48x32/4w";
  }
}

extern "C" int wd_conv_gemm(const WdConvGemm* pp, void* stream) {
  if (!pp) return WD_ERR_BAD_ARG;
  const WdConvGemm& p = *pp;
  if (!p.a || !p.w || !p.c) return WD_ERR_BAD_ARG;
  if (p.m <= 0 || p.n <= 0 || p.k <= 0) return WD_ERR_BAD_ARG;
  if (p.cin <= 0 || p.cin % 4 || p.lda % 4 || p.lda < p.cin) return WD_ERR_BAD_ARG;
  if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0) return WD_ERR_BAD_ARG;
  if (p.k != p.kh * p.kw * p.cin) return WD_ERR_BAD_ARG;
  if ((long long)p.batch * p.hout * p.wout != (long long)p.m) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(p.a) || !wd_aligned16(p.w)) return WD_ERR_BAD_ARG;
  if (p.act < WD_ACT_NONE || p.act > WD_ACT_GELU) return WD_ERR_BAD_ARG;
  if (p.out_mode == WD_OUT_DECONV2X2) {
    if (p.n % 16 || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.res) return WD_ERR_BAD_ARG;
    if (p.ldc < p.n / 4) return WD_ERR_BAD_ARG;
  } else if (p.out_mode == WD_OUT_ROWS) {
    if (p.ldc < p.n) return WD_ERR_BAD_ARG;
  } else {
    return WD_ERR_BAD_ARG;
  }
  if (p.res && p.ldres < p.n) return WD_ERR_BAD_ARG;
  if (p.c_batch_stride < 0 || (p.c_batch_stride > 0 && (p.out_mode != WD_OUT_ROWS || p.c_batch_stride < p.hout * p.wout)))
    return WD_ERR_BAD_ARG;
  if (p.seg_rows < 0 || (p.seg_rows > 0 && !(0 <= p.seg_end0 && p.seg_end0 <= p.seg_end1 && p.seg_end1 <= p.seg_rows)))
    return WD_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (pick_bn(p.n)) {
    case 128: return launch_cfg<4, 4, 2, 2>(p, st);
    case 96: return launch_cfg<4, 6, 4, 1>(p, st);
    case 80: return launch_cfg<4, 5, 4, 1>(p, st);
    case 64: return launch_cfg<4, 4, 4, 1>(p, st);
    default: return launch_cfg<4, 3, 4, 1>(p, st);
  }
}

extern "C" int wd_retrieval_max(const float* e, const float* t, const float* scale, const float* bias,
                                const int32_t* count, float* out, int32_t n_img, int32_t rows_per_img,
                                int32_t n_cls, int32_t dim, void* stream) {
  using T = Tile<RT_TM, RT_TN, RT_WM, RT_WN>;
  if (!e || !t || !scale || !bias || !count || !out) return WD_ERR_BAD_ARG;
  if (n_img <= 0 || n_cls <= 0 || dim <= 0 || dim % 4) return WD_ERR_BAD_ARG;
  if (rows_per_img <= 0 || rows_per_img > T::BM) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(e) || !wd_aligned16(t)) return WD_ERR_BAD_ARG;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(retrieval_max_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES) != hipSuccess)
      return WD_ERR_LAUNCH;
    attr = true;
  }
  const int nbn = (n_cls + T::BN - 1) / T::BN;
  hipLaunchKernelGGL(retrieval_max_kernel, dim3(nbn, n_img), dim3(256), T::LDS_BYTES,
                     static_cast<hipStream_t>(stream), e, t, scale, bias, count, out, rows_per_img, n_cls, dim);
  return wd_launch_status();
}
