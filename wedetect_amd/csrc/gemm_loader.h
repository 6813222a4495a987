// Operand gather shared by the fp32 (conv_gemm.hip) and split-fp16 (split_gemm.hip) GEMM kernels.
#pragma once
#include "common.h"

namespace {

// Out-of-range operand chunks (M/N/K tails, conv zero padding) are READ from this zero
// block instead of being masked after the load: the K loop stays branch-free and the loaded
// registers have no consumer before the LDS store, so the loads stay in flight across the
// whole MFMA phase.
__device__ __attribute__((aligned(16))) float g_zero4[4] = {0.f, 0.f, 0.f, 0.f};

// Device address of this translation unit's zero block ON THE CURRENT DEVICE: a __device__ symbol has one address per
// device, and a process may drive several GPUs (round 3 cached the first device's address per process — ADVICE r3: a
// second device would have read its halo zeros from device 0's memory).  nullptr on failure.
static inline const float* wd_zero_block() {
  static const float* zero[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const float*& z = zero[dev & 63];
  if (!z) {
    void* zp = nullptr;
    if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero4)) != hipSuccess || !zp) return nullptr;
    z = static_cast<const float*>(zp);
  }
  return z;
}

// ---------------------------------------------------------------------------------------
// A-operand gather, branch-free: an out-of-range chunk reads g_zero4.
// CONV=false: plain row-major [m][k] with row stride lda.
// CONV=true : NHWC implicit im2col, k = (kh, kw, ci).
// ---------------------------------------------------------------------------------------
template <int A_PT, int RSTEP, bool CONV>
struct ALoader {
  const float* a;
  const float* zero;
  int lda, hin, win, cin, kw_, K;
  int pix[A_PT];
  int hi0[A_PT];
  int wi0[A_PT];
  bool ok[A_PT];

  // zero_block: 16 readable bytes of zeros (default: this translation unit's g_zero4; a kernel
  // argument avoids re-deriving the symbol address inside the K loop)
  __device__ __forceinline__ void init(const WdConvGemm& p, int m0, int r0, const float* zero_block = g_zero4) {
    zero = zero_block;
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
        pix[i] = ok[i] ? m : 0; hi0[i] = 0; wi0[i] = 0;
      }
    }
  }

  // kbase = K-step origin (wave-uniform), kc4 = this thread's offset inside the step.
  // When cin is a multiple of the K step a step never straddles a filter tap, so the tap
  // decomposition is the same for every thread and runs on the scalar unit.
  template <int BKS>
  __device__ __forceinline__ void load(int kbase, int kc4, f32x4 (&reg)[A_PT]) const {
    const int k = kbase + kc4;
    const bool kok = k < K;
    int kh = 0, kw = 0, ci = kok ? k : 0;
    if (CONV) {
      if (cin % BKS == 0) {
        const int tap = kbase / cin;
        ci = kbase - tap * cin + kc4;
        kh = tap / kw_;
        kw = tap - kh * kw_;
      } else {
        const int tap = k / cin;
        ci = k - tap * cin;
        kh = tap / kw_;
        kw = tap - kh * kw_;
      }
    }
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
      bool valid;
      size_t off;
      if (CONV) {
        const int hi = hi0[i] + kh, wi = wi0[i] + kw;
        valid = ok[i] && kok && (unsigned)hi < (unsigned)hin && (unsigned)wi < (unsigned)win;
        off = (size_t)(pix[i] + hi * win + wi) * lda + ci;
      } else {
        valid = ok[i] && kok;
        off = (size_t)pix[i] * lda + ci;
      }
      reg[i] = *reinterpret_cast<const f32x4*>(valid ? a + off : zero);
    }
  }
};

}  // namespace
