// stem.hip — the ConvNeXt stem as ONE kernel: uint8 RGB NHWC image -> x / 255 -> Conv2d(3, C0, 4, stride 4) + bias ->
// LayerNorm over channels (channels_first, eps 1e-6) -> fp32 rows [B * H/4 * W/4, C0]
// (wedetect/models/backbones/mm_backbone.py:185-190 stem, :25-47 LayerNorm; data_preprocessor.py:35-36 for the / 255).
//
// The three-kernel form (wd_stem_patchify -> wd_conv_gemm fp32 -> wd_layernorm_rows) moves the 48-float patch matrix
// (157 MB at WeDetect-Base batch 32) and the pre-norm activations (419 MB, twice) through HBM: 520 us.  Here the image is
// read once (39 MB) and the normalised rows are written once (419 MB).  A wave owns 16 output pixels at a time:
//   * the whole weight matrix [C0 x 48] lives in its registers as v_mfma_f32_16x16x4_f32 A-operand fragments (12 C0 / 16
//     VGPRs);
//   * a lane (pixel p = lane & 15, k group kk = lane >> 4) loads three dwords of its pixel's 4 x 12-byte patch — the bytes
//     k = 16 ks + 4 kk + 0..3 for ks = 0, 1, 2 (one block ahead) — converts them (x / 255.0f: a true division like the
//     reference's, done once per byte value into a 1 KB LDS table), and feeds
//     byte r of dword ks to MFMA (ks, r): exactly the K order of the fp32 GEMM kernel (conv_gemm.hip: k = 16 ks + 4 kk + r
//     in MFMA (ks, r)), so every accumulator sees the same chain of operations;
//   * the accumulator layout gives a lane 4 consecutive channels (16 jb + 4 kk + 0..3) of its pixel per 16-channel block:
//     the quad structure of wd_layernorm_rows.  Its row sums are butterflies over the quad index q = 4 jb + kk; the levels
//     over jb are register adds in the butterfly's order, the last two are lane exchanges (kk ^ 2 = lane ^ 32, kk ^ 1 =
//     lane ^ 16).
// Same operations in the same order as the three kernels: BIT-IDENTICAL (tests/test_gpu_kernels.py::test_stem_fused_*).
#include "common.h"

namespace {

template <int NB>      // C0 = 16 NB
__global__ void __launch_bounds__(256, NB <= 8 ? 2 : 1) stem_fused_kernel(const uint8_t* __restrict__ img, const float* __restrict__ wgt,
                                                         const float* __restrict__ bias, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ out, int h, int w,
                                                         long long pixels, float eps) {
  // nothing here may be contracted: the three-kernel form rounds every product before it adds
#pragma clang fp contract(off)
  constexpr int C0 = 16 * NB;
  constexpr int NBP = NB <= 8 ? 8 : 16;              // quads per lane padded to the LayerNorm group (32 or 64 lanes x 1 quad)
  const int lane = threadIdx.x & 63, p = lane & 15, kk = lane >> 4;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;

  f32x4 wf[NB][3];
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) wf[jb][ks] = *reinterpret_cast<const f32x4*>(wgt + (size_t)(16 * jb + p) * 48 + 16 * ks + 4 * kk);
  // Nothing but the next block's image bytes may be in flight inside the loop: loads retire in order and stores count with
  // them, so one consumed in the loop would wait for the prefetch behind it and for the previous block's stores.  The weight
  // fragments are therefore complete (and pinned) before the loop, and bias / gamma / beta come from LDS.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int jb = 0; jb < NB; ++jb)
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) asm volatile("" : "+v"(wf[jb][ks]));
  // x / 255.0f for the 256 byte values, divided once (a true division, like the reference) and looked up afterwards
  __shared__ float lut[256];
  __shared__ __attribute__((aligned(16))) float sb[C0], sg[C0], st[C0];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  if (threadIdx.x < C0) {
    sb[threadIdx.x] = bias[threadIdx.x];
    sg[threadIdx.x] = gamma[threadIdx.x];
    st[threadIdx.x] = beta[threadIdx.x];
  }
  __syncthreads();
  const unsigned wo_n = w >> 2, ho_n = h >> 2;
  const long long nblk = (pixels + 15) >> 4;
  // the three dwords of pixel block `blk` this lane converts: patch byte k = 12 kh + (3 kw + c); dword ks starts at
  // k0 = 16 ks + 4 kk -> image row kh = k0 / 12, dword (k0 % 12) / 4 of the pixel's 12 bytes there
  // issued as inline assembly: the compiler does not track these loads, so it cannot put a conservative "everything has
  // landed" wait (stores included) at the loop's back edge; the loop waits for them by count instead
  auto fetch = [&](long long blk, uint32_t (&u)[3]) {
    const long long pix = blk * 16 + p;
    const unsigned pc = (unsigned)(pix < pixels ? pix : pixels - 1);
    const unsigned q_ = pc / wo_n, wo = pc - q_ * wo_n;
    const unsigned b = q_ / ho_n, ho = q_ - b * ho_n;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      const int k0 = 16 * ks + 4 * kk, kh = k0 / 12, j = (k0 - 12 * kh) >> 2;
      const uint8_t* src = img + (((long long)b * h + (ho * 4 + kh)) * (long long)w + wo * 4) * 3 + 4 * j;   // 4-byte aligned (w % 4 == 0)
      asm volatile("global_load_dword %0, %1, off" : "=v"(u[ks]) : "v"(src) : "memory");
    }
  };
  uint32_t un[3] = {0u, 0u, 0u};
  if (wave < nblk) fetch(wave, un);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (long long blk = wave; blk < nblk; blk += nwaves) {
    const long long pix = blk * 16 + p;
    const bool ok = pix < pixels;
    // the bytes requested during the previous block have landed once at most its NB row stores (issued after them) remain
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
    uint32_t u[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      asm volatile("" : "+v"(un[ks]));
      u[ks] = un[ks];
    }
    if (blk + nwaves < nblk) fetch(blk + nwaves, un);            // the next block's bytes fly under this block's MFMAs
    f32x4 xv[3];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r) xv[ks][r] = lut[(u[ks] >> (8 * r)) & 255u];
    const float *bp = sb + 4 * kk, *gp = sg + 4 * kk, *tp = st + 4 * kk;     // quads re-read per block: 12 NB registers less
    f32x4 acc[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) acc[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) acc[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[jb][ks][r], xv[ks][r], acc[jb], 0, 0, 0);

    // + bias (the GEMM epilogue), then LayerNorm with the reduction tree of wd_layernorm_rows (one quad per lane of a
    // 32- or 64-lane group, xor butterfly from the top bit down)
    f32x4 v[NB];
    float t[NBP];
#pragma unroll
    for (int jb = 0; jb < NBP; ++jb) t[jb] = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      const f32x4 bq = *reinterpret_cast<const f32x4*>(bp + 16 * jb);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[jb][r] = acc[jb][r] + bq[r];
      t[jb] = 0.f + ((v[jb][0] + v[jb][1]) + (v[jb][2] + v[jb][3]));
    }
#pragma unroll
    for (int o = NBP >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int jb = 0; jb < o; ++jb) t[jb] = t[jb] + t[jb + o];
    float s = t[0];
    s += __shfl_xor(s, 32, 64);
    s += __shfl_xor(s, 16, 64);
    const float mean = s / (float)C0;
    f32x4 d[NB];
#pragma unroll
    for (int jb = 0; jb < NBP; ++jb) t[jb] = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      d[jb] = v[jb] - mean;
      t[jb] = 0.f + ((d[jb][0] * d[jb][0] + d[jb][1] * d[jb][1]) + (d[jb][2] * d[jb][2] + d[jb][3] * d[jb][3]));
    }
#pragma unroll
    for (int o = NBP >> 1; o > 0; o >>= 1)
#pragma unroll
      for (int jb = 0; jb < o; ++jb) t[jb] = t[jb] + t[jb + o];
    float sq = t[0];
    sq += __shfl_xor(sq, 32, 64);
    sq += __shfl_xor(sq, 16, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)C0 + eps);
    if (!ok) continue;
    float* orow = out + pix * C0 + 4 * kk;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      const f32x4 tn = d[jb] * rstd;
      const f32x4 gq = *reinterpret_cast<const f32x4*>(gp + 16 * jb), tq = *reinterpret_cast<const f32x4*>(tp + 16 * jb);
      f32x4 o4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o4[e] = fmaf(tn[e], gq[e], tq[e]);
      *reinterpret_cast<f32x4*>(orow + 16 * jb) = o4;
    }
  }
}

}  // namespace

extern "C" int wd_stem_fused(const uint8_t* img, int32_t batch, int32_t h, int32_t w, const float* wgt, const float* bias,
                             const float* gamma, const float* beta, int32_t c0, float eps, float* out, void* stream) {
  if (!img || !wgt || !bias || !gamma || !beta || !out || batch <= 0 || h <= 0 || w <= 0 || (h & 3) || (w & 3)) return WD_ERR_BAD_ARG;
  if (!wd_aligned16(wgt) || !wd_aligned16(bias) || !wd_aligned16(gamma) || !wd_aligned16(beta) || !wd_aligned16(out) ||
      (reinterpret_cast<uintptr_t>(img) & 3))
    return WD_ERR_BAD_ARG;
  const long long pixels = (long long)batch * (h >> 2) * (w >> 2);
  if (pixels > 0x7fffffffLL) return WD_ERR_UNSUPPORTED;
  const long long nblk = (pixels + 15) >> 4;
  const unsigned grid = (unsigned)(nblk < 4 * 512 ? (nblk + 3) / 4 : 512);         // two workgroups per CU, resident; a wave walks its blocks
  hipStream_t st = static_cast<hipStream_t>(stream);
#define WD_STEM(NB_)                                                                                                      \
  case 16 * NB_:                                                                                                           \
    hipLaunchKernelGGL(stem_fused_kernel<NB_>, dim3(grid), dim3(256), 0, st, img, wgt, bias, gamma, beta, out, h, w, pixels, eps); \
    break;
  switch (c0) {
    WD_STEM(4) WD_STEM(6) WD_STEM(8) WD_STEM(12)
    default: return WD_ERR_UNSUPPORTED;
  }
#undef WD_STEM
  return wd_launch_status();
}
