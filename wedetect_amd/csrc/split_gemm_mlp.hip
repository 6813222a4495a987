// split_gemm_mlp.hip — the ConvNeXt block's pointwise MLP as ONE kernel for the narrow stage (C = 128, hidden 512):
//     x <- x + W2 · GELU(W1 · LN(x) + b1) + b2          (mm_backbone.py:117-124, gamma folded into W2 / b2)
// fp16x3 arithmetic on pre-split operands, like the two-kernel form (wd_conv_gemm_split twice), but the 4C hidden
// activation never leaves the CU — not even for LDS.  At WeDetect-Base batch 32 the stage-1 hidden tensor is 1.68 GB per
// block: the two-kernel form writes it (pwconv1: 420 M GELUs + 1.7 GB of stores, 710-740 us) and reads it back (pwconv2:
// 2.5 GB of HBM traffic, 490-515 us).  Here a workgroup of four waves owns 128 rows, a wave 32 of them end to end:
//   * its LayerNorm rows [32 x 128] live in registers as MFMA fragments (64 VGPRs), loaded straight from memory;
//   * the hidden dimension is walked in 16 chunks of 32 columns.  GEMM 1 (K = 128: 8 k16 steps, one accumulator) ->
//     bias + GELU + hi/lo split on the accumulator where it lies -> one v_permlane32_swap per dword turns the accumulator
//     layout (lane = pixel, 4 channels of each 8-channel group per lane half) into GEMM 2's operand layout (lane half =
//     8 consecutive k) -> GEMM 2 (K = 32: 2 k16 steps) accumulates the wave's [32 x 128] output tile across all chunks;
//   * the only thing the four waves share is the weight chunks: W1 rows [32 x 128] and W2 columns [128 x 32] (16 KB each),
//     double-buffered rings filled by LDS-DMA a chunk ahead, ONE barrier per chunk; the pwconv1 bias vector sits in LDS too;
//   * 66 KB of LDS and <= 256 VGPRs: two workgroups per CU.  The loop is software-pipelined inside the wave — an iteration
//     carries GEMM 2 of chunk j - 1, the GELU / split of chunk j and GEMM 1 of chunk j + 1 as four regions of 12 MFMAs + one
//     epilogue group each, interleaved one MFMA : seven VALU by sched_group_barrier.
// Same halves, same K order per output (hidden columns ascending, 16 at a time), the same epilogue arithmetic as the two
// kernels it replaces: BIT-IDENTICAL (tests/test_gpu_split.py::test_fused_mlp_*).
//
// Measured (MI355X, 819200 rows, profiles/r03_mlp_fused.txt): 700-760 us against 1190-1290 us for the two launches.  The
// road there: one workgroup per CU with 64-row waves and the hidden chunk through LDS (three barriers per chunk): 1157 us;
// this wave-per-32-rows form without the in-wave pipelining: 843 us; pipelined: 760 us; epilogue operands requested before
// the first output store (split_epi_oct.h): 720 us (requesting them two GEMM-2 passes earlier still: no change, 28 more
// registers).  What bounds it now is VALU issue,
// not MFMA (MFMA-busy 44 %, VALU-busy 50 %, and they do not overlap: dropping the GELU arithmetic gives 471 us): ~27 VALU
// instructions per hidden value against 3 MFMAs.  Packed fp32 instructions (half the count) bought nothing — they
// serialise with the matrix pipe — and neither did one wave per SIMD (942 us).
#include "split_epi_oct.h"
#include "split_mlp_epi.h"

namespace {

constexpr int ML_C = 128, ML_H = 512, ML_BM = 128;
constexpr int MB_HC = 32, MB_NCH = ML_H / MB_HC;
constexpr int MB_W1STAGE = MB_HC * 64, MB_W2STAGE = ML_C * 64;       // a k16 step of a chunk: rows x 64 B, XOR-swizzled
constexpr int MB_W1BUF = 8 * MB_W1STAGE, MB_W2BUF = 2 * MB_W2STAGE;  // one ring slot
constexpr int MC_W1 = 0, MC_W2 = 2 * MB_W1BUF;
constexpr int MC_B1 = 2 * MB_W1BUF + 2 * MB_W2BUF, MC_LDS = MC_B1 + ML_H * 4;   // + the pwconv1 bias vector
constexpr int MC_VHEAD = 12, MC_VPM = 7;           // VALU instructions ahead of a region's first MFMA / behind each MFMA
static_assert(4 * 32 * EPI_LDT * 4 <= MC_B1, "the epilogue patches reuse the weight rings");
static_assert(2 * MC_LDS <= 160 * 1024, "two workgroups per CU");

// one 1 KB LDS-DMA: 64 lanes x 16 B from base + voff[lane] to LDS [lds_addr, +1024)
__device__ __forceinline__ void ml_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

struct MlpArgs {
  const unsigned char* a;       // LayerNorm rows as fp16 hi/lo groups [m][128] (row = 512 B)
  const unsigned char* w1;      // split weights [512][128] (row = 512 B)
  const unsigned char* w2;      // split weights [128][512] (row = 2048 B)
  const float* b1;              // [512]
  const float* b2;              // [128]
  float* x;                     // residual in / output out, fp32 [m][128]
  unsigned* range_flag;
  int m;
  float unscale1, unscale2, hid_scale;
};

// one k16 step of GEMM 2 (12 MFMAs) / four k16 steps of GEMM 1 (12 MFMAs)
__device__ __forceinline__ void mc_gemm2_step(const unsigned char* w2s, int off_h, int off_l, const h8 xh, const h8 xl, f32x16 (&out)[1][4]) {
  h8 wh[4], wl[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) {
    wl[jn] = *reinterpret_cast<const h8*>(w2s + jn * 2048 + off_l);
    wh[jn] = *reinterpret_cast<const h8*>(w2s + jn * 2048 + off_h);
  }
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[jn], xh, out[0][jn], 0, 0, 0);
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xl, out[0][jn], 0, 0, 0);
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xh, out[0][jn], 0, 0, 0);
}
template <int S0>
__device__ __forceinline__ void mc_gemm1_steps(const unsigned char* w1b, int off_h, int off_l, const h8 (&afh)[8], const h8 (&afl)[8], f32x16& hid) {
  h8 wh[4], wl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    wl[s] = *reinterpret_cast<const h8*>(w1b + (S0 + s) * MB_W1STAGE + off_l);
    wh[s] = *reinterpret_cast<const h8*>(w1b + (S0 + s) * MB_W1STAGE + off_h);
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[s], afh[S0 + s], hid, 0, 0, 0);
    hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], afl[S0 + s], hid, 0, 0, 0);
    hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[s], afh[S0 + s], hid, 0, 0, 0);
  }
}
// the order inside a region of 12 MFMAs + one epilogue group: LDS reads, a head of VALU while they fly, then every MFMA
// followed by its share of the VALU
#define MC_REGION(NDS)                                             \
  __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);             \
  __builtin_amdgcn_sched_group_barrier(0x002, MC_VHEAD, 0);        \
  _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {              \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             \
    __builtin_amdgcn_sched_group_barrier(0x002, MC_VPM, 0);        \
  }                                                                \
  __builtin_amdgcn_sched_barrier(0);

__global__ void __launch_bounds__(256, 2) fused_mlp128_kernel(const MlpArgs q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int m0 = tile * ML_BM + wave * 32;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  const int lrow = lane >> 2;
  const int logical = (lane & 3) ^ ((lrow >> 2) & 3);
  const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
  const unsigned va = (unsigned)lrow * 512u + memchunk * 16;
  const unsigned vw2 = (unsigned)lrow * 2048u + memchunk * 16;
  auto issue_w1 = [&](int j) {                       // W1 rows [32 j, +32) -> ring slot j & 1: 16 DMAs, 4 per wave
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wave * 4 + i, s = idx >> 1, rg = idx & 1;
      ml_dma(lds0 + MC_W1 + (j & 1) * MB_W1BUF + s * MB_W1STAGE + rg * 1024, va, q.w1 + (size_t)(j * MB_HC + rg * 16) * 512 + s * 64);
    }
  };
  auto issue_w2 = [&](int j) {                       // W2 columns [32 j, +32) -> ring slot j & 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wave * 4 + i, s = idx >> 3, rg = idx & 7;
      ml_dma(lds0 + MC_W2 + (j & 1) * MB_W2BUF + s * MB_W2STAGE + rg * 1024, vw2, q.w2 + (size_t)(rg * 16) * 2048 + (size_t)(j * 2 + s) * 64);
    }
  };
  issue_w1(0);
  issue_w1(1);
  if (t < ML_H / 4) *reinterpret_cast<f32x4*>(smem_raw + MC_B1 + t * 16) = *reinterpret_cast<const f32x4*>(q.b1 + t * 4);

  const int hsel = lane >> 5, prow = lane & 31, f = (prow >> 2) & 3;
  h8 afh[8], afl[8];
  {
    const unsigned char* ar = q.a + (size_t)(m0 + prow) * 512 + hsel * 32;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      afh[s] = *reinterpret_cast<const h8*>(ar + s * 64);
      afl[s] = *reinterpret_cast<const h8*>(ar + s * 64 + 16);
    }
  }
  const int off_h = prow * 64 + ((hsel ^ f) << 4), off_l = prow * 64 + (((2 + hsel) ^ f) << 4);
  const unsigned char* bl = smem_raw + MC_B1 + hsel * 16;           // bias quads of chunk j: bl + 128 j + 32 g

  f32x16 out[1][4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[0][jn][r] = 0.0f;
  f32x16 hid, hidn;
  h8 xh[2], xl[2];                                   // the hidden chunk GEMM 2 consumes next, as operand fragments
  u32x2 hi[4], lo[4];
  f32x4 bq[4];

#define MC_TOP()                                   \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  __syncthreads();
#define MC_BIAS(J)                                 \
  _Pragma("unroll") for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(bl + (J) * 128 + g * 32);
#define MC_ZERO(H)                                 \
  _Pragma("unroll") for (int r = 0; r < 16; ++r) H[r] = 0.0f;

  // prologue: GEMM 1 of chunk 0, then iteration 0 without a GEMM 2
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(afh[s]), "+v"(afl[s]));
  __syncthreads();
  MC_ZERO(hid);
  mc_gemm1_steps<0>(smem_raw + MC_W1, off_h, off_l, afh, afl, hid);
  mc_gemm1_steps<4>(smem_raw + MC_W1, off_h, off_l, afh, afl, hid);
  MC_TOP();                                          // every wave has left GEMM 1 of chunk 0
  issue_w1(2);
  issue_w2(0);
  MC_BIAS(0);
  MC_ZERO(hidn);
#pragma unroll
  for (int g = 0; g < 4; ++g) mc_epi_group(hid, g, bq[g], q.unscale1, q.hid_scale, hi[g], lo[g]);
  mc_epi_swap(hi, lo, 0, xh[0], xl[0]);
  mc_epi_swap(hi, lo, 1, xh[1], xl[1]);
  mc_gemm1_steps<0>(smem_raw + MC_W1 + MB_W1BUF, off_h, off_l, afh, afl, hidn);
  mc_gemm1_steps<4>(smem_raw + MC_W1 + MB_W1BUF, off_h, off_l, afh, afl, hidn);
  hid = hidn;

  for (int j = 1; j < MB_NCH - 1; ++j) {
    MC_TOP();                                        // W1_{j+1} and W2_{j-1} are complete; ring slots j & 1 are free
    if (j + 2 < MB_NCH) issue_w1(j + 2);
    issue_w2(j);
    const unsigned char* w2b = smem_raw + MC_W2 + ((j - 1) & 1) * MB_W2BUF;
    const unsigned char* w1b = smem_raw + MC_W1 + ((j + 1) & 1) * MB_W1BUF;
    __builtin_amdgcn_sched_barrier(0);
    // four regions of 12 MFMAs, each carrying one group of chunk j's GELU / split: GEMM 2 of chunk j - 1 (two k16 steps),
    // then GEMM 1 of chunk j + 1 (two halves)
    MC_BIAS(j);
    mc_gemm2_step(w2b, off_h, off_l, xh[0], xl[0], out);
    mc_epi_group(hid, 0, bq[0], q.unscale1, q.hid_scale, hi[0], lo[0]);
    MC_REGION(12)
    mc_gemm2_step(w2b + MB_W2STAGE, off_h, off_l, xh[1], xl[1], out);
    mc_epi_group(hid, 1, bq[1], q.unscale1, q.hid_scale, hi[1], lo[1]);
    mc_epi_swap(hi, lo, 0, xh[0], xl[0]);
    MC_REGION(8)
    MC_ZERO(hidn);
    mc_gemm1_steps<0>(w1b, off_h, off_l, afh, afl, hidn);
    mc_epi_group(hid, 2, bq[2], q.unscale1, q.hid_scale, hi[2], lo[2]);
    MC_REGION(8)
    mc_gemm1_steps<4>(w1b, off_h, off_l, afh, afl, hidn);
    mc_epi_group(hid, 3, bq[3], q.unscale1, q.hid_scale, hi[3], lo[3]);
    mc_epi_swap(hi, lo, 1, xh[1], xl[1]);
    MC_REGION(8)
    hid = hidn;
  }
  // last chunk: no GEMM 1 left
  MC_TOP();
  issue_w2(MB_NCH - 1);
  MC_BIAS(MB_NCH - 1);
  mc_gemm2_step(smem_raw + MC_W2 + ((MB_NCH - 2) & 1) * MB_W2BUF, off_h, off_l, xh[0], xl[0], out);
  mc_gemm2_step(smem_raw + MC_W2 + ((MB_NCH - 2) & 1) * MB_W2BUF + MB_W2STAGE, off_h, off_l, xh[1], xl[1], out);
#pragma unroll
  for (int g = 0; g < 4; ++g) mc_epi_group(hid, g, bq[g], q.unscale1, q.hid_scale, hi[g], lo[g]);
  mc_epi_swap(hi, lo, 0, xh[0], xl[0]);
  mc_epi_swap(hi, lo, 1, xh[1], xl[1]);
  MC_TOP();
  mc_gemm2_step(smem_raw + MC_W2 + ((MB_NCH - 1) & 1) * MB_W2BUF, off_h, off_l, xh[0], xl[0], out);
  mc_gemm2_step(smem_raw + MC_W2 + ((MB_NCH - 1) & 1) * MB_W2BUF + MB_W2STAGE, off_h, off_l, xh[1], xl[1], out);
#undef MC_TOP
#undef MC_BIAS
#undef MC_ZERO
  __syncthreads();                                   // the weight rings are free: the epilogue patches reuse them

  WdConvGemm pe{};
  pe.bias = q.b2; pe.res = q.x; pe.c = q.x; pe.m = q.m; pe.n = ML_C; pe.ldc = ML_C; pe.ldres = ML_C;
  pe.act = WD_ACT_NONE; pe.out_mode = WD_OUT_ROWS; pe.res_alpha = 1.0f; pe.out_scale = 1.0f; pe.range_flag = q.range_flag;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  EpiOctOperands<1, 4> ops;
  ops.load(pe, m0, 0, lane);
  EpiOctWalk<0, 1, 4, WD_ACT_NONE, false, false>::run(pe, q.unscale2, m0, 0, lane, out, patch, ops);
}

}  // namespace

extern "C" int wd_mlp_fused_split(const void* a_split, int64_t rows, int32_t c, int32_t hidden, const void* w1_split,
                                  float w1_unscale, const float* b1, const void* w2_split, float w2_unscale, const float* b2,
                                  float* x, float hid_scale, uint32_t* range_flag, void* stream) {
  if (!a_split || !w1_split || !w2_split || !b1 || !b2 || !x) return WD_ERR_BAD_ARG;
  if (c != ML_C || hidden != ML_H) return WD_ERR_UNSUPPORTED;
  if (rows <= 0 || rows % ML_BM || rows / ML_BM > 0x7fffffffLL) return WD_ERR_UNSUPPORTED;
  if (!wd_aligned16(a_split) || !wd_aligned16(w1_split) || !wd_aligned16(w2_split) || !wd_aligned16(b1) || !wd_aligned16(b2) ||
      !wd_aligned16(x))
    return WD_ERR_BAD_ARG;
  if (!(w1_unscale > 0.f) || !(w2_unscale > 0.f) || !(hid_scale > 0.f)) return WD_ERR_BAD_ARG;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(fused_mlp128_kernel), MC_LDS) != WD_OK) return WD_ERR_LAUNCH;
  MlpArgs q{static_cast<const unsigned char*>(a_split), static_cast<const unsigned char*>(w1_split),
            static_cast<const unsigned char*>(w2_split), b1, b2, x, range_flag, (int)rows, w1_unscale, w2_unscale, hid_scale};
  WD_LAUNCH_GEMM(fused_mlp128_kernel, dim3((unsigned)(rows / ML_BM)), dim3(256), MC_LDS, static_cast<hipStream_t>(stream), q);
  return wd_launch_status();
}
