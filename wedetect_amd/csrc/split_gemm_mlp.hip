// split_gemm_mlp.hip — the ConvNeXt block's pointwise MLP as ONE kernel for the narrow stage (C = 128, hidden 512):
//     x <- x + W2 · GELU(W1 · LN(x) + b1) + b2          (mm_backbone.py:117-124, gamma folded into W2 / b2)
// fp16x3 arithmetic on pre-split operands, like the two-kernel form (wd_conv_gemm_split twice), but the 4C hidden
// activation never leaves the CU.  At WeDetect-Base batch 32 the stage-1 hidden tensor is 1.68 GB per block: the
// two-kernel form writes it (the pwconv1 launch is an epilogue: 420 M GELUs + 1.7 GB of stores, 710-740 us) and reads
// it back (pwconv2: 2.5 GB of HBM traffic, 515 us).  Here a workgroup owns 128 rows:
//   * the LayerNorm rows [128 x 128] (fp16 hi/lo groups, 64 KB) are DMA'd into LDS once; every wave then holds the MFMA
//     fragments of its 64 rows in registers (128 VGPRs) for all eight chunks — GEMM 1 reads only weight fragments from LDS;
//   * the hidden dimension is walked in chunks of 64 columns: GEMM 1 (K = 128: 8 k16 steps) -> accumulators ->
//     bias + GELU + hi/lo split IN REGISTERS -> written straight into LDS in the operand layout of GEMM 2 (a lane holds 4
//     channels of a pixel = half of an 8-channel hi/lo group: two ds_write_b64 per group, no transpose) -> GEMM 2
//     (K = 64: 4 k16 steps) accumulates the [128 x 128] output tile in registers across all eight chunks;
//   * the chunk's weights — W1 rows [64 x 128] and W2 columns [128 x 64], 32 KB each — come by LDS-DMA one chunk ahead:
//     W2_j and W1_{j+1} are requested when GEMM 1 of chunk j has finished reading W1_j, and land under the GELU epilogue /
//     GEMM 2; bias vectors are loaded before the requests so that no compiler-inserted wait drains the DMA queue;
//   * LDS: 64 (rows) + 32 (W1) + 32 (hidden chunk) + 32 (W2) = 160 KB, one workgroup of four waves per CU; a wave owns
//     64 rows x 32 hidden columns in GEMM 1 and 64 x 64 outputs in GEMM 2 (32 + 64 accumulator registers);
//   * three barriers per chunk (W1 landed | W1 consumed, request next | hidden chunk + W2 ready).
// Same halves, same K order per output (hidden columns ascending, 16 at a time), the same epilogue arithmetic as the
// two kernels it replaces: BIT-IDENTICAL (tests/test_gpu_split.py::test_fused_mlp_*).
#include "split_epi_oct.h"

namespace {

constexpr int ML_C = 128, ML_H = 512, ML_BM = 128, ML_HC = 64;
constexpr int ML_ASTAGE = ML_BM * 64, ML_W1STAGE = ML_HC * 64, ML_W2STAGE = ML_C * 64;
constexpr int ML_A = 0, ML_W1 = ML_A + 8 * ML_ASTAGE, ML_HID = ML_W1 + 8 * ML_W1STAGE, ML_W2 = ML_HID + 4 * ML_ASTAGE;
constexpr int ML_LDS = ML_W2 + 4 * ML_W2STAGE;
static_assert(ML_LDS == 160 * 1024, "the fused MLP uses the whole LDS");

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

__global__ void __launch_bounds__(256) fused_mlp128_kernel(const MlpArgs q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int m0 = tile * ML_BM;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  // lane part of every DMA source: row (lane >> 2) of a 16-row group + the 16-byte chunk that belongs in physical slot
  // (lane & 3) of that row (slots XOR-swizzled by (row / 4) & 3; group starts are multiples of 16, so the swizzle term
  // only depends on the lane)
  const int lrow = lane >> 2;
  const int logical = (lane & 3) ^ ((lrow >> 2) & 3);
  const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
  const unsigned va = (unsigned)lrow * 512u + memchunk * 16;          // activation rows and W1 rows: 512 B pitch
  const unsigned vw2 = (unsigned)lrow * 2048u + memchunk * 16;        // W2 rows: 2048 B pitch

  // W1 chunk j: 8 stages x 4 row groups = 32 DMAs, 8 per wave; W2 chunk j: 4 stages x 8 row groups = 32 DMAs, 8 per wave
  auto issue_w1 = [&](int j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = wave * 8 + i, s = idx >> 2, rg = idx & 3;
      ml_dma(lds0 + ML_W1 + s * ML_W1STAGE + rg * 1024, va, q.w1 + (size_t)(j * ML_HC + rg * 16) * 512 + s * 64);
    }
  };
  auto issue_w2 = [&](int j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = wave * 8 + i, s = idx >> 3, rg = idx & 7;
      ml_dma(lds0 + ML_W2 + s * ML_W2STAGE + rg * 1024, vw2, q.w2 + (size_t)(rg * 16) * 2048 + (size_t)(j * 4 + s) * 64);
    }
  };
  // the tile's rows: 8 stages x 8 row groups = 64 DMAs, 16 per wave
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = wave * 16 + i, s = idx >> 3, rg = idx & 7;
    ml_dma(lds0 + ML_A + s * ML_ASTAGE + rg * 1024, va, q.a + (size_t)(m0 + rg * 16) * 512 + s * 64);
  }
  issue_w1(0);

  // fragment addresses (byte offsets inside a k16 stage)
  const int hsel = lane >> 5;
  int aoff_h[2], aoff_l[2], w1off_h, w1off_l, w2off_h[2], w2off_l[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + (lane & 31), f = (row >> 2) & 3;
    aoff_h[i] = row * 64 + ((hsel ^ f) << 4);
    aoff_l[i] = row * 64 + (((2 + hsel) ^ f) << 4);
  }
  {
    const int row = wn * 32 + (lane & 31), f = (row >> 2) & 3;
    w1off_h = row * 64 + ((hsel ^ f) << 4);
    w1off_l = row * 64 + (((2 + hsel) ^ f) << 4);
  }
#pragma unroll
  for (int jn = 0; jn < 2; ++jn) {
    const int row = wn * 64 + jn * 32 + (lane & 31), f = (row >> 2) & 3;
    w2off_h[jn] = row * 64 + ((hsel ^ f) << 4);
    w2off_l[jn] = row * 64 + (((2 + hsel) ^ f) << 4);
  }
  // where this lane's hidden values go: row (64 wm + 32 i + lane & 31), chunk column 32 wn + 8 g + 4 (lane >> 5) + 0..3
  // = k16 stage 2 wn + (g >> 1), 8-channel group g & 1, half (lane >> 5) of the group
  int hid_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 64 + i * 32 + (lane & 31);
    hid_off[i] = ML_HID + (2 * wn) * ML_ASTAGE + row * 64 + hsel * 8;
  }
  const int hid_f = ((lane & 31) >> 2) & 3;          // (row >> 2) & 3: rows are 32 i + 64 wm + (lane & 31)

  f32x16 out[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) out[i][jn][r] = 0.0f;

  h8 afh[8][2], afl[8][2];                           // this wave's 64 rows x K = 128 as MFMA fragments, loaded once
  for (int j = 0; j < ML_H / ML_HC; ++j) {
    // the chunk's bias quads BEFORE any DMA of this chunk is issued, and pinned as "ready" right after the wait below: a
    // load (or the compiler's own wait for one) placed behind the DMAs would drain the whole queue — loads retire in order
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(q.b1 + j * ML_HC + wn * 32 + 8 * g + 4 * hsel);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // rows (first chunk) + W1_j (+ the bias loads) have landed
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[g]));
    __syncthreads();
    if (j == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const unsigned char* as = smem_raw + ML_A + s * ML_ASTAGE;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          afh[s][i] = *reinterpret_cast<const h8*>(as + aoff_h[i]);
          afl[s][i] = *reinterpret_cast<const h8*>(as + aoff_l[i]);
        }
      }
    }

    // ---- GEMM 1: hidden chunk [64 rows of this wave x 32 columns], K = 128
    f32x16 hid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[i][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const unsigned char* ws = smem_raw + ML_W1 + s * ML_W1STAGE;
      const h8 wh = *reinterpret_cast<const h8*>(ws + w1off_h), wl = *reinterpret_cast<const h8*>(ws + w1off_l);
#pragma unroll
      for (int i = 0; i < 2; ++i) hid[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, afh[s][i], hid[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) hid[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, afl[s][i], hid[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) hid[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, afh[s][i], hid[i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                 // W1_j consumed by every wave; GEMM 2 of chunk j - 1 is over too
    __builtin_amdgcn_sched_barrier(0);
    issue_w2(j);
    if (j + 1 < ML_H / ML_HC) issue_w1(j + 1);
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue 1 in registers: bias + GELU (+ range scale) + hi/lo split, straight into GEMM 2's operand layout
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = {hid[i][4 * g], hid[i][4 * g + 1], hid[i][4 * g + 2], hid[i][4 * g + 3]};
        if (q.range_flag) {
          if (wd_any_nonfinite4(v[0], v[1], v[2], v[3])) *q.range_flag = 1u;
        }
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = sact<WD_ACT_GELU>(fmaf(v[r], q.unscale1, bq[g][r]));
        if (q.hid_scale != 1.0f) o = o * q.hid_scale;
        u32x2 hi, lo;
        split4(o, hi, lo);
        unsigned char* dst = smem_raw + hid_off[i] + (g >> 1) * ML_ASTAGE;
        const int sh = (g & 1) ^ hid_f, sl = (2 + (g & 1)) ^ hid_f;
        *reinterpret_cast<u32x2*>(dst + (sh << 4)) = hi;
        *reinterpret_cast<u32x2*>(dst + (sl << 4)) = lo;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (j + 1 < ML_H / ML_HC) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // W2_j landed; W1_{j+1} may still fly
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // hidden chunk written by all four waves, W2_j complete

    // ---- GEMM 2: out[64 x 64 of this wave] += hidden chunk [64 x 64] . W2_j^T
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const unsigned char* hs = smem_raw + ML_HID + s * ML_ASTAGE;
      const unsigned char* ws = smem_raw + ML_W2 + s * ML_W2STAGE;
      h8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xh[i] = *reinterpret_cast<const h8*>(hs + aoff_h[i]);
        xl[i] = *reinterpret_cast<const h8*>(hs + aoff_l[i]);
      }
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) {
        wh[jn] = *reinterpret_cast<const h8*>(ws + w2off_h[jn]);
        wl[jn] = *reinterpret_cast<const h8*>(ws + w2off_l[jn]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[jn], xh[i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xl[i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xh[i], out[i][jn], 0, 0, 0);
    }
  }
  __syncthreads();                                   // every wave is done with the LDS operands: the patches may reuse them

  // ---- epilogue 2: + b2 + residual, fp32 rows in place (the row / 8-channel epilogue of the pre-split kernels)
  WdConvGemm pe{};
  pe.bias = q.b2; pe.res = q.x; pe.c = q.x; pe.m = q.m; pe.n = ML_C; pe.ldc = ML_C; pe.ldres = ML_C;
  pe.act = WD_ACT_NONE; pe.out_mode = WD_OUT_ROWS; pe.res_alpha = 1.0f; pe.out_scale = 1.0f; pe.range_flag = q.range_flag;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  EpiOctWalk<0, 2, 2, WD_ACT_NONE, false, false>::run(pe, q.unscale2, m0 + wm * 64, wn * 64, lane, out, patch);
}

// ---------------------------------------------------------------------------------------
// Variant B: two workgroups per CU.  A wave owns 32 rows end to end (its LayerNorm fragments in registers, its hidden chunk
// in a private 4 KB LDS patch, a [32 x 128] output tile), so the only thing the four waves share is the weight chunks —
// hidden chunks of 32 columns, W1 [32 x 128] and W2 [128 x 32] (16 KB each) double-buffered, ONE barrier per chunk.
// 80 KB of LDS: with two workgroups (eight waves) on a CU the GELU epilogue (VALU) of one wave runs under the MFMAs of
// another — the one-workgroup form above serialises them (a wave per SIMD, three barriers per chunk).
// ---------------------------------------------------------------------------------------
constexpr int MB_HC = 32, MB_NCH = ML_H / MB_HC;
constexpr int MB_W1STAGE = MB_HC * 64, MB_W2STAGE = ML_C * 64, MB_HSTAGE = 32 * 64;
constexpr int MB_W1 = 0, MB_W1BUF = 8 * MB_W1STAGE, MB_W2 = MB_W1 + 2 * MB_W1BUF, MB_W2BUF = 2 * MB_W2STAGE;
constexpr int MB_HID = MB_W2 + 2 * MB_W2BUF, MB_HIDW = 2 * MB_HSTAGE, MB_LDS = MB_HID + 4 * MB_HIDW;
static_assert(MB_LDS == 80 * 1024, "two workgroups per CU");
static_assert(4 * 32 * EPI_LDT * 4 <= MB_HID, "the epilogue patches reuse the weight buffers");

__global__ void __launch_bounds__(256, 2) fused_mlp128_rows_kernel(const MlpArgs q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int m0 = tile * ML_BM + wave * 32;           // this wave's rows
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  const int lrow = lane >> 2;
  const int logical = (lane & 3) ^ ((lrow >> 2) & 3);
  const int memchunk = ((logical & 1) << 1) | ((logical >> 1) & 1);
  const unsigned va = (unsigned)lrow * 512u + memchunk * 16;
  const unsigned vw2 = (unsigned)lrow * 2048u + memchunk * 16;
  // chunk j -> buffer j & 1: W1 rows [32 j, +32) as 8 stages x 2 row groups, W2 columns [32 j, +32) as 2 stages x 8 row
  // groups: 32 DMAs of 1 KB, 8 per wave
  auto issue = [&](int j) {
    const int b = j & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wave * 4 + i, s = idx >> 1, rg = idx & 1;
      ml_dma(lds0 + MB_W1 + b * MB_W1BUF + s * MB_W1STAGE + rg * 1024, va, q.w1 + (size_t)(j * MB_HC + rg * 16) * 512 + s * 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = wave * 4 + i, s = idx >> 3, rg = idx & 7;
      ml_dma(lds0 + MB_W2 + b * MB_W2BUF + s * MB_W2STAGE + rg * 1024, vw2, q.w2 + (size_t)(rg * 16) * 2048 + (size_t)(j * 2 + s) * 64);
    }
  };
  issue(0);

  const int hsel = lane >> 5, prow = lane & 31, f = (prow >> 2) & 3;
  // the wave's 32 LayerNorm rows as MFMA fragments, straight from memory: 8 k16 steps x (hi, lo) x 16 B per lane
  h8 afh[8], afl[8];
  {
    // memory: per k16 step [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15], 16 B each; lane half (lane >> 5) holds k 8 (lane >> 5) ..
    const unsigned char* ar = q.a + (size_t)(m0 + prow) * 512 + hsel * 32;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      afh[s] = *reinterpret_cast<const h8*>(ar + s * 64);
      afl[s] = *reinterpret_cast<const h8*>(ar + s * 64 + 16);
    }
  }
  const int off_h = prow * 64 + ((hsel ^ f) << 4), off_l = prow * 64 + (((2 + hsel) ^ f) << 4);   // fragment of row (lane & 31)
  unsigned char* hidw = smem_raw + MB_HID + wave * MB_HIDW;
  const int hid_off = prow * 64 + hsel * 8;

  f32x16 out[1][4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[0][jn][r] = 0.0f;

  // bias quads of the NEXT chunk are requested a chunk ahead, behind that chunk's DMAs: loads retire in order, so the wait
  // at the top of a chunk (everything requested during the previous one) covers them and nothing else ever waits on memory
  f32x4 bqn[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bqn[g] = *reinterpret_cast<const f32x4*>(q.b1 + 8 * g + 4 * hsel);
  bool bad = false;

  for (int j = 0; j < MB_NCH; ++j) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bq[g] = bqn[g];
      asm volatile("" : "+v"(bq[g]));
    }
    if (j == 0) {
#pragma unroll
      for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(afh[s]), "+v"(afl[s]));
    }
    __syncthreads();                                 // chunk j's weights are complete; every wave has left chunk j - 1
    if (j + 1 < MB_NCH) {
      issue(j + 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) bqn[g] = *reinterpret_cast<const f32x4*>(q.b1 + (j + 1) * MB_HC + 8 * g + 4 * hsel);
    }
    const unsigned char* w1b = smem_raw + MB_W1 + (j & 1) * MB_W1BUF;
    const unsigned char* w2b = smem_raw + MB_W2 + (j & 1) * MB_W2BUF;

    // ---- GEMM 1: [32 rows x 32 hidden columns], K = 128
    f32x16 hid;
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const h8 wh = *reinterpret_cast<const h8*>(w1b + s * MB_W1STAGE + off_h), wl = *reinterpret_cast<const h8*>(w1b + s * MB_W1STAGE + off_l);
      hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, afh[s], hid, 0, 0, 0);
      hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, afl[s], hid, 0, 0, 0);
      hid = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, afh[s], hid, 0, 0, 0);
    }

    // ---- epilogue 1: bias + GELU (+ range scale) + hi/lo split into the wave's own patch, GEMM 2's operand layout
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {hid[4 * g], hid[4 * g + 1], hid[4 * g + 2], hid[4 * g + 3]};
      bad |= wd_any_nonfinite4(v[0], v[1], v[2], v[3]);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = sact<WD_ACT_GELU>(fmaf(v[r], q.unscale1, bq[g][r]));
      if (q.hid_scale != 1.0f) o = o * q.hid_scale;
      u32x2 hi, lo;
      split4(o, hi, lo);
      unsigned char* dst = hidw + (g >> 1) * MB_HSTAGE + hid_off;
      *reinterpret_cast<u32x2*>(dst + (((g & 1) ^ f) << 4)) = hi;
      *reinterpret_cast<u32x2*>(dst + (((2 + (g & 1)) ^ f) << 4)) = lo;
    }
    // the patch is written and read by this wave only: LDS operations of a wave execute in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- GEMM 2: out[32 x 128] += hidden chunk [32 x 32] . W2_j^T
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const h8 xh = *reinterpret_cast<const h8*>(hidw + s * MB_HSTAGE + off_h), xl = *reinterpret_cast<const h8*>(hidw + s * MB_HSTAGE + off_l);
      h8 wh[4], wl[4];
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        wh[jn] = *reinterpret_cast<const h8*>(w2b + s * MB_W2STAGE + jn * 2048 + off_h);
        wl[jn] = *reinterpret_cast<const h8*>(w2b + s * MB_W2STAGE + jn * 2048 + off_l);
      }
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[jn], xh, out[0][jn], 0, 0, 0);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xl, out[0][jn], 0, 0, 0);
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) out[0][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[jn], xh, out[0][jn], 0, 0, 0);
    }
  }
  if (q.range_flag && bad) *q.range_flag = 1u;
  __syncthreads();                                   // the weight buffers are free: the epilogue patches reuse them

  WdConvGemm pe{};
  pe.bias = q.b2; pe.res = q.x; pe.c = q.x; pe.m = q.m; pe.n = ML_C; pe.ldc = ML_C; pe.ldres = ML_C;
  pe.act = WD_ACT_NONE; pe.out_mode = WD_OUT_ROWS; pe.res_alpha = 1.0f; pe.out_scale = 1.0f; pe.range_flag = q.range_flag;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  EpiOctWalk<0, 1, 4, WD_ACT_NONE, false, false>::run(pe, q.unscale2, m0, 0, lane, out, patch);
}

int g_mlp_variant = 1;

}  // namespace

extern "C" void wd_debug_mlp_variant(int v) { g_mlp_variant = v; }

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
  static WdAttrOnce attr, attr_b;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(fused_mlp128_kernel), ML_LDS) != WD_OK) return WD_ERR_LAUNCH;
  if (wd_set_max_lds(attr_b, reinterpret_cast<const void*>(fused_mlp128_rows_kernel), MB_LDS) != WD_OK) return WD_ERR_LAUNCH;
  MlpArgs q{static_cast<const unsigned char*>(a_split), static_cast<const unsigned char*>(w1_split),
            static_cast<const unsigned char*>(w2_split), b1, b2, x, range_flag, (int)rows, w1_unscale, w2_unscale, hid_scale};
  if (g_mlp_variant == 1)
    WD_LAUNCH_GEMM(fused_mlp128_rows_kernel, dim3((unsigned)(rows / ML_BM)), dim3(256), MB_LDS, static_cast<hipStream_t>(stream), q);
  else
    WD_LAUNCH_GEMM(fused_mlp128_kernel, dim3((unsigned)(rows / ML_BM)), dim3(256), ML_LDS, static_cast<hipStream_t>(stream), q);
  return wd_launch_status();
}
