// split_gemm_mlpw.hip — the ConvNeXt block's pointwise MLP as ONE kernel for the WIDE stages (C = 256 / 512, hidden 4C):
//     x <- x + W2 · GELU(W1 · LN(x) + b1) + b2          (mm_backbone.py:117-124, gamma folded into W2 / b2)
// fp16x3 arithmetic on pre-split operands.  The two-kernel form (split_gemm_p8.hip twice) writes the 4C hidden activation
// to HBM and reads it back — 0.42 GB each way per stage-3 block at WeDetect-Base batch 32, 27 blocks per step — and its
// pwconv1 launch is 40 % epilogue: 105 M GELUs, hi/lo splits and 420 MB of stores with the matrix pipe idle.  Here the
// hidden activation never leaves the CU.
//
// Why not the 128-channel scheme (split_gemm_mlp.hip: a wave owns 32 rows end to end, LayerNorm rows + hidden chunk + output
// tile in registers)?  At C = 512 a wave's 32 x 512 output tile alone is 256 registers and its LayerNorm rows another 256.
// The roles are therefore turned round: the WAVES SPLIT THE COLUMNS, the rows are shared.
//   * workgroup = 4 waves = ONE wave per SIMD with the whole 512-entry register file (256 VGPR + 256 AGPR), 128 rows;
//   * the [128 x C] output tile lives in the four waves' accumulators for the whole kernel: wave w owns columns
//     [C/4 w, + C/4) of all 128 rows (C = 512: 4 x 4 accumulators = 256 registers);
//   * the hidden dimension is walked in chunks of 128 columns.  GEMM 1 of a chunk (K = C): wave w owns hidden columns
//     [32 w, + 32) of all 128 rows (4 accumulators); bias + GELU + hi/lo split on the accumulators where they lie, one
//     v_permlane32_swap per dword makes [hi x8 | lo x8] groups (split_mlp_epi.h), 16 ds_write_b128 put the wave's 128 x 32
//     piece into the hidden-chunk buffer in LDS (64 KB, XOR-swizzled); GEMM 2 of the chunk (K = 128): every wave reads the
//     whole chunk as its activation operand and accumulates its own output columns;
//   * operands.  What the waves SHARE goes through LDS, what is private to a wave does not:
//       - LayerNorm rows (shared: every wave multiplies all 128 rows): K tiles of 64 k (128 rows x 256 B = 32 KB) through a
//         ring of three slots filled by LDS-DMA two tiles ahead, XOR-swizzled on the global side (slot ^ (row & 15):
//         conflict-free ds_read_b128 groups); re-streamed once per chunk from L2 / the Infinity Cache (the panel is 256 KB
//         per workgroup; HBM sees it once);
//       - weights (private: a wave is the only reader of the W1 rows / W2 rows of its columns): straight from global
//         memory into registers, NO LDS — they are stored FRAGMENT-MAJOR at pack time ([32-row block][k16 step][hi | lo]
//         [lane] x 16 B, lib.mlp_wide_pack), so one MFMA operand = one fully coalesced 1 KB wave load, prefetched one K
//         tile (GEMM 1) / one k16 step (GEMM 2) ahead in registers.
//     LDS traffic per MFMA is half the 256 x 256 kernel's (8 ds_read_b128 per 12 MFMAs in GEMM 1, 8 per 48 in GEMM 2: 0.42
//     per MFMA against 0.5, with no weight operand on the LDS path at all), global -> LDS a fifth.
//   * synchronisation: one workgroup barrier per K tile of GEMM 1 (48 MFMAs per wave; ring-slot hand-over) and one per
//     chunk (hidden chunk complete).
// Same halves, same K order per output (hidden columns ascending 16 at a time; lo.hi -> hi.lo -> hi.hi inside a k16 step),
// the same epilogue arithmetic as the two launches it replaces: BIT-IDENTICAL to them (tests/test_gpu_split.py::
// test_fused_mlp_wide_*).
#include "split_epi_oct.h"
#include "split_mlp_epi.h"

namespace {

template <int C>
struct MW {
  static constexpr int H = 4 * C, BM = 128, HC = 128, NCH = H / HC;
  static constexpr int KT = 64, NKT = C / KT;                 // LayerNorm-row K tiles of 64 k: 256 B per row
  static constexpr int XSLOT = BM * 256, NSLOT = 3;           // 32 KB per ring slot
  static constexpr int HOFF = NSLOT * XSLOT;                  // hidden chunk: 128 rows x 512 B ([hi x8 | lo x8] groups)
  static constexpr int LDS = HOFF + BM * 512;                 // 160 KB: the whole CU
  static constexpr int TN2 = C / 128;                         // 32-column output blocks per wave
  static constexpr int S1 = C / 16, S2 = H / 16;              // k16 steps of GEMM 1 / GEMM 2
  static_assert(C % 128 == 0 && NKT >= 2 && LDS <= 160 * 1024, "shape");
  static_assert(4 * 32 * EPI_LDT * 4 <= HOFF, "the epilogue patches reuse the ring");
};

struct MwArgs {
  const unsigned char* a;       // LayerNorm rows as fp16 hi/lo groups [m][C] (row = 4 C bytes)
  const unsigned char* w1;      // fragment-major split weights of pwconv1: [H / 32][C / 16][2][64][16 B]
  const unsigned char* w2;      // fragment-major split weights of pwconv2: [C / 32][H / 16][2][64][16 B]
  const float* b1;              // [H]
  const float* b2;              // [C]
  float* x;                     // residual in / output out, fp32 [m][C]
  unsigned* range_flag;
  int m;
  float unscale1, unscale2, hid_scale;
};

// one 1 KB LDS-DMA: 64 lanes x 16 B from base + voff[lane] to LDS [lds_addr, +1024)
__device__ __forceinline__ void mw_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

template <int C>
__global__ void __launch_bounds__(256) fused_mlp_wide_kernel(const MwArgs q) {
  using P = MW<C>;
  constexpr int TN2 = P::TN2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  int tile = blockIdx.x;
  {
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int m0 = tile * P::BM;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  // ---- LayerNorm-row K tiles: 32 DMAs of 1 KB (4 rows x 256 B) per tile, 8 per wave.  Lane i of a DMA fills row i >> 4,
  // 16-byte slot i & 15 of its 4-row group; the slot holds global chunk slot ^ (row & 15).  row & 15 = 4 (group & 3) + (i >> 4):
  // four lane-offset variants, the group's first row and the K offset ride on the scalar base.
  unsigned la[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) la[c] = (unsigned)(lane >> 4) * (unsigned)(C * 4) + (unsigned)((((lane & 15) ^ (4 * c + (lane >> 4))) & 15) << 4);
  const unsigned char* arow = q.a + (size_t)m0 * (C * 4);
  auto issue_x = [&](int kt, int slot) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int g = wave * 8 + i;
      mw_dma(lds0 + slot * P::XSLOT + g * 1024, la[i & 3], arow + (size_t)(4 * g) * (C * 4) + kt * 256);
    }
  };

  // ---- fragment addresses.  Activation operand (LayerNorm rows / hidden chunk): lane (r, hsel) reads k = 8 hsel .. + 7 of
  // row 32 i + r: logical 16-byte chunk 4 ks + 2 hsel + part (part 0 = hi, 1 = lo), stored at chunk ^ (row & 15).
  const int r = lane & 31, hsel = lane >> 5, f = r & 15;
  const int xb = r * 256 + (((2 * hsel) ^ f) << 4);           // + i * 8192; ^ (ks * 64) for the k16 step, ^ 16 for lo
  const int hb = r * 512 + (((2 * hsel) ^ f) << 4);           // + i * 16384; the same variants
  const unsigned char* hbuf = smem_raw + P::HOFF;

  // ---- weights, fragment-major: one operand = 1 KB contiguous, lane * 16
  const unsigned char* w1p = q.w1 + (size_t)wave * (P::S1 * 2048) + lane * 16;        // chunk j: + 4 j S1 2048; k16 step s: + 2048 s; lo + 1024
  const unsigned char* w2p = q.w2 + (size_t)(TN2 * wave) * (P::S2 * 2048) + lane * 16;  // block jn: + jn S2 2048; step: + 2048 s; lo + 1024

  f32x16 out[4][TN2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
      for (int e = 0; e < 16; ++e) out[i][jn][e] = 0.0f;

  // W1 fragments of one K tile (4 k16 steps): current and next
  h8 wch[4], wcl[4], wnh[4], wnl[4];
  auto load_w1 = [&](const unsigned char* p, h8 (&fh)[4], h8 (&fl)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      fh[s] = *reinterpret_cast<const h8*>(p + s * 2048);
      fl[s] = *reinterpret_cast<const h8*>(p + s * 2048 + 1024);
    }
  };

  // prologue: K tiles 0 and 1 of chunk 0, W1 fragments of tile 0
  issue_x(0, 0);
  issue_x(1, 1);
  load_w1(w1p, wch, wcl);
  int slot = 0;                                                 // ring slot of the K tile about to be consumed

  for (int j = 0; j < P::NCH; ++j) {
    f32x16 g1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) g1[i][e] = 0.0f;
    // bias quads of this wave's hidden columns: channels 8 g + 4 hsel + 0..3 of [128 j + 32 wave, + 32)
    f32x4 bq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[g] = *reinterpret_cast<const f32x4*>(q.b1 + j * P::HC + wave * 32 + 8 * g + 4 * hsel);

    // ================= GEMM 1 of chunk j: K tiles of the LayerNorm rows
    for (int kt = 0; kt < P::NKT; ++kt) {
      // tile (j, kt) has landed (this wave's share: vmcnt; everybody's: the barrier), and every wave has left the tile
      // before it, whose slot the request below overwrites
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      {
        // request the tile two ahead into the slot that just went dead, fetch the W1 fragments of the next tile
        int kt2 = kt + 2, s2 = slot + 2;
        if (kt2 >= P::NKT) kt2 -= P::NKT;
        if (s2 >= P::NSLOT) s2 -= P::NSLOT;
        const bool more = (kt + 2 < P::NKT) || (j + 1 < P::NCH);
        if (more) issue_x(kt2, s2);
        const bool last = (kt + 1 == P::NKT);
        const unsigned char* nx = last ? (j + 1 < P::NCH ? w1p + (size_t)(4 * (j + 1)) * (P::S1 * 2048) : w1p)
                                       : w1p + (size_t)(4 * j) * (P::S1 * 2048) + (kt + 1) * 8192;
        load_w1(nx, wnh, wnl);
      }
      const unsigned char* xs = smem_raw + slot * P::XSLOT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        h8 xh[4], xl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          xh[i] = *reinterpret_cast<const h8*>(xs + i * 8192 + (xb ^ (ks * 64)));
          xl[i] = *reinterpret_cast<const h8*>(xs + i * 8192 + (xb ^ (ks * 64 + 16)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wcl[ks], xh[i], g1[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[ks], xl[i], g1[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wch[ks], xh[i], g1[i], 0, 0, 0);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) { wch[s] = wnh[s]; wcl[s] = wnl[s]; }
      slot = slot + 1 == P::NSLOT ? 0 : slot + 1;
    }

    // W2 fragments of the chunk's first k16 step: they land during the GELU
    h8 vh[TN2], vl[TN2], vnh[TN2], vnl[TN2];
    const unsigned char* w2c = w2p + (size_t)(8 * j) * 2048;
#pragma unroll
    for (int jn = 0; jn < TN2; ++jn) {
      vh[jn] = *reinterpret_cast<const h8*>(w2c + (size_t)jn * (P::S2 * 2048));
      vl[jn] = *reinterpret_cast<const h8*>(w2c + (size_t)jn * (P::S2 * 2048) + 1024);
    }

    // ================= bias + GELU + split of the wave's 128 x 32 piece -> hidden-chunk buffer
    // (every wave passed a K-tile barrier after its GEMM 2 of chunk j - 1: the buffer is free)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 hi[4], lo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) mc_epi_group(g1[i], g, bq[g], q.unscale1, q.hid_scale, hi[g], lo[g]);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        h8 fh8, fl8;
        mc_epi_swap(hi, lo, s, fh8, fl8);
        // lane (r, hsel) holds hidden columns 32 wave + 16 s + 8 hsel + 0..7 of row 32 i + r: 8-column group
        // 4 wave + 2 s + hsel, logical chunks 2 group (hi), 2 group + 1 (lo)
        const int grp = 4 * wave + 2 * s + hsel;
        unsigned char* hp = smem_raw + P::HOFF + (32 * i + r) * 512;
        *reinterpret_cast<h8*>(hp + (((2 * grp) ^ f) << 4)) = fh8;
        *reinterpret_cast<h8*>(hp + (((2 * grp + 1) ^ f) << 4)) = fl8;
      }
    }
    __syncthreads();                                              // the hidden chunk is complete

    // ================= GEMM 2 of chunk j: out += W2[:, chunk] . hidden chunk
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) {
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) {
          vnh[jn] = *reinterpret_cast<const h8*>(w2c + (size_t)jn * (P::S2 * 2048) + (ks + 1) * 2048);
          vnl[jn] = *reinterpret_cast<const h8*>(w2c + (size_t)jn * (P::S2 * 2048) + (ks + 1) * 2048 + 1024);
        }
      }
      h8 hh[4], hl[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        hh[i] = *reinterpret_cast<const h8*>(hbuf + i * 16384 + (hb ^ (ks * 64)));
        hl[i] = *reinterpret_cast<const h8*>(hbuf + i * 16384 + (hb ^ (ks * 64 + 16)));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[jn], hh[i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[jn], hl[i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[jn], hh[i], out[i][jn], 0, 0, 0);
      if (ks + 1 < 8) {
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) { vh[jn] = vnh[jn]; vl[jn] = vnl[jn]; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                                // ring and hidden buffer are free: epilogue patches

  WdConvGemm pe{};
  pe.bias = q.b2; pe.res = q.x; pe.c = q.x; pe.m = q.m; pe.n = C; pe.ldc = C; pe.ldres = C;
  pe.act = WD_ACT_NONE; pe.out_mode = WD_OUT_ROWS; pe.res_alpha = 1.0f; pe.out_scale = 1.0f; pe.range_flag = q.range_flag;
  float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
  const int nw = wave * (C / 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x16 row[1][TN2];
#pragma unroll
    for (int jn = 0; jn < TN2; ++jn) row[0][jn] = out[i][jn];
    EpiOctOperands<1, TN2> ops;
    ops.load(pe, m0 + 32 * i, nw, lane);
    EpiOctWalk<0, 1, TN2, WD_ACT_NONE, false, false>::run(pe, q.unscale2, m0 + 32 * i, nw, lane, row, patch, ops);
  }
}

template <int C>
int launch_mlp_wide(const MwArgs& q, hipStream_t st) {
  using P = MW<C>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(fused_mlp_wide_kernel<C>), P::LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(fused_mlp_wide_kernel<C>, dim3((unsigned)(q.m / P::BM)), dim3(256), P::LDS, st, q);
  return wd_launch_status();
}

}  // namespace

// Weights: FRAGMENT-MAJOR copies of the wd_split_weights buffers (lib.mlp_wide_pack): for a [n][k] weight matrix,
// [n / 32][k / 16][2 (hi, lo)][64 lanes: (k half) * 32 + (row in block)][16 B].
extern "C" int wd_mlp_fused_wide(const void* a_split, int64_t rows, int32_t c, int32_t hidden, const void* w1_frag,
                                 float w1_unscale, const float* b1, const void* w2_frag, float w2_unscale, const float* b2,
                                 float* x, float hid_scale, uint32_t* range_flag, void* stream) {
  if (!a_split || !w1_frag || !w2_frag || !b1 || !b2 || !x) return WD_ERR_BAD_ARG;
  if ((c != 256 && c != 512) || hidden != 4 * c) return WD_ERR_UNSUPPORTED;
  if (rows <= 0 || rows % 128 || rows / 128 > 0x7fffffffLL) return WD_ERR_UNSUPPORTED;
  if (!wd_aligned16(a_split) || !wd_aligned16(w1_frag) || !wd_aligned16(w2_frag) || !wd_aligned16(b1) || !wd_aligned16(b2) ||
      !wd_aligned16(x))
    return WD_ERR_BAD_ARG;
  if (!(w1_unscale > 0.f) || !(w2_unscale > 0.f) || !(hid_scale > 0.f)) return WD_ERR_BAD_ARG;
  MwArgs q{static_cast<const unsigned char*>(a_split), static_cast<const unsigned char*>(w1_frag),
           static_cast<const unsigned char*>(w2_frag), b1, b2, x, range_flag, (int)rows, w1_unscale, w2_unscale, hid_scale};
  hipStream_t st = static_cast<hipStream_t>(stream);
  return c == 512 ? launch_mlp_wide<512>(q, st) : launch_mlp_wide<256>(q, st);
}
