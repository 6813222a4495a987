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
#include <stdlib.h>
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
  // persistent form (PERSIST): one workgroup per CU walks a contiguous range of (row block, hidden chunk) units
  unsigned* flag;               // [workgroup]: 1 = the workgroup's parked accumulators are complete; reset by the consumer
  float* park;                  // [workgroup][4 TN2 accumulators x 4 quads][256 threads] x 16 B
  int nblk;                     // row blocks of 128
  // round 6 (FOLD): the block's LayerNorm folded into pwconv1 — a = the RAW depthwise output (wd_dwconv7_stats), w1 = W1 gamma,
  // b1 = W1 beta + b1, ln_u = (W1 gamma) 1, ln_stats = per-row (mean, rstd): hidden = GELU(rstd (W'd - mean u) + v), the
  // epilogue of the two-launch fold (epi_lds_tile_csplit), element for element
  const float* ln_stats;        // [m][2]
  const float* ln_u;            // [H]
};

// one 1 KB LDS-DMA: 64 lanes x 16 B from base + voff[lane] to LDS [lds_addr, +1024).
// s_nop 4, not 0: a VMEM instruction that reads an SGPR written by a VALU instruction needs 5 wait states (gfx9 hazard
// "VALU writes SGPR -> VMEM reads that SGPR"), the hazard recognizer does not look inside inline asm, and the compiler DOES put
// VALU writes of the base right in front of these statements — v_readlane restores of spilled SGPRs.  Found on this kernel:
// the 256-channel build (31 SGPR spills) computed whole row blocks from a stale base on its first, cold launch, the persistent
// 512-channel build (615 spills) faulted.  (s_mov m0 itself needs 1 wait state before the DMA reads M0.)
// Two defences: (1) the scalar bases are re-derived inside the chunk loop from opaque copies of the kernel arguments (MW_OPAQUE
// below), so hipcc no longer hoists ~100 loop-invariant address pairs out of the loop and spills them — no spill, no
// v_readlane in front of the asm; (2) scripts/check_sgpr_vmem_hazard.py runs over the ISA of every build (wedetect_amd/build.py)
// and fails it if a VALU write of an asm VMEM instruction's SGPR operand sits less than 5 wait states ahead.  The persistent
// form, whose piece loop keeps enough scalars alive that hipcc still spills SGPRs, pads every asm VMEM instruction with
// s_nop 4 instead (SAFE = true; ~5 % of its time).
#define MW_OPAQUE(p) asm volatile("" : "+s"(p))
template <bool SAFE>
__device__ __forceinline__ void mw_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  if constexpr (SAFE)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

// Every VMEM operation of the main loop is issued from inline assembly and retired with a COUNTED s_waitcnt: with one wave
// per SIMD a full vmcnt(0) drain per K tile (the first form of this kernel, 840 us) exposes the whole L2 / Infinity-Cache
// latency of whatever was requested last — timing-only builds priced the LayerNorm-row DMA + its wait at 210 us and the
// weight loads at 140 us of the launch.  The compiler cannot do the counting: it does not see the LDS-DMA (inline asm, because
// a builtin DMA makes hipcc wait vmcnt(0) before every ds_read), so its own waits for the register loads would also retire
// the youngest DMAs.  Register loads therefore come as asm too; their destinations are tied ("+v") into the wait statement
// that retires them, so no consumer can be scheduled ahead of it.
#ifdef MW_EXPERIMENT_LGKM
#define MW_LGKM0 "s_waitcnt lgkmcnt(0)"
#else
#define MW_LGKM0 ""
#endif
#define MW_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile(MW_LGKM0 ::: "memory");      \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

// hi and lo halves of one fragment-major operand: two 1 KB wave loads, 16 B per lane
template <bool SAFE>
__device__ __forceinline__ void mw_wload(h8& hi, h8& lo, unsigned lane16, const unsigned char* base) {
  if constexpr (SAFE)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(hi), "=&v"(lo) : "v"(lane16), "s"(base) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                 : "=&v"(hi), "=&v"(lo) : "v"(lane16), "s"(base) : "memory");
}
template <bool SAFE>
__device__ __forceinline__ void mw_bload(f32x4& d, unsigned voff, const float* base) {
  if constexpr (SAFE)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(base) : "memory");
  else
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(d) : "v"(voff), "s"(base) : "memory");
}
#define MW_WAIT8_(N, A0, A1, A2, A3, A4, A5, A6, A7) \
  asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7) :: "memory")
#ifdef MW_EXPERIMENT_ALL0
#define MW_WAIT8(N, ...) MW_WAIT8_(0, __VA_ARGS__)
#else
#define MW_WAIT8(N, ...) MW_WAIT8_(N, __VA_ARGS__)
#endif

// ABL: timing-only ablations (WRONG results; compiled only with -DWD_DEBUG_ABLATIONS, selected by $WD_MLPW_ABL): 1 = no
// GELU arithmetic, 2 = no LayerNorm-row DMA, 4 = no K-tile barriers, 8 = no weight loads after the first
//
// PERSIST: 400 row blocks on 256 CUs are two rounds with the second one 56 % full (measured: 32768 rows 397 us, 65536 rows
// 787 us, 51200 rows 706 us = 1.79 rounds).  The persistent form deals (row block, hidden chunk) units instead: workgroup b
// (XCD b % 8, slot b / 8) takes a contiguous range of its XCD's units, 1.5625 row blocks' worth.  A range starts and / or
// ends inside a row block.  The head of a block (chunks [0, e)) is computed FIRST, the raw output accumulators are parked in
// the workspace and a flag is published (agent-scope release); whole blocks follow; LAST comes the tail [o, 16) of the block
// the previous workgroup of the XCD chain began: wait for its flag (set long before), reload the accumulators, continue the
// chunk loop where it stopped — the same MFMA chain per output, carried by two CUs: bit-identical.  (The scheme of the
// persistent 256 x 256 kernel, split_gemm_p8.hip.)
template <int C, int ABL = 0, bool PERSIST = false, bool FOLD = false>
__global__ void __launch_bounds__(256) fused_mlp_wide_kernel(const MwArgs q) {
  using P = MW<C>;
  constexpr int TN2 = P::TN2, NKT = P::NKT;
  static_assert(NKT % 2 == 0 && NKT >= 4, "K tiles alternate between two W1 fragment buffers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  // ---- work assignment: units [first_blk * NCH + seg_o, last_blk * NCH + seg_e)
  int first_blk, seg_o, last_blk, seg_e;
  if (PERSIST) {
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, per = gridDim.x >> 3;
    const long long t0 = (long long)q.nblk * xcd / 8, t1 = (long long)q.nblk * (xcd + 1) / 8;
    const long long units = (t1 - t0) * P::NCH;
    const long long u0 = t0 * P::NCH + units * slot / per, u1 = t0 * P::NCH + units * (slot + 1) / per;
    first_blk = (int)(u0 / P::NCH); seg_o = (int)(u0 - (long long)first_blk * P::NCH);
    last_blk = (int)(u1 / P::NCH); seg_e = (int)(u1 - (long long)last_blk * P::NCH);
  } else {
    int tile = blockIdx.x;
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int qq = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    first_blk = tile; seg_o = 0; last_blk = tile + 1; seg_e = 0;
  }
  // pieces in execution order: [head of the last block] [whole blocks] [tail of the first block]
  const int has_head = seg_e > 0, has_tail = seg_o > 0;
  const int full0 = first_blk + has_tail, nfull = last_blk - full0;
  const int nseg = has_head + nfull + has_tail;
  unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;

  // ---- LayerNorm-row K tiles: 32 DMAs of 1 KB (4 rows x 256 B) per tile, 8 per wave.  Lane i of a DMA fills row i >> 4,
  // 16-byte slot i & 15 of its 4-row group; the slot holds global chunk slot ^ (row & 15).  row & 15 = 4 (group & 3) + (i >> 4):
  // four lane-offset variants, the group's first row and the K offset ride on the scalar base.
  unsigned la[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) la[c] = (unsigned)(lane >> 4) * (unsigned)(C * 4) + (unsigned)((((lane & 15) ^ (4 * c + (lane >> 4))) & 15) << 4);
  const unsigned char* arow = q.a;                                // + m0 rows: set per piece
  auto issue_x = [&](int kt, int slot) {
    if (ABL & 2) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int g = wave * 8 + i;
      mw_dma<PERSIST>(lds0 + slot * P::XSLOT + g * 1024, la[i & 3], arow + (size_t)(4 * g) * (C * 4) + kt * 256);
    }
  };

  // ---- fragment addresses.  Activation operand (LayerNorm rows / hidden chunk): lane (r, hsel) reads k = 8 hsel .. + 7 of
  // row 32 i + r: logical 16-byte chunk 4 ks + 2 hsel + part (part 0 = hi, 1 = lo), stored at chunk ^ (row & 15).
  const int r = lane & 31, hsel = lane >> 5, f = r & 15;
  const int xb = r * 256 + (((2 * hsel) ^ f) << 4);           // + i * 8192; ^ (ks * 64) for the k16 step, ^ 16 for lo
  const int hb = r * 512 + (((2 * hsel) ^ f) << 4);           // + i * 16384; the same variants
  const unsigned char* hbuf = smem_raw + P::HOFF;

  // ---- weights, fragment-major: one operand = 1 KB contiguous, lane * 16
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned char* w1w_ = q.w1 + (size_t)wave * (P::S1 * 2048);            // chunk j: + 4 j S1 2048; K tile kt: + 8192 kt; step s: + 2048 s
  const unsigned char* w2w_ = q.w2 + (size_t)(TN2 * wave) * (P::S2 * 2048);    // block jn: + jn S2 2048; k16 step S of the hidden axis: + 2048 S
  const unsigned char *w1w = w1w_, *w2w = w2w_;
  const unsigned bvoff = (unsigned)hsel * 16u;
  const float* b1p = q.b1 + wave * 32;                                 // bias quads: b1 + 128 j + 32 wave + 8 g + 4 hsel
  const float* u1p = FOLD ? q.ln_u + wave * 32 : nullptr;              // FOLD: the u quads, same addressing

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  for (int sg = 0; sg < nseg; ++sg) {
  const bool is_head = has_head && sg == 0;
  const bool is_tail = has_tail && sg == nseg - 1;
  const int blk = is_head ? last_blk : (is_tail ? first_blk : full0 + sg - has_head);
  const int c0 = is_tail ? seg_o : 0, c1 = is_head ? seg_e : P::NCH;      // hidden chunks [c0, c1) of this piece
  const int m0 = blk * P::BM;
  arow = q.a + (size_t)m0 * (C * 4);
  MW_BARRIER();                                                   // the previous piece's epilogue is done with the LDS

  f32x16 out[4][TN2];
  if (PERSIST && is_tail) {
    // resume from the accumulators the previous workgroup of the chain parked at the START of its work
    const int prev = blockIdx.x - 8;
    if (t == 0) {
      while (__hip_atomic_load(q.flag + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(8);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    // uniform base + (thread * 16): the 64 quads of a lane are 4 KB apart, past the immediate-offset range — as per-lane
    // 64-bit addresses hipcc hoists all of them out of the piece loop and spills them
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(q.park) + (size_t)prev * (4 * TN2 * 4 * 4096);
    const unsigned toff = (unsigned)t * 16u;
    // one row block of accumulators (TN2 x 64 B per lane) at a time: all 4 TN2 at once would need as many VGPRs to land in
    // as the AGPRs they end up in, and hipcc spills (the persistent 256 x 256 kernel met the same).  ONE running scalar
    // base, bumped by 4 KB per quad behind an opaque copy: 64 distinct address expressions get hoisted out of the piece loop
    // and spilled
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v[TN2][4];
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          MW_OPAQUE(sb);
          v[jn][g] = *reinterpret_cast<const f32x4*>(sb + toff);
          sb += 4096;
        }
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          out[i][jn][4 * g] = v[jn][g][0]; out[i][jn][4 * g + 1] = v[jn][g][1]; out[i][jn][4 * g + 2] = v[jn][g][2]; out[i][jn][4 * g + 3] = v[jn][g][3];
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn) asm volatile("" : "+a"(out[i][jn]));   // pin the block into its accumulator registers before the next one is requested
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) __hip_atomic_store(q.flag + prev, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int e = 0; e < 16; ++e) out[i][jn][e] = 0.0f;
  }

  h8 wh[2][4], wl[2][4];          // W1 fragments of K tiles t (buffer t & 1): refilled for tile t + 2 step by step
  h8 vh[2][TN2], vl[2][TN2];      // W2 fragments of k16 step S (buffer S & 1): refilled for step S + 2
  f32x4 bq[4];
  f32x4 uq[4] = {};               // FOLD: u quads of the chunk, requested with the bias quads (4 more VMEM operations per chunk)
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  f32x2_ st[4] = {};              // FOLD: (mean, rstd) of this lane's four rows 32 i + r of the row block

  auto load_w1_tile = [&](int j, int kt, int buf) {              // all four steps of one K tile
    if ((ABL & 8) && (j > c0 || kt > 1)) return;
    const unsigned char* b = w1w + (size_t)(4 * j) * (P::S1 * 2048) + kt * 8192;
#pragma unroll
    for (int s = 0; s < 4; ++s) mw_wload<PERSIST>(wh[buf][s], wl[buf][s], lane16, b + s * 2048);
  };
  auto load_w2_step = [&](int S, int buf) {
    if ((ABL & 8) && S > 1) return;
    const unsigned char* b = w2w + (size_t)S * 2048;
#pragma unroll
    for (int jn = 0; jn < TN2; ++jn) mw_wload<PERSIST>(vh[buf][jn], vl[buf][jn], lane16, b + (size_t)jn * (P::S2 * 2048));
  };
  auto load_bias = [&](int j) {
    const float* b = b1p + j * P::HC;
#pragma unroll
    for (int g = 0; g < 4; ++g) mw_bload<PERSIST>(bq[g], bvoff, b + 8 * g);
    if constexpr (FOLD) {
      const float* u = u1p + j * P::HC;
#pragma unroll
      for (int g = 0; g < 4; ++g) mw_bload<PERSIST>(uq[g], bvoff, u + 8 * g);
    }
  };

  // prologue: K tiles 0 and 1 of chunk 0, their W1 fragments, the bias quads — drained ONCE, so that the counted waits of
  // chunk 0's first two tiles (which assume the tail of a previous chunk ahead of them) are trivially met
  MW_OPAQUE(arow); MW_OPAQUE(w1w); MW_OPAQUE(w2w); MW_OPAQUE(b1p); MW_OPAQUE(lds0);
  issue_x(0, 0);
  issue_x(1, 1);
  load_w1_tile(c0, 0, 0);
  load_w1_tile(c0, 1, 1);
  load_bias(c0);
  if constexpr (FOLD) {
    const float* sp = q.ln_stats + 2 * (size_t)m0;
    MW_OPAQUE(sp);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned so = (unsigned)(32 * i + r) * 8u;
      if constexpr (PERSIST) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "=&v"(st[i]) : "v"(so), "s"(sp) : "memory");
      else asm volatile("global_load_dwordx2 %0, %1, %2" : "=&v"(st[i]) : "v"(so), "s"(sp) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]) :: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  for (int j = c0; j < c1; ++j) {
    const int jn1 = j + 1 < c1 ? j + 1 : j;
    MW_OPAQUE(arow); MW_OPAQUE(w1w); MW_OPAQUE(w2w); MW_OPAQUE(b1p); MW_OPAQUE(lds0);                  // the last chunk re-requests its own data: the op counts stay uniform
    f32x16 g1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) g1[i][e] = 0.0f;

    // ================= GEMM 1 of chunk j.  VMEM operations of tile kt, in issue order: 8 DMAs (LayerNorm-row tile kt + 2,
    // into the ring slot tile kt - 1 just left), then — kt + 2 < NKT only — 2 W1 loads behind each k16 step (tile kt + 2,
    // into the registers that step just consumed).  The W1 fragments of the NEXT chunk's tiles 0 / 1 are requested during
    // GEMM 2 (steps 4 / 5), when the GEMM-1 accumulators are dead.
    //   wait at the top of tile kt = operations issued after (DMA, W1) of tile kt:
    //     kt = 0: W2 step 7 (2 TN2) + W1 of tile 1 (8) + bias (4) = 20 / 16     kt = 1: bias (4) + tile 0's 16 = 20
    //     kt = 2 .. NKT - 2: the 16 of tile kt - 1                         kt = NKT - 1: the 8 DMAs of tile NKT - 2
    auto ktile = [&](auto kt_c, auto buf_c, int slot) {
      constexpr int KT_ = decltype(kt_c)::value, BUF = decltype(buf_c)::value;
      if (!(ABL & 2)) {
#ifdef MW_EXPERIMENT_TILE_WAIT
        MW_WAIT8(MW_EXPERIMENT_TILE_WAIT, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
#else
        // (FOLD: the u quads ride with the bias quads — every count that holds "bias (4)" holds 8)
        if constexpr (KT_ == 0 && TN2 == 2 && FOLD) MW_WAIT8(20, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
        else if constexpr (KT_ == 0 && TN2 == 2) MW_WAIT8(16, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);   // a W2 step is 2 TN2 loads
        else if constexpr (KT_ <= 1 && FOLD) MW_WAIT8(24, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
        else if constexpr (KT_ <= 1) MW_WAIT8(20, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
        else if constexpr (KT_ == NKT - 1) MW_WAIT8(8, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
        else MW_WAIT8(16, wh[BUF][0], wl[BUF][0], wh[BUF][1], wl[BUF][1], wh[BUF][2], wl[BUF][2], wh[BUF][3], wl[BUF][3]);
#endif
      }
      if (!(ABL & 4)) MW_BARRIER();
      {
        int s2 = slot + 2;
        if (s2 >= P::NSLOT) s2 -= P::NSLOT;
        issue_x(KT_ + 2 < NKT ? KT_ + 2 : KT_ + 2 - NKT, s2);   // the rows are re-streamed per chunk: tile kt + 2 wraps round
      }
      const unsigned char* xs = smem_raw + slot * P::XSLOT;
      const unsigned char* wnext = w1w + (size_t)(4 * j) * (P::S1 * 2048) + (KT_ + 2) * 8192;
      // The fragments of k16 step ks + 1 are read WHILE the 12 MFMAs of step ks run (two register sets, one ds_read_b128
      // behind each of the first eight MFMAs): left to itself hipcc reads the lo halves just in time — "ds_read, wait
      // lgkmcnt(0), MFMA" four times per step, an exposed LDS round trip each.  Only a tile's first step waits for its reads.
      h8 xh[2][4], xl[2][4];
      auto read_x = [&](int ks, h8 (&fh)[4], h8 (&fl)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          fh[i] = *reinterpret_cast<const h8*>(xs + i * 8192 + (xb ^ (ks * 64)));
          fl[i] = *reinterpret_cast<const h8*>(xs + i * 8192 + (xb ^ (ks * 64 + 16)));
        }
      };
      read_x(0, xh[0], xl[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cb = ks & 1;
        if (ks < 3) read_x(ks + 1, xh[cb ^ 1], xl[cb ^ 1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[BUF][ks], xh[cb][i], g1[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[BUF][ks], xl[cb][i], g1[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) g1[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[BUF][ks], xh[cb][i], g1[i], 0, 0, 0);
        if (ks < 3) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (KT_ + 2 < NKT) {
          if (!(ABL & 8)) mw_wload<PERSIST>(wh[BUF][ks], wl[BUF][ks], lane16, wnext + ks * 2048);
        }
      }
    };
    {
      int slot = 0;                                                // NKT tiles per chunk and NSLOT = 3: the slot of tile 0 moves on by NKT % 3 per chunk
      slot = ((j - c0) * (NKT % P::NSLOT)) % P::NSLOT;
      auto adv = [&]() { slot = slot + 1 == P::NSLOT ? 0 : slot + 1; };
      ktile(std::integral_constant<int, 0>{}, I0{}, slot); adv();
      ktile(std::integral_constant<int, 1>{}, I1{}, slot); adv();
      ktile(std::integral_constant<int, 2>{}, I0{}, slot); adv();
      ktile(std::integral_constant<int, 3>{}, I1{}, slot); adv();
      if constexpr (NKT == 8) {
        ktile(std::integral_constant<int, 4>{}, I0{}, slot); adv();
        ktile(std::integral_constant<int, 5>{}, I1{}, slot); adv();
        ktile(std::integral_constant<int, 6>{}, I0{}, slot); adv();
        ktile(std::integral_constant<int, 7>{}, I1{}, slot); adv();
      }
    }

    // W2 fragments of the chunk's first two k16 steps: they land during the GELU
    load_w2_step(8 * j, 0);
    load_w2_step(8 * j + 1, 1);
    // the bias quads were requested a whole GEMM 1 ago (complete since the wait of tile 2): tie them in
#ifdef MW_EXPERIMENT_ALL0
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) :: "memory");
#else
    asm volatile("s_waitcnt vmcnt(48)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]) :: "memory");
#endif
    if constexpr (FOLD) asm volatile("" : "+v"(uq[0]), "+v"(uq[1]), "+v"(uq[2]), "+v"(uq[3]));      // requested with the bias quads: complete with them

    // ================= bias + GELU + split of the wave's 128 x 32 piece -> hidden-chunk buffer
    // (every wave passed a K-tile barrier after its GEMM 2 of chunk j - 1: the buffer is free)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 hi[4], lo[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (ABL & 1) {
          hi[g] = u32x2{__builtin_bit_cast(unsigned, g1[i][4 * g]), __builtin_bit_cast(unsigned, g1[i][4 * g + 1])};
          lo[g] = u32x2{__builtin_bit_cast(unsigned, g1[i][4 * g + 2]), __builtin_bit_cast(unsigned, g1[i][4 * g + 3])};
        } else {
          // this phase has no MFMAs to hide behind: the vector form (hipcc packs it: v_pk_fma_f32 / v_pk_mul_f32) is half
          // the instructions of mc_epi_group's scalar one and the same arithmetic per element
          const f32x4 v = {g1[i][4 * g], g1[i][4 * g + 1], g1[i][4 * g + 2], g1[i][4 * g + 3]};
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (FOLD) o[e] = wd_gelu(fmaf(st[i][1], fmaf(-st[i][0], uq[g][e], v[e] * q.unscale1), bq[g][e]));   // epi_lds_tile_csplit's fold
            else o[e] = wd_gelu(fmaf(v[e], q.unscale1, bq[g][e]));
          }
          if (q.hid_scale != 1.0f) o = o * q.hid_scale;            // power of two (exact); wave-uniform branch
          split4(o, hi[g], lo[g]);
        }
      }
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MW_BARRIER();                                                 // the hidden chunk is complete

    // ================= GEMM 2 of chunk j: out += W2[:, chunk] . hidden chunk.  VMEM operations behind the MFMAs of step ks:
    // W2 step ks + 2 (8; ks < 6), at ks = 4 / 5 then the W1 fragments of the next chunk's K tile 0 / 1 (8), at ks = 6 the
    // next chunk's bias quads (4).  wait before step ks = operations issued after W2 step ks:
    //   ks = 0 .. 4: the 8 of step ks + 1      ks = 5: W2 6 (8) + W1 tile 0 (8) = 16      ks = 6: W1 tile 0 (8) + W2 7 (8) + W1 tile 1 (8) = 24
    //   ks = 7: W1 tile 1 (8) + bias (4) = 12
    h8 hh[2][4], hl[2][4];
    auto read_h = [&](int ks, h8 (&fh)[4], h8 (&fl)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fh[i] = *reinterpret_cast<const h8*>(hbuf + i * 16384 + (hb ^ (ks * 64)));
        fl[i] = *reinterpret_cast<const h8*>(hbuf + i * 16384 + (hb ^ (ks * 64 + 16)));
      }
    };
    auto g2step = [&](auto ks_c, auto buf_c) {
      constexpr int KS = decltype(ks_c)::value, BUF = decltype(buf_c)::value;
      static_assert(TN2 == 4 || TN2 == 2, "wait operand lists");
      if constexpr (TN2 == 4) {
#define MW_W2WAIT(N) MW_WAIT8(N, vh[BUF][0], vl[BUF][0], vh[BUF][1], vl[BUF][1], vh[BUF][2], vl[BUF][2], vh[BUF][3], vl[BUF][3])
        if constexpr (KS <= 4) MW_W2WAIT(8); else if constexpr (KS == 5) MW_W2WAIT(16); else if constexpr (KS == 6) MW_W2WAIT(24); else if constexpr (FOLD) MW_W2WAIT(16); else MW_W2WAIT(12);
#undef MW_W2WAIT
      } else {
#ifdef MW_EXPERIMENT_ALL0
#define MW_W2WAIT(N) asm volatile("s_waitcnt vmcnt(0)" : "+v"(vh[BUF][0]), "+v"(vl[BUF][0]), "+v"(vh[BUF][1]), "+v"(vl[BUF][1]) :: "memory")
#else
#define MW_W2WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(vh[BUF][0]), "+v"(vl[BUF][0]), "+v"(vh[BUF][1]), "+v"(vl[BUF][1]) :: "memory")
#endif
        // TN2 = 2: a W2 step is 4 loads
        if constexpr (KS <= 4) MW_W2WAIT(4); else if constexpr (KS == 5) MW_W2WAIT(12); else if constexpr (KS == 6) MW_W2WAIT(20); else if constexpr (FOLD) MW_W2WAIT(16); else MW_W2WAIT(12);
#undef MW_W2WAIT
      }
      // hidden-chunk fragments: step KS + 1 is read behind the first eight MFMAs of step KS (see GEMM 1)
      if constexpr (KS == 0) {
        read_h(0, hh[0], hl[0]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (KS < 7) read_h(KS + 1, hh[BUF ^ 1], hl[BUF ^ 1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[BUF][jn], hh[BUF][i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[BUF][jn], hl[BUF][i], out[i][jn], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < TN2; ++jn) out[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[BUF][jn], hh[BUF][i], out[i][jn], 0, 0, 0);
      if constexpr (KS < 7) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12 * TN2 - 8, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KS < 6) load_w2_step(8 * j + KS + 2, BUF);
      if constexpr (KS == 4) load_w1_tile(jn1, 0, 0);
      if constexpr (KS == 5) load_w1_tile(jn1, 1, 1);
      if constexpr (KS == 6) load_bias(jn1);
    };
    g2step(std::integral_constant<int, 0>{}, I0{});
    g2step(std::integral_constant<int, 1>{}, I1{});
    g2step(std::integral_constant<int, 2>{}, I0{});
    g2step(std::integral_constant<int, 3>{}, I1{});
    g2step(std::integral_constant<int, 4>{}, I0{});
    g2step(std::integral_constant<int, 5>{}, I1{});
    g2step(std::integral_constant<int, 6>{}, I0{});
    g2step(std::integral_constant<int, 7>{}, I1{});
  }
  // The last chunk re-requested its own W1 fragments and bias quads (uniform op counts) and nobody consumes them: tie every
  // asm-loaded register into / behind the final drain, or the compiler hands a register to the epilogue's address
  // arithmetic while a load into it is still in flight (found by scripts/check_asm_loads.py on the first persistent build:
  // a memory fault, not a wrong digit)
  MW_WAIT8(0, wh[0][0], wl[0][0], wh[0][1], wl[0][1], wh[0][2], wl[0][2], wh[0][3], wl[0][3]);
  asm volatile("" : "+v"(wh[1][0]), "+v"(wl[1][0]), "+v"(wh[1][1]), "+v"(wl[1][1]), "+v"(wh[1][2]), "+v"(wl[1][2]), "+v"(wh[1][3]), "+v"(wl[1][3]));
  asm volatile("" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
  if constexpr (FOLD) asm volatile("" : "+v"(uq[0]), "+v"(uq[1]), "+v"(uq[2]), "+v"(uq[3]));
#pragma unroll
  for (int jn = 0; jn < TN2; ++jn) asm volatile("" : "+v"(vh[0][jn]), "+v"(vl[0][jn]), "+v"(vh[1][jn]), "+v"(vl[1][jn]));
  MW_BARRIER();                                                   // ring and hidden buffer are free: epilogue patches

  if (PERSIST && is_head) {
    // park the raw accumulators for the next workgroup of the chain: coalesced stores, drained, then ONE agent-scope
    // release and the flag (MI355X_MICROARCH.md, inter-workgroup visibility)
    unsigned char* sb = reinterpret_cast<unsigned char*>(q.park) + (size_t)blockIdx.x * (4 * TN2 * 4 * 4096);
    const unsigned toff = (unsigned)t * 16u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < TN2; ++jn)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          MW_OPAQUE(sb);
          *reinterpret_cast<f32x4*>(sb + toff) = f32x4{out[i][jn][4 * g], out[i][jn][4 * g + 1], out[i][jn][4 * g + 2], out[i][jn][4 * g + 3]};
          sb += 4096;
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(q.flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    continue;
  }
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
  }  // pieces
}

// one workgroup per CU, the same number on every XCD — of the CURRENT device
int mw_workgroups() {
  static int n[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  int& slot = n[dev & 63];
  if (!slot) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    slot = prop.multiProcessorCount / 8 * 8;
  }
  return slot;
}

template <int C, int ABL = 0, bool FOLD = false>
int launch_mlp_wide(MwArgs q, hipStream_t st, float* ws, long long ws_floats) {
  using P = MW<C>;
  const int nblk = q.m / P::BM;
  q.nblk = nblk;
  const int wgs = mw_workgroups();
  // persistent form: every XCD chain needs at least one whole row block per workgroup (a shorter range would chain serially)
  // The persistent form is built for C = 256 only (round 5): at C = 512 hipcc gives it 44 bytes of scratch (10 VGPR + 359
  // SGPR spills) — scratch reloads are VMEM operations the counted vmcnt waits of the chunk loop do not know about, and
  // the build now refuses scratch in every kernel that counts (build.py: NO_SCRATCH).  The 512-channel form is off by
  // default (2-3 % slower than the two launches either way, profiles/r04_mlp_wide.txt) and keeps the tile form.
  constexpr bool PERSIST_BUILT = C != 512;
  const bool persist = PERSIST_BUILT && ws != nullptr && wgs > 0 && nblk / 8 >= wgs / 8 && nblk > wgs &&
                       ws_floats >= 1024 + (long long)wgs * (4 * P::TN2 * 4 * 256 * 4);
  if constexpr (PERSIST_BUILT) if (persist) {
    q.flag = reinterpret_cast<unsigned*>(ws);                     // first 4 KB: flags (zero between launches)
    q.park = ws + 1024;
    static WdAttrOnce attr;
    if (wd_set_max_lds(attr, reinterpret_cast<const void*>(fused_mlp_wide_kernel<C, ABL, true, FOLD>), P::LDS) != WD_OK) return WD_ERR_LAUNCH;
    WD_LAUNCH_GEMM((fused_mlp_wide_kernel<C, ABL, true, FOLD>), dim3((unsigned)wgs), dim3(256), P::LDS, st, q);
    return wd_launch_status();
  }
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(fused_mlp_wide_kernel<C, ABL, false, FOLD>), P::LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM((fused_mlp_wide_kernel<C, ABL, false, FOLD>), dim3((unsigned)nblk), dim3(256), P::LDS, st, q);
  return wd_launch_status();
}

}  // namespace

// Weights: FRAGMENT-MAJOR copies of the wd_split_weights buffers (lib.mlp_wide_pack): for a [n][k] weight matrix,
// [n / 32][k / 16][2 (hi, lo)][64 lanes: (k half) * 32 + (row in block)][16 B].
extern "C" int wd_mlp_fused_wide(const void* a_split, int64_t rows, int32_t c, int32_t hidden, const void* w1_frag,
                                 float w1_unscale, const float* b1, const void* w2_frag, float w2_unscale, const float* b2,
                                 float* x, float hid_scale, uint32_t* range_flag, void* workspace, int64_t workspace_bytes,
                                 void* stream) {
  if (workspace && (!wd_aligned16(workspace) || workspace_bytes < 0)) return WD_ERR_BAD_ARG;
  if (!a_split || !w1_frag || !w2_frag || !b1 || !b2 || !x) return WD_ERR_BAD_ARG;
  if ((c != 256 && c != 512) || hidden != 4 * c) return WD_ERR_UNSUPPORTED;
  if (rows <= 0 || rows % 128 || rows / 128 > 0x7fffffffLL) return WD_ERR_UNSUPPORTED;
  if (!wd_aligned16(a_split) || !wd_aligned16(w1_frag) || !wd_aligned16(w2_frag) || !wd_aligned16(b1) || !wd_aligned16(b2) ||
      !wd_aligned16(x))
    return WD_ERR_BAD_ARG;
  if (!(w1_unscale > 0.f) || !(w2_unscale > 0.f) || !(hid_scale > 0.f)) return WD_ERR_BAD_ARG;
  MwArgs q{static_cast<const unsigned char*>(a_split), static_cast<const unsigned char*>(w1_frag),
           static_cast<const unsigned char*>(w2_frag), b1, b2, x, range_flag, (int)rows, w1_unscale, w2_unscale, hid_scale,
           nullptr, nullptr, 0, nullptr, nullptr};
  float* ws = static_cast<float*>(workspace);
  const long long wsf = workspace ? workspace_bytes / 4 : 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
#ifdef WD_DEBUG_ABLATIONS
  if (const char* e = getenv("WD_MLPW_ABL")) {
    if (c != 512) return WD_ERR_UNSUPPORTED;
    switch (atoi(e)) {
      case 1: return launch_mlp_wide<512, 1>(q, st, ws, wsf);
      case 2: return launch_mlp_wide<512, 2>(q, st, ws, wsf);
      case 4: return launch_mlp_wide<512, 4>(q, st, ws, wsf);
      case 6: return launch_mlp_wide<512, 6>(q, st, ws, wsf);
      case 7: return launch_mlp_wide<512, 7>(q, st, ws, wsf);
      case 8: return launch_mlp_wide<512, 8>(q, st, ws, wsf);
      case 14: return launch_mlp_wide<512, 14>(q, st, ws, wsf);
      case 15: return launch_mlp_wide<512, 15>(q, st, ws, wsf);
      default: break;
    }
  }
#endif
  return c == 512 ? launch_mlp_wide<512>(q, st, ws, wsf) : launch_mlp_wide<256>(q, st, ws, wsf);
}

// wd_mlp_fused_wide_ln (ABI 14): the same block MLP with the block's LayerNorm FOLDED into pwconv1 (WdConvGemm.ln_stats /
// ln_u semantics): d_split = the raw depthwise output of wd_dwconv7_stats, w1g_frag = fragment-major split of W1 gamma,
// v = W1 beta + b1, u = (W1 gamma) 1, ln_stats = wd_ln_stats_finalize's per-row (mean, rstd).  256 channels.
extern "C" int wd_mlp_fused_wide_ln(const void* d_split, int64_t rows, int32_t c, int32_t hidden, const void* w1g_frag,
                                    float w1_unscale, const float* v, const float* u, const float* ln_stats, const void* w2_frag,
                                    float w2_unscale, const float* b2, float* x, float hid_scale, uint32_t* range_flag,
                                    void* workspace, int64_t workspace_bytes, void* stream) {
  if (workspace && (!wd_aligned16(workspace) || workspace_bytes < 0)) return WD_ERR_BAD_ARG;
  if (!d_split || !w1g_frag || !w2_frag || !v || !u || !ln_stats || !b2 || !x) return WD_ERR_BAD_ARG;
  if (c != 256 || hidden != 4 * c) return WD_ERR_UNSUPPORTED;
  if (rows <= 0 || rows % 128 || rows / 128 > 0x7fffffffLL) return WD_ERR_UNSUPPORTED;
  if (!wd_aligned16(d_split) || !wd_aligned16(w1g_frag) || !wd_aligned16(w2_frag) || !wd_aligned16(v) || !wd_aligned16(u) ||
      !wd_aligned16(b2) || !wd_aligned16(x) || (reinterpret_cast<uintptr_t>(ln_stats) & 7u))
    return WD_ERR_BAD_ARG;
  if (!(w1_unscale > 0.f) || !(w2_unscale > 0.f) || !(hid_scale > 0.f)) return WD_ERR_BAD_ARG;
  MwArgs q{static_cast<const unsigned char*>(d_split), static_cast<const unsigned char*>(w1g_frag),
           static_cast<const unsigned char*>(w2_frag), v, b2, x, range_flag, (int)rows, w1_unscale, w2_unscale, hid_scale,
           nullptr, nullptr, 0, ln_stats, u};
  return launch_mlp_wide<256, 0, true>(q, static_cast<hipStream_t>(stream), static_cast<float*>(workspace), workspace ? workspace_bytes / 4 : 0);
}
