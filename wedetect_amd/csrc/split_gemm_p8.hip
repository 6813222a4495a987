// split_gemm_p8.hip — fp16x3 GEMM for the big plain layers with pre-split operands (the ConvNeXt
// pointwise MLPs, mm_backbone.py:115-124): 256 x 256 output tile, K tiles of 32, eight waves in two
// row groups running half a phase apart, four phases per K tile.
//
// Why another structure.  The 128 x 128 direct-to-LDS kernel (split_gemm_pre.hip) pays one
// workgroup barrier and one full vmcnt(0) drain per 12 MFMAs and reads 8 ds_read_b128 per 12
// MFMAs; its matrix pipe is busy a third of the time.  Here a wave owns 128 x 64 outputs (eight
// 32 x 32 accumulators = 128 registers), so a K tile of 32 costs 24 ds_read_b128 per 48 MFMAs,
// and global -> LDS traffic per MFMA halves.  The K loop never drains the DMA queue:
//
//   * LDS (144 KB): two K-tile buffers of 48 KB, each cut into three 16 KB regions (128 rows x 128 B)
//       A0 / A1 = activation rows {0-63, 128-191} / {64-127, 192-255} of the tile (the first / second
//                 64 rows of each row group),  W1 = weight rows {64 wn + 32-63},
//     plus a ring of THREE 16 KB slots for W0 = weight rows {64 wn + 0-31} (K tile t lives in slot t % 3).
//   * a K tile is consumed in four phases, one 64 x 32 output quadrant of the wave per phase
//     (12 MFMAs = 2 row blocks x 2 k16 steps x {lo.hi, hi.lo, hi.hi}):
//       phase 1  read A0 (8 x ds_read_b128) + W0 (4)   quadrant (rows 0-63,  cols 0-31)
//       phase 2  read W1 (4)                           quadrant (rows 0-63,  cols 32-63)
//       phase 3  read A1 (8)                           quadrant (rows 64-127, cols 32-63)
//       phase 4  (W0 is still in registers)            quadrant (rows 64-127, cols 0-31)
//     One activation and two weight fragment sets (64 registers) next to the 128 accumulator registers (198 in
//     all; the persistent form 234, no scratch).  W0 keeps its three-slot ring from the form that re-read it in
//     phase 4 (same speed within noise, 4 more reads per K tile): a request two K tiles ahead never waits for a slot.
//     The per-accumulator MFMA sequence (k ascending, lo.hi -> hi.lo -> hi.hi inside a k16 step) is the
//     one of every other fp16x3 kernel: results are bit-identical to theirs.
//   * every phase also re-fills one region that went dead two phases earlier (tile t: phase 1: A1 of tile
//     t+1, 2: W0 of t+2, 3: A0 of t+2, 4: W1 of t+2): 16 one-KB global_load_lds_dwordx4 per region, two per
//     wave, issued from inline asm so that hipcc does not serialise them against the ds_reads.  A region is
//     read six or seven phases after it was requested; each wave retires its own requests with a COUNTED
//     s_waitcnt vmcnt(10) (the five newest regions stay in flight across the barriers) and never waits
//     vmcnt(0) in steady state.
//   * phase = [ds_reads, DMA issue, vmcnt] s_barrier [12 MFMAs] s_barrier.  The two row groups
//     (waves 0-3 / 4-7; waves w and w + 4 share a SIMD) are offset by one barrier, so a SIMD's
//     matrix pipe is fed by one wave while its partner reads and requests.  A region is re-filled
//     two phases after its last read: the other group's reads of it (issued up to one barrier later,
//     completed before its MFMAs) are over before the DMA can be issued.
//   * LDS rows are unpadded 128 B with the 16-byte slot XOR-swizzled by (row / 2) & 7 — applied on the
//     GLOBAL side of the DMA and on the ds_read address (LDS-DMA writes lane-linear); out-of-range
//     rows are clamped to the last valid row (their outputs are never stored), so no zero page.
//
// Tried and dropped (round 2, profiles/r02_p8_sched.txt): the same four phases software-pipelined INSIDE the wave —
// fragment registers used as two half sets (k16 step 0 / 1), the ds_reads of one step in flight behind the six MFMAs
// of the other (inline-asm reads, hand-counted lgkmcnt), one barrier per phase placed between the 9th and 10th MFMA,
// all eight waves in step (or the second row group's barrier three MFMAs later).  Bit-identical, no spills (208
// registers), and 7-8 % SLOWER (stage-3 pwconv2 321-333 us against 298): with both waves of a SIMD in the same
// position of the same stream, their read / DMA-issue bursts (an LDS-DMA instruction costs 60-185 issue cycles)
// coincide instead of hiding behind the partner's MFMAs.  The staggered two-barrier form below keeps one wave of
// every SIMD in its MFMA run while the other issues memory work, which is what the matrix pipe needs.
#include <stdlib.h>
#include "split_gemm_impl.h"

long long wd_p8_workspace_floats();

namespace {

constexpr int P8_ROWB = 128, P8_REGION = 128 * P8_ROWB, P8_TILE = 3 * P8_REGION, P8_LDS = 2 * P8_TILE + 3 * P8_REGION;
constexpr int P8_A0 = 0, P8_W1 = P8_REGION, P8_A1 = 2 * P8_REGION, P8_W0RING = 2 * P8_TILE;   // W0 of K tile t: P8_W0RING + (t % 3) * P8_REGION

// one 1 KB LDS-DMA: 64 lanes x 16 B from base + voff[lane] to LDS [lds_addr, +1024)
__device__ __forceinline__ void p8_dma(unsigned lds_addr, unsigned voff, const unsigned char* base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
               :: "s"(lds_addr), "v"(voff), "s"(base) : "memory", "m0");
}

#define P8_KBARRIER() do { if (!(ABL & 8)) P8_BARRIER(); } while (0)
#define P8_BARRIER()                          \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_s_barrier();             \
    asm volatile("" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

// ABL: timing-only ablations for on-device diagnosis (WRONG results by construction; compiled only with
// -DWD_DEBUG_ABLATIONS, reachable through cfg 640 + ABL): 1 = no DMA in the K loop, 2 = no ds_reads in the K loop,
// 4 = no epilogue, 8 = no barriers in the K loop, 16 = both wave groups in step (no stagger).
//
// PERSIST (cfg 65): one workgroup per CU walks a contiguous range of (tile, K tile) work units instead of
// owning whole output tiles, so 400 tiles on 256 CUs cost 1.5625 tile times, not 2 (and the epilogues of
// different CUs stop happening in the same microseconds).  Round 4: the units are dealt to GANGS of ngrp workgroups,
// not to single workgroups.  The round-2/3 form gave every workgroup its own contiguous run of tiles: the column tiles
// of one row panel, which the tile form runs on neighbouring CUs of an XCD at the same time (the activation panel is
// fetched once and hits in L2 for the others), were then walked by ONE workgroup one after the other while its 31
// neighbours walked 31 other panels — 32 panels live per 4 MB L2 instead of 4, and the PMC pass says what that costs
// (profiles/r04_persist_pmc.txt, stage-3 pwconv1 / pwconv2: FETCH_SIZE 1.35 / 1.34 GB per launch against 0.33 / 0.58 GB
// for the tile form, TCC misses 2.4 x) — hidden by the Infinity Cache in a one-kernel loop, 12 % slower inside the step.
// A gang = ngrp workgroups with consecutive slots on one XCD; member j owns column tile j of every gang tile (= the ngrp
// column tiles of a row panel, exactly the tiles the tile form runs side by side) and all members share the same unit
// range, so the gang moves through K in step and the panel is fetched once.  The gang-tile list is cut into 8 chunks,
// one per XCD (workgroup b runs on XCD b % 8), each chunk's units are dealt evenly to the XCD's gangs.  A gang's
// range therefore starts and / or ends inside a tile.  The head of a tile [0, e) is computed FIRST, its raw accumulators are parked in the workspace and a
// flag is published (agent-scope release); whole tiles follow; LAST comes the tail [o, nk) of the tile the
// previous workgroup of the chain began: wait for that flag (set long before), acquire, reload the
// accumulators and continue the K loop where it stopped — the same MFMA chain, only carried by two CUs, so the
// result is bit-identical to the one-workgroup-per-tile kernels.  A workgroup waits only on the same member of the
// previous gang of its XCD chain (block b - 8 ngrp, dispatched earlier), and only at the very end of its own work; the
// first gang of a chain never waits.  Needs at least one whole gang tile per gang on every XCD.
constexpr long long P8_PARK_BYTES = 8ll * 32768;       // per workgroup: 8 accumulators x 512 lanes x 64 B
struct P8Persist {
  float* acc;            // [workgroup][512 lanes][128 floats]: parked accumulators of the tile a workgroup shares with its successor
  unsigned* flag;        // [workgroup]: 1 = parked; reset to 0 by the consumer
  const float* zero;     // 512 bytes of zeros
  int ntiles;            // gang tiles: row panels x column groups
};

// ---- retrieval scoring on this kernel (round 5; wd_retrieval_max_split, retrieval_metric.py:367-377) ----------------------
// The GEMM is run with the operand ROLES SWAPPED: the text bank is the "activation" operand (p.a, p.m = classes of this
// launch) and the region rows of all images, back to back, are the "weight" operand (wsp, p.n = region rows).  In the
// accumulator layout of v_mfma_f32_32x32x16 the activation row is the LANE and the weight row the REGISTER, so a lane then
// holds, for ONE class, 32 of the wave's 64 region rows: the max over an image's regions is 31 in-register v_max plus one
// lane exchange (the form on the 256 x 128 ping-pong kernel — classes in registers, rows in lanes — paid a five-step
// segmented shuffle reduction PER VALUE: ~28 VALU instructions per accumulator element).  sigmoid is monotone, so it is
// applied once to the maximum of the affine logits instead of to every (row, class): max_r s(x_r) = s(max_r x_r).
// Tile order: region-row tiles in groups of eight, four bank blocks x eight row tiles live per XCD at a time (the gang
// order of the MLP launches): per 32 tiles an L2 fetches 12 operand panels instead of 33, the bank streams from HBM once per
// row-tile group (five times for 32 images x 300 regions), the region rows (29 MB) stay in the Infinity Cache.
constexpr int P8VAR_RETR = 4096;
constexpr int P8VAR_DIRECT = 8192;      // with SVAR_CSPLIT: the direct (no LDS transpose) hi/lo epilogue; GELU / no activation only
struct P8Retr {
  const float* scale;    // [region rows]: logit scale of a row, applied as exp(scale)
  const float* bias;     // [region rows]
  const int* count;      // [images]: valid rows of an image (the first count[i] of its rows_per_img)
  float* out;            // [images][ldo] zero-filled: max over an image's valid rows of sigmoid(<e, t> exp(scale) + bias)
  int rows_per_img, ldo;
};

template <int TM, int TN>
__device__ __forceinline__ void p8_retr_epilogue(const WdConvGemm& p, const P8Retr& rt, float unscale, int mw, int nw, int lane,
                                                 const f32x16 (&acc)[TM][TN]) {
  const int nrows = p.n;
  if (nw >= nrows || mw >= p.m) return;                            // wave-uniform
  const int half = lane >> 5;
  if (p.range_flag) {                                              // an fp16 half that overflowed to inf: inf / NaN accumulators
    bool bad = false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bad |= wd_any_nonfinite4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
    if (bad) *p.range_flag = 1u;
  }
  // this lane's rows: row(j, r) = nw + 32 j + 8 (r >> 2) + 4 half + (r & 3).  Their exp(scale) * unscale and bias are
  // loaded per (image, j) — 32 live registers next to the 128 accumulators; holding all 64 spilled (the build refuses scratch)
  const float ninf = -__builtin_inff();
  const int rpi = rt.rows_per_img;
  const int last = (nw + 63 < nrows ? nw + 63 : nrows - 1);
  const int img_last = last / rpi;
  for (int img = nw / rpi; img <= img_last; ++img) {               // usually one image, two where a wave's 64 rows straddle
    int cnt = rt.count[img];
    cnt = cnt < rpi ? cnt : rpi;
    if (cnt <= 0) continue;
    const int d = nw - img * rpi + 4 * half;                       // row(j, r) - first row of the image = d + const(j, r)
    float v[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) v[i] = ninf;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float e_[16], b_[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int base = nw + 32 * j + 8 * g + 4 * half;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
        if (base + 3 < nrows) {
          s4 = *reinterpret_cast<const f32x4*>(rt.scale + base);
          b4 = *reinterpret_cast<const f32x4*>(rt.bias + base);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (base + q < nrows) { s4[q] = rt.scale[base + q]; b4[q] = rt.bias[base + q]; }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool in = (unsigned)(d + 32 * j + 8 * g + q) < (unsigned)cnt;
          e_[4 * g + q] = in ? expf(s4[q]) * unscale : 0.0f;       // rows of other images / beyond the count: 0 * acc - inf
          b_[4 * g + q] = in ? b4[q] : ninf;
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[i] = fmaxf(v[i], fmaf(acc[i][j][r], e_[r], b_[r]));
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float o = __shfl_xor(v[i], 32, 64);                    // the other four rows of every group of eight
      const float m = fmaxf(v[i], o);
      const int cls = mw + 32 * i + (lane & 31);
      if ((i & 1) == half && cls < p.m && m > ninf)                // sigmoid > 0 and out starts at 0: uint order = float order
        atomicMax(reinterpret_cast<unsigned int*>(rt.out + (size_t)img * rt.ldo + cls), __float_as_uint(wd_sigmoid(m)));
    }
  }
}

template <int VAR, int ABL = 0, bool PERSIST = false>
__global__ void __launch_bounds__(512, 2)
split_gemm_p8_kernel(const WdConvGemm p, const unsigned char* __restrict__ wsp, int k16, float unscale, int nbn,
                     int vec_c, int vec_res, int vec_bias, int ngrp, int nbm, const P8Persist ps, const P8Retr rt, int stagger,
                     int arows, int wrows) {
  constexpr int TM = 4, TN = 2, BM = 256, BN = 256, ROWB = P8_ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int group = wave >> 2, wn = wave & 3;
  const int nk = p.k >> 5;
  const int gsz = ngrp * nbm;

  // Experiment ($WD_P8_STAGGER, tile form, -DWD_DEBUG_ABLATIONS builds; measured: no gain, profiles/r05_p8_epilogue_ablation.txt): every tile costs the same, so all CUs reach their epilogues — 256 KB of stores each —
  // in the same microseconds of every round.  Delaying the first tile of every other CU by `stagger` x ~4 us takes the two
  // halves of the chip out of phase for the whole launch.
  if (!PERSIST && stagger > 0 && blockIdx.x < 256u && ((blockIdx.x >> 3) & 1))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  // ---- work assignment
  int seg_first_tile, seg_o, seg_last_tile, seg_e;   // units [first_tile * nk + o, last_tile * nk + e)
  if (PERSIST) {
    // gang = slot / ngrp, member = slot % ngrp; gang tile gt <-> tiles gt * ngrp + member of the tile form's list
    const int b = blockIdx.x, xcd = b & 7, slot = b >> 3, per = (gridDim.x >> 3) / ngrp, gang = slot / ngrp;
    const long long t0 = (long long)ps.ntiles * xcd / 8, t1 = (long long)ps.ntiles * (xcd + 1) / 8;
    const long long units = (t1 - t0) * nk;
    const long long u0 = t0 * nk + units * gang / per, u1 = t0 * nk + units * (gang + 1) / per;
    seg_first_tile = (int)(u0 / nk); seg_o = (int)(u0 - (long long)seg_first_tile * nk);
    seg_last_tile = (int)(u1 / nk); seg_e = (int)(u1 - (long long)seg_last_tile * nk);
  } else {
    int tile = blockIdx.x;
    const int nwg = gridDim.x, xcd = tile & 7, idx = tile >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    seg_first_tile = tile; seg_o = 0; seg_last_tile = tile + 1; seg_e = 0;
  }
  // segments in execution order: [head of the last tile] [whole tiles] [tail of the first tile]
  const int has_head = seg_e > 0, has_tail = seg_o > 0;
  const int full0 = seg_first_tile + has_tail, nfull = seg_last_tile - full0;
  const int nseg = has_head + nfull + has_tail;

  const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.a);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const unsigned dma_dst = lds0 + wave * 1024;                   // + region offset + j * 8192: instruction j of wave w fills slots [(w + 8 j) * 8, + 8)
  // ---- fragment addresses (byte offsets inside a region): slot * 128 + ((logical ^ f(slot)) << 4), logical =
  // 4 ks + 2 (hi / lo) + hsel.  With g = hsel ^ f(slot) the four (ks, hi / lo) variants of a lane are
  // base ^ {0, 32, 64, 96}, base = slot * 128 + (g << 4): ONE register per operand (the second activation row
  // block is + 4096), the variants cost one v_xor each.
  const int hsel = lane >> 5;
  int abase_off, wbase_off;
  {
    const int slot = group * 64 + (lane & 31);
    abase_off = slot * ROWB + ((hsel ^ ((slot >> 1) & 7)) << 4);
  }
  {
    const int slot = wn * 32 + (lane & 31);
    wbase_off = slot * ROWB + ((hsel ^ ((slot >> 1) & 7)) << 4);
  }

  // lane part of the DMA source offset: row (lane >> 3) of the 8-row group + the swizzled 16-byte chunk.  With slots
  // dealt as (wave + 8 j) * 8 + r8 the swizzle term f(slot) = ((wave & 1) * 4 + (r8 >> 1)) is the same for both DMAs.
  unsigned la, lw;
  {
    const int r8 = lane >> 3;
    const int logical = (lane & 7) ^ (((wave & 1) << 2) | (r8 >> 1));
    const int memchunk = (logical & ~3) | ((logical & 1) << 1) | ((logical >> 1) & 1);
    la = (unsigned)r8 * (unsigned)p.lda * 4u + memchunk * 16;
    lw = (unsigned)r8 * (unsigned)k16 * 4u + memchunk * 16;
  }

  for (int sg = 0; sg < nseg; ++sg) {
    // which piece: head (park the accumulators), whole tile, or tail (resume from parked accumulators)
    const bool is_head = has_head && sg == 0;
    const bool is_tail = has_tail && sg == nseg - 1;
    int tile = is_head ? seg_last_tile : (is_tail ? seg_first_tile : full0 + sg - has_head);
    if (PERSIST) tile = tile * ngrp + (int)(blockIdx.x >> 3) % ngrp;
    const int kb = is_tail ? seg_o : 0;
    const int nks = (is_head ? seg_e : nk) - kb;                  // K tiles of this piece (>= 1)
    const int grp = tile / gsz, rem = tile - grp * gsz;
    const int bm = rem / ngrp, bn = grp * ngrp + (rem - bm * ngrp);
    const int m0 = bm * BM, n0 = bn * BN;
    if ((VAR & P8VAR_RETR) && n0 >= p.n) continue;               // row-tile groups are padded to eight tiles: nothing to do

    // request K tile kt (relative to kb) of one operand region into LDS offset `reg` (buffer / ring slot included):
    // two 1 KB DMAs per wave.  half: 0 = A0 / W0, 1 = A1 / W1.  The 8-row group's first row is wave-uniform and rides
    // on the scalar base together with the K offset; the lane part (la / lw) never changes.
    auto stage = [&](int kt, int reg, bool is_w, int half) {
      if (ABL & 1) return;
      const unsigned char* base = is_w ? wsp : abase;
      const unsigned pitch = (unsigned)(is_w ? k16 : p.lda) * 4u;
      // retrieval: any row counts — the operand buffers of wd_split_weights are padded to whole groups of eight rows
      // arows / wrows: rows the operand BUFFERS hold, whole groups of eight (= p.m / p.n for the plain launches, which require
      // multiples of eight; the similarity launch and the retrieval pad their buffers instead)
      const int limit = (VAR & P8VAR_RETR) ? (((is_w ? p.n : p.m) + 7) & ~7) - 8 : (is_w ? wrows : arows) - 8;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = is_w ? n0 + (2 * j + (wave >> 2)) * 64 + (wave & 3) * 8 + half * 32 : m0 + j * 128 + wave * 8 + half * 64;
        row = row < limit ? row : limit;                            // whole groups past the end re-read the last one
        const unsigned char* src = base + (size_t)row * pitch + (size_t)(kb + kt) * ROWB;
        p8_dma(dma_dst + reg + j * 8192, is_w ? lw : la, src);
      }
    };

    f32x16 acc[TM][TN];
    // ---- prologue: K tiles 0 and 1 of the piece (the previous piece's epilogue is done with the LDS)
    P8_BARRIER();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
      if (kt < nks) {
        stage(kt, kt * P8_TILE + P8_A0, false, 0);
        stage(kt, P8_W0RING + kt * P8_REGION, true, 0);
        stage(kt, kt * P8_TILE + P8_W1, true, 1);
        stage(kt, kt * P8_TILE + P8_A1, false, 1);
      }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // The tail of a tile resumes from the accumulators its head parked (written by the previous workgroup of the
    // chain at the START of its work).  They are NOT loaded here: each phase of the piece's first K tile loads the
    // two accumulators it is about to use (ldacc below) — 128 live accumulator registers from this point on make
    // hipcc spill inside the K loop, accumulators that become live one pair per phase do not.
    const unsigned char* park_src = nullptr;
    if (PERSIST && is_tail) {
      const int prev = blockIdx.x - 8 * ngrp;
      if (t == 0) {
        while (__hip_atomic_load(ps.flag + prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
      park_src = reinterpret_cast<const unsigned char*>(ps.acc) + (size_t)prev * P8_PARK_BYTES + (unsigned)t * 512u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    P8_BARRIER();
    if (group == 1 && !(ABL & 24)) P8_BARRIER();                   // the second row group runs one barrier behind

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    h8 xh[2][2] = {}, xl[2][2] = {}, wh[2] = {}, wl[2] = {};       // activation fragments [row block][k16 step]; weights [k16 step]
    h8 w0h[2] = {}, w0l[2] = {};                                   // W0 stays in registers from phase 1 to phase 4
    auto read_a = [&](const unsigned char* region) {
      if (ABL & 2) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(xh[i][ks]), "+v"(xl[i][ks]));
        return;
      }
      int b = abase_off;
      asm volatile("" : "+v"(b));                                  // keep the xor variants out of long-lived registers
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          xh[i][ks] = *reinterpret_cast<const h8*>(region + i * 4096 + (b ^ (ks * 64)));
          xl[i][ks] = *reinterpret_cast<const h8*>(region + i * 4096 + (b ^ (ks * 64 + 32)));
        }
    };
    auto read_w = [&](const unsigned char* region, h8 (&fh)[2], h8 (&fl)[2]) {
      if (ABL & 2) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) asm volatile("" : "+v"(fh[ks]), "+v"(fl[ks]));
        return;
      }
      int b = wbase_off;
      asm volatile("" : "+v"(b));
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        fh[ks] = *reinterpret_cast<const h8*>(region + (b ^ (ks * 64)));
        fl[ks] = *reinterpret_cast<const h8*>(region + (b ^ (ks * 64 + 32)));
      }
    };
    // 12 MFMAs of one quadrant: two 32 x 32 accumulators
    auto quadrant = [&](f32x16& c0, f32x16& c1, const h8 (&fh)[2], const h8 (&fl)[2]) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], xh[0][ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[ks], xh[1][ks], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], xl[0][ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], xl[1][ks], c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], xh[0][ks], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[ks], xh[1][ks], c1, 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    };

    // One K tile = four phases.  Counted waits: in steady state the five newest region requests (10 DMA
    // instructions of this wave) stay in flight; the last two K tiles request less, so less may be left
    // outstanding (TAIL 1 = tile nks - 2, TAIL 2 = tile nks - 1).  S1 / S234: whether phase 1 / phases 2-4 request a
    // region (tile 0's phase 1 does not: the prologue already did; the last tiles have nothing left to request) —
    // compile-time, so the loop body is branch-free.  w0s = W0 ring slot of this K tile (kt % 3).
#define P8_WAIT(a, b, c)                                                                       \
    do {                                                                                         \
      if constexpr (TAIL == 0) asm volatile("s_waitcnt vmcnt(" #a ")" ::: "memory");             \
      else if constexpr (TAIL == 1) asm volatile("s_waitcnt vmcnt(" #b ")" ::: "memory");        \
      else asm volatile("s_waitcnt vmcnt(" #c ")" ::: "memory");                                 \
    } while (0)
    // accumulators a0 = i0 * TN + j0 and a1 of this lane <- parked block (64 bytes each), loads + wait in one statement
    auto ldacc = [&](f32x16& c0, f32x16& c1, int a0, int a1) {
      f32x4 q0, q1, q2, q3, q4, q5, q6, q7;
      const unsigned char* p0 = park_src + a0 * 64;
      const unsigned char* p1 = park_src + a1 * 64;
      asm volatile("global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
                   "global_load_dwordx4 %2, %8, off offset:32\n\tglobal_load_dwordx4 %3, %8, off offset:48\n\t"
                   "global_load_dwordx4 %4, %9, off\n\tglobal_load_dwordx4 %5, %9, off offset:16\n\t"
                   "global_load_dwordx4 %6, %9, off offset:32\n\tglobal_load_dwordx4 %7, %9, off offset:48\n\t"
                   "s_waitcnt vmcnt(0)"
                   : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3), "=&v"(q4), "=&v"(q5), "=&v"(q6), "=&v"(q7)
                   : "v"(p0), "v"(p1) : "memory");
      c0 = __builtin_shufflevector(__builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q2, q3, 0, 1, 2, 3, 4, 5, 6, 7),
                                   0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
      c1 = __builtin_shufflevector(__builtin_shufflevector(q4, q5, 0, 1, 2, 3, 4, 5, 6, 7), __builtin_shufflevector(q6, q7, 0, 1, 2, 3, 4, 5, 6, 7),
                                   0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    };
    bool resume = PERSIST && is_tail;                              // first K tile of a tail piece: load accumulators per phase
    int w0s = 0;
    auto ktile = [&](int kt, auto tail_c, auto s1_c, auto s234_c) {
      constexpr int TAIL = decltype(tail_c)::value;
      constexpr bool S1 = decltype(s1_c)::value, S234 = decltype(s234_c)::value;
      const int cur = (kt & 1) * P8_TILE, oth = P8_TILE - cur;
      const unsigned char* buf = smem_raw + cur;
      const unsigned char* w0 = smem_raw + P8_W0RING + w0s * P8_REGION;
      const int w0_free = P8_W0RING + (w0s == 0 ? 2 : w0s - 1) * P8_REGION;   // slot (kt + 2) % 3: its K tile kt - 1 is done
      // phase 1
      read_a(buf + P8_A0);
      read_w(w0, w0h, w0l);
      if constexpr (S1) stage(kt + 1, oth + P8_A1, false, 1);
      if (PERSIST && resume) ldacc(acc[0][0], acc[1][0], 0 * TN + 0, 1 * TN + 0);
      P8_WAIT(10, 10, 2);
      P8_KBARRIER();
      quadrant(acc[0][0], acc[1][0], w0h, w0l);
      P8_KBARRIER();
      // phase 2
      read_w(buf + P8_W1, wh, wl);
      if constexpr (S234) stage(kt + 2, w0_free, true, 0);
      if (PERSIST && resume) ldacc(acc[0][1], acc[1][1], 0 * TN + 1, 1 * TN + 1);
      P8_WAIT(10, 8, 0);
      P8_KBARRIER();
      quadrant(acc[0][1], acc[1][1], wh, wl);
      P8_KBARRIER();
      // phase 3
      read_a(buf + P8_A1);
      if constexpr (S234) stage(kt + 2, cur + P8_A0, false, 0);
      if (PERSIST && resume) ldacc(acc[2][1], acc[3][1], 2 * TN + 1, 3 * TN + 1);
      P8_WAIT(10, 6, 0);
      P8_KBARRIER();
      quadrant(acc[2][1], acc[3][1], wh, wl);
      P8_KBARRIER();
      // phase 4: W0 is still in registers
      if constexpr (S234) stage(kt + 2, cur + P8_W1, true, 1);
      if (PERSIST && resume) ldacc(acc[2][0], acc[3][0], 2 * TN + 0, 3 * TN + 0);
      P8_WAIT(10, 4, 0);
      P8_KBARRIER();
      quadrant(acc[2][0], acc[3][0], w0h, w0l);
      P8_KBARRIER();
      w0s = w0s == 2 ? 0 : w0s + 1;
      if (PERSIST && resume) {                                     // every parked value is in registers: release the slot
        resume = false;
        __syncthreads();
        if (t == 0) __hip_atomic_store(ps.flag + (blockIdx.x - 8 * ngrp), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    };

#undef P8_WAIT
    if (nks >= 3) {
      ktile(0, I0{}, std::false_type{}, std::true_type{});
      for (int kt = 1; kt + 2 < nks; ++kt) ktile(kt, I0{}, std::true_type{}, std::true_type{});
      ktile(nks - 2, I1{}, std::true_type{}, std::false_type{});
      ktile(nks - 1, I2{}, std::false_type{}, std::false_type{});
    } else if (nks == 2) {                                         // everything came with the prologue
      ktile(0, I1{}, std::false_type{}, std::false_type{});
      ktile(1, I2{}, std::false_type{}, std::false_type{});
    } else {
      ktile(0, I2{}, std::false_type{}, std::false_type{});
    }
    if (group == 0 && !(ABL & 24)) P8_BARRIER();                   // pairs with the extra barrier of group 1
    P8_BARRIER();                                                  // every read of the operand tiles is over: LDS becomes epilogue patches

    if (PERSIST && is_head) {
      // park the raw accumulators for the next workgroup of the chain: plain coalesced stores, drained, then ONE
      // agent-scope release and the flag (MI355X_MICROARCH.md, inter-workgroup visibility)
      f32x4* lp = reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(ps.acc) + (size_t)blockIdx.x * P8_PARK_BYTES +
                                           (unsigned)t * 512u);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            lp[(i * TN + j) * 4 + g] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          __builtin_amdgcn_sched_barrier(0);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(ps.flag + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;
    }
    if (ABL & 4) {                                                  // keep the accumulators live, store (almost) nothing
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
      if (sum == 12345.678f) p.c[t] = sum;
      continue;
    }
    if constexpr ((VAR & P8VAR_RETR) != 0) {
      p8_retr_epilogue<TM, TN>(p, rt, unscale, m0 + group * 128, n0 + wn * 64, lane, acc);
      continue;
    }
    const EpiVec ev{vec_c, vec_res, vec_bias, unscale};
    const int mw = m0 + group * 128, nw = n0 + wn * 64;
    float* patch = reinterpret_cast<float*>(smem_raw) + wave * 32 * EPI_LDT;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));          // epilogue address arithmetic starts HERE (hipcc otherwise computes it before the K loop and keeps ~50 registers alive across it)
    static_assert(P8_LDS >= 8 * 32 * EPI_LDT * 4, "operand LDS must hold one patch per wave");
    if constexpr ((VAR & P8VAR_DIRECT) != 0) {                      // hi/lo output without the LDS transpose (experiment, see split_gemm_impl.h)
      if (p.act == WD_ACT_GELU) EpiCsplitDirectWalk<0, TM, TN, WD_ACT_GELU>::run(p, ev, mw, nw, lane_e, acc);
      else EpiCsplitDirectWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane_e, acc);
    } else if (VAR & SVAR_CSPLIT) {
      switch (p.act) {
        case WD_ACT_RELU: EpiCsplitWalk<0, TM, TN, WD_ACT_RELU>::run(p, ev, mw, nw, lane_e, acc, patch); break;
        case WD_ACT_SILU: EpiCsplitWalk<0, TM, TN, WD_ACT_SILU>::run(p, ev, mw, nw, lane_e, acc, patch); break;
        case WD_ACT_GELU: EpiCsplitWalk<0, TM, TN, WD_ACT_GELU, (ABL & 96)>::run(p, ev, mw, nw, lane_e, acc, patch); break;
        default: EpiCsplitWalk<0, TM, TN, WD_ACT_NONE>::run(p, ev, mw, nw, lane_e, acc, patch); break;
      }
    } else if (epi_res_prefetch_ok(p, ev, nw, 64) && mw < p.m) {
      split_epilogue_res_prefetch<TM, TN, 3>(p, ev, mw, nw, lane_e, acc, patch);
    } else if (!PERSIST && (p.seg_rows > 0 || p.sigmoid || p.out_scale != 1.0f || p.out_bias != 0.0f)) {
      // round 6 (wd_similarity_split): per-level affine + sigmoid of the region x text logits, ragged n; no activation
      split_epilogue_lds<TM, TN, WD_ACT_NONE, true>(p, ev, mw, nw, lane_e, acc, patch);
    } else {
      switch (p.act) {
        case WD_ACT_RELU: split_epilogue_lds<TM, TN, WD_ACT_RELU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
        case WD_ACT_SILU: split_epilogue_lds<TM, TN, WD_ACT_SILU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
        case WD_ACT_GELU: split_epilogue_lds<TM, TN, WD_ACT_GELU, false>(p, ev, mw, nw, lane_e, acc, patch); break;
        default: split_epilogue_lds<TM, TN, WD_ACT_NONE, false>(p, ev, mw, nw, lane_e, acc, patch); break;
      }
    }
  }
}

constexpr long long P8_PARK_FLOATS = P8_PARK_BYTES / 4;

// one workgroup per CU, the same number on every XCD — a fact of the CURRENT device (a process may drive several)
int p8_workgroups() {
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

int p8_stagger() {
#ifdef WD_DEBUG_ABLATIONS                               // the CU-phase experiment of profiles/r05_p8_epilogue_ablation.txt: ablation builds only
  const char* e = getenv("WD_P8_STAGGER");
  return e ? atoi(e) : 0;
#else
  return 0;
#endif
}

// gang size of the persistent form for nbn column tiles: the largest of 8 / 4 / 2 / 1 that divides nbn (and the 32
// workgroups of an XCD)
int p8_gang(int nbn) { return nbn % 8 == 0 ? 8 : nbn % 4 == 0 ? 4 : nbn % 2 == 0 ? 2 : 1; }

template <int VAR, int ABL = 0, bool PERSIST = false>
int launch_p8(const WdConvGemm& p, const void* wsp, float unscale, hipStream_t st, float* ws = nullptr, long long ws_floats = 0,
              int arows = 0, int wrows = 0) {
  const int nbm = (p.m + 255) / 256, nbn = (p.n + 255) / 256;
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  // 32-bit DMA offsets from the operand bases
  if ((unsigned long long)(arows > 0 ? arows : p.m) * p.lda * 4 >= (1ull << 32) || (unsigned long long)(wrows > 0 ? wrows : p.n) * k16 * 4 >= (1ull << 32))
    return WD_ERR_UNSUPPORTED;
  const int vec_c = (p.ldc % 4 == 0) && wd_aligned16(p.c);
  const int vec_res = p.res ? ((p.ldres % 4 == 0) && wd_aligned16(p.res)) : 0;
  const int vec_bias = p.bias ? wd_aligned16(p.bias) : 0;
  // Column tiles of one row block are walked together, up to eight at a time: the activation panel of a row block is then
  // fetched once for all of them (a pwconv1 with n = 2048 re-read it per pair before: 42.79 -> 42.05 ms per step, same
  // box, three rounds; pairs = 2, the round-2 default until then, and 4 / 8 equal).  Up to 2048 weight rows live per group.
  int ngrp = 8;
  if (ngrp > nbn || nbn % ngrp) ngrp = nbn;
  P8Persist ps{nullptr, nullptr, nullptr, (int)nblk};
  long long grid = nblk;
  if (PERSIST) {
    const int wgs = p8_workgroups();
    if (wgs <= 0) return WD_ERR_LAUNCH;
    ngrp = p8_gang(nbn);
    const int per = wgs / 8;
    if (per % ngrp) return WD_ERR_UNSUPPORTED;
    ps.ntiles = (int)(nblk / ngrp);                                // gang tiles
    if (ps.ntiles / 8 < per / ngrp) return WD_ERR_UNSUPPORTED;     // ranges shorter than a tile would chain serially
    if (!ws || ws_floats < wd_p8_workspace_floats()) return WD_ERR_WORKSPACE;
    ps.flag = reinterpret_cast<unsigned*>(ws);                     // first 4 KB: flags (zero between launches); [512, 516): zeros
    ps.zero = ws + 512;
    ps.acc = ws + 1024;
    grid = wgs;
  }
  auto k = split_gemm_p8_kernel<VAR, ABL, PERSIST>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), P8_LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)grid), dim3(512), P8_LDS, st, p, static_cast<const unsigned char*>(wsp), k16, unscale, nbn,
                 vec_c, vec_res, vec_bias, ngrp, nbm, ps, P8Retr{}, p8_stagger(), arows > 0 ? arows : p.m, wrows > 0 ? wrows : p.n);
  return wd_launch_status();
}

// retrieval scoring: p.a = bank rows [p.m][p.lda] (fp16 hi/lo groups), wsp = region rows [p.n][k16], see P8Retr
int launch_p8_retr(const WdConvGemm& p, const void* wsp, float unscale, const P8Retr& rt, hipStream_t st) {
  const int nbm = (p.m + 255) / 256, nbn = ((p.n + 255) / 256 + 7) / 8 * 8;      // row tiles padded to whole groups of eight
  const long long nblk = (long long)nbm * nbn;
  if (nblk <= 0 || nblk > 0x7fffffffLL) return WD_ERR_BAD_ARG;
  const int k16 = (p.k + 15) / 16 * 16;
  if ((unsigned long long)((p.m + 7) & ~7) * p.lda * 4 >= (1ull << 32) || (unsigned long long)((p.n + 7) & ~7) * k16 * 4 >= (1ull << 32))
    return WD_ERR_UNSUPPORTED;
  auto k = split_gemm_p8_kernel<P8VAR_RETR, 0, false>;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(k), P8_LDS) != WD_OK) return WD_ERR_LAUNCH;
  WD_LAUNCH_GEMM(k, dim3((unsigned)nblk), dim3(512), P8_LDS, st, p, static_cast<const unsigned char*>(wsp), k16, unscale, nbn,
                 0, 0, 0, 8, nbm, P8Persist{nullptr, nullptr, nullptr, (int)nblk}, rt, 0, 0, 0);
  return wd_launch_status();
}

}  // namespace

// does the persistent form apply to an m x n problem on the current device?  (every gang of every XCD gets at least one
// whole gang tile)
bool wd_p8_persist_ok(int m, int n) {
  const int wgs = p8_workgroups();
  if (wgs <= 0 || n % 256) return false;
  const int nbm = (m + 255) / 256, nbn = n / 256, g = p8_gang(nbn), per = wgs / 8;
  if (per % g) return false;
  return (long long)nbm * (nbn / g) / 8 >= per / g;
}

// workspace of the persistent form: 1024 flag words + one parked accumulator set per workgroup (fp32 elements)
long long wd_p8_workspace_floats() {
  const int wgs = p8_workgroups();
  return wgs <= 0 ? 0 : 1024 + (long long)wgs * P8_PARK_FLOATS;
}

// cfg 64: plain 1x1 layer, both operands pre-split, K % 32 == 0.  csplit: output written as fp16 hi/lo groups.
// persist (cfg 65): the work-unit form; ws = caller's workspace, zero-filled once before its first use.
int wd_launch_p8(const WdConvGemm& p, const void* w, float unscale, bool csplit, hipStream_t st, int abl, bool persist,
                 float* ws, long long ws_floats) {
#ifdef WD_DEBUG_ABLATIONS
  if (abl && !persist) {
    if (csplit) return abl == 4 ? launch_p8<SVAR_CSPLIT, 4>(p, w, unscale, st) : abl == 32 ? launch_p8<SVAR_CSPLIT, 32>(p, w, unscale, st)
                     : abl == 64 ? launch_p8<SVAR_CSPLIT, 64>(p, w, unscale, st) : WD_ERR_UNSUPPORTED;
    switch (abl) {
      case 1: return launch_p8<0, 1>(p, w, unscale, st);
      case 2: return launch_p8<0, 2>(p, w, unscale, st);
      case 3: return launch_p8<0, 3>(p, w, unscale, st);
      case 4: return launch_p8<0, 4>(p, w, unscale, st);
      case 7: return launch_p8<0, 7>(p, w, unscale, st);
      case 15: return launch_p8<0, 15>(p, w, unscale, st);
      case 16: return launch_p8<0, 16>(p, w, unscale, st);
      default: return WD_ERR_UNSUPPORTED;
    }
  }
#endif
  if (abl) return WD_ERR_UNSUPPORTED;
  if (p.k % 32 || p.k < 32 || p.lda % 8 || p.m % 8 || p.n % 8 || p.m < 8 || p.n < 8 || !wd_aligned16(p.a) || !wd_aligned16(w))
    return WD_ERR_UNSUPPORTED;                                     // DMA row groups of 8 are clamped as a whole
  if (csplit) {
    if (p.res || p.n % 8 || p.ldc % 8 || (p.bias && !wd_aligned16(p.bias)) || !wd_aligned16(p.c)) return WD_ERR_BAD_ARG;
#ifdef WD_DEBUG_ABLATIONS       // the direct hi/lo epilogue (no LDS transpose): measured 1.5 - 3 x slower, profiles/r05_csplit_direct.txt; ablation builds only
    const char* e = getenv("WEDETECT_CSPLIT_DIRECT");
    if (e && e[0] == '1' && (p.act == WD_ACT_GELU || p.act == WD_ACT_NONE))
      return persist ? launch_p8<SVAR_CSPLIT | P8VAR_DIRECT, 0, true>(p, w, unscale, st, ws, ws_floats)
                     : launch_p8<SVAR_CSPLIT | P8VAR_DIRECT>(p, w, unscale, st);
#endif
    return persist ? launch_p8<SVAR_CSPLIT, 0, true>(p, w, unscale, st, ws, ws_floats) : launch_p8<SVAR_CSPLIT>(p, w, unscale, st);
  }
  return persist ? launch_p8<0, 0, true>(p, w, unscale, st, ws, ws_floats) : launch_p8<0>(p, w, unscale, st);
}

// wd_retrieval_max_split on the 256 x 256 kernel (operand roles swapped, see P8Retr): bank rows [n_cls][dim] as the
// activation operand, region rows [n_rows][dim] as the weight operand; out [images][ldo] must be zero.  Needs dim % 32 == 0,
// 16-byte aligned scale / bias, both split buffers padded to whole groups of eight rows (wd_split_weights does that).
int wd_launch_p8_retrieval(const void* t_split, int n_cls, const void* e_split, int n_rows, int dim, float unscale,
                           const float* scale, const float* bias, const int* count, int rows_per_img, float* out, int ldo,
                           unsigned* range_flag, hipStream_t st) {
  if (dim % 32 || dim < 32 || n_cls <= 0 || n_rows <= 0 || rows_per_img <= 0) return WD_ERR_UNSUPPORTED;
  if (!wd_aligned16(t_split) || !wd_aligned16(e_split) || !wd_aligned16(scale) || !wd_aligned16(bias)) return WD_ERR_UNSUPPORTED;
  WdConvGemm p{};
  p.a = static_cast<const float*>(t_split);
  p.c = out;
  p.batch = 1; p.hin = 1; p.win = n_cls; p.cin = dim; p.lda = dim;
  p.kh = p.kw = p.stride = 1; p.hout = 1; p.wout = n_cls;
  p.m = n_cls; p.n = n_rows; p.k = dim; p.ldc = ldo;
  p.out_scale = 1.0f;
  p.range_flag = range_flag;
  return launch_p8_retr(p, e_split, unscale, P8Retr{scale, bias, count, out, rows_per_img, ldo}, st);
}

// wd_similarity_split on the 256 x 256 kernel (round 6; yolo_world_head.py:90-108, generate_proposal.py:1129-1131): region rows
// [rows][dim] as fp16 hi/lo groups (buffer padded to a multiple of eight rows) x text rows [n_cls][dim] split by
// wd_split_weights_padded; out[row][cls] = (sigmoid)(<e, t> * unscale * seg_scale[level(row)] + seg_bias[level(row)]), the
// epilogue of the fp32 similarity GEMM.  Ragged n_cls (1203) and ldo are fine (scalar stores when ldo % 4).
int wd_launch_p8_similarity(const WdConvGemm& p, const void* t_split, float unscale, hipStream_t st) {
  if (p.k % 32 || p.k < 32 || p.lda % 8 || !wd_aligned16(p.a) || !wd_aligned16(t_split)) return WD_ERR_UNSUPPORTED;
  if (p.res || p.c2 || p.bias || p.act != WD_ACT_NONE || p.out_mode != WD_OUT_ROWS || p.c_batch_stride > 0 || p.ln_stats) return WD_ERR_UNSUPPORTED;
  return launch_p8<0>(p, t_split, unscale, st, nullptr, 0, (p.m + 7) & ~7, (p.n + 7) & ~7);
}
