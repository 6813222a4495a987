// probe.hip — diagnostic micro-benchmarks (not on the product path): what the hardware gives the LDS-fed GEMM kernels.
#include "common.h"

namespace {

__device__ __forceinline__ void probe_dma16(unsigned lds_addr, const unsigned char* src) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src) : "memory", "m0");
}

// Every wave streams `iters` x 8 LDS-DMA instructions (1 KB each) from its workgroup's window of `window` bytes
// (walked cyclically: a small window is cache-resident, a large one streams from HBM) into the workgroup's LDS ring,
// keeping 8-16 instructions in flight.  pattern 0: 1 KB contiguous per instruction; pattern 1: 16 rows x 64 B with a
// row pitch of `pitch` bytes — the access shape of a K = 16 operand stage of the GEMM kernels.
__global__ void __launch_bounds__(512) lds_dma_probe_kernel(const unsigned char* __restrict__ src, long long window, int iters,
                                                            int pattern, int pitch, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw + wave * 16 * 1024;
  const unsigned char* base = src + (size_t)blockIdx.x * window;
  long long off = (long long)wave * (pattern == 0 ? 1024 : 16LL * pitch);
  const long long step = 8LL * (pattern == 0 ? 1024 : 16LL * pitch);      // 8 waves interleave their pieces
  const int lane_off = pattern == 0 ? lane * 16 : (lane >> 2) * pitch + (lane & 3) * 16;
  const long long span = pattern == 0 ? 1024 : 16LL * pitch;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (off + span > window) off -= (window / step) * step;
      if (off < 0) off = 0;
      probe_dma16(__builtin_amdgcn_readfirstlane(lds0 + ((it & 1) * 8 + j) * 1024), base + off + lane_off);
      off += step;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = reinterpret_cast<const unsigned*>(smem_raw)[blockIdx.x & 1023];
}

// ---------------------------------------------------------------------------------------
// MFMA / VALU issue probe: what does a wave pay for VALU instructions placed between its own MFMAs?
// Every wave runs `iters` iterations of 8 x { one v_mfma_f32_32x32x16_f16 (four independent accumulators) + NV VALU
// instructions of kind KIND (0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32, 3: v_mul_f32) on eight independent chains }.
// MODE 0: both, interleaved; 1: the MFMAs only; 2: the VALU only.  out[0..1] = shader-clock ticks (s_memtime) and
// 100 MHz ticks (s_memrealtime) of workgroup 0 wave 0.
// ---------------------------------------------------------------------------------------
typedef _Float16 ph8 __attribute__((ext_vector_type(8)));
typedef float pf16 __attribute__((ext_vector_type(16)));
typedef float pf2 __attribute__((ext_vector_type(2)));

template <int MODE, int KIND, int NV>
__global__ void __launch_bounds__(256) issue_probe_kernel(int iters, float seed, unsigned long long* __restrict__ out, float* __restrict__ sink) {
  ph8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 1e-3f + i); b[i] = (_Float16)(seed * 0.5f + i); }
  pf16 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = seed;
  pf2 x[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) x[c] = pf2{seed + c, seed - c};
  const pf2 m = {seed * 0.999f, seed * 1.001f}, d = {1e-3f, 2e-3f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 3 || MODE == 4) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);   // one dependent chain
      else if (MODE != 2) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
      if (MODE != 1 && MODE != 3) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const int c = (u * NV + v) & 7;
          if (KIND == 0) x[c][0] = __builtin_fmaf(x[c][0], m[0], d[0]);
          else if (KIND == 1) x[c] = __builtin_elementwise_fma(x[c], m, d);
          else if (KIND == 2) x[c][0] = __builtin_amdgcn_exp2f(x[c][0]);
          else if (KIND == 3) x[c][0] = x[c][0] * m[0];
          else if (KIND == 4) {                       // v_cvt_pk_f16_f32
            typedef _Float16 ph2 __attribute__((ext_vector_type(2)));
            const ph2 h = __builtin_convertvector(x[c], ph2);
            x[c][0] = __builtin_bit_cast(float, h);
          } else if (KIND == 5) x[c] = x[c] * m;      // v_pk_mul_f32
          else if (KIND == 6) x[c] = x[c] + d;        // v_pk_add_f32
          else if (KIND == 7) x[c][0] = x[c][1] >= 0.f ? x[c][0] : m[0];   // v_cmp + v_cndmask
          else if (KIND == 8) x[c][0] = __builtin_amdgcn_rcpf(x[c][0]);
          else if (KIND == 9) {
            const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x[c][0]), __builtin_bit_cast(unsigned, x[c][1]), false, false);
            x[c][0] = __builtin_bit_cast(float, r[0]); x[c][1] = __builtin_bit_cast(float, r[1]);
          } else if (KIND == 10) x[c][0] = (float)__builtin_bit_cast(_Float16, (unsigned short)__builtin_bit_cast(unsigned, x[c][0]));  // v_cvt_f32_f16
          else if (KIND == 11) x[c][0] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x[c][0]) & 0x7fffffffu) ;   // v_and
          else if (KIND == 13) x[0][0] = __builtin_fmaf(x[0][0], m[0], d[0]);                  // one dependent chain
          else if (KIND == 14) x[c & 1][0] = __builtin_fmaf(x[c & 1][0], m[0], d[0]);          // two chains
          else if (KIND == 15) x[c & 3][0] = __builtin_fmaf(x[c & 3][0], m[0], d[0]);          // four chains
          else if (KIND == 16) x[c & 3][0] = __builtin_fmaf(x[c & 3][0], 1.061405429f, -1.453152027f);   // four chains, literal constants (v_fmaak)
          else if (KIND == 12) x[c][0] = __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x[c][0]));  // v_cvt_f16_f32
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float z = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) z += acc[k][0] + acc[k][7];
#pragma unroll
  for (int c = 0; c < 8; ++c) z += x[c][0] + x[c][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = z;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

}  // namespace

// bytes moved = grid * 8 waves * iters * 8 KB.  window_bytes per workgroup; src must hold grid * window_bytes bytes.
extern "C" int wd_probe_lds_dma(const void* src, int64_t window_bytes, int32_t grid, int32_t iters, int32_t pattern,
                                int32_t pitch_bytes, void* sink, void* stream) {
  if (!src || !sink || grid <= 0 || iters <= 0 || window_bytes < 128 * 1024 || (pattern != 0 && pattern != 1)) return WD_ERR_BAD_ARG;
  if (pattern == 1 && (pitch_bytes < 64 || pitch_bytes % 16 || 128LL * pitch_bytes > window_bytes)) return WD_ERR_BAD_ARG;
  static WdAttrOnce attr;
  if (wd_set_max_lds(attr, reinterpret_cast<const void*>(lds_dma_probe_kernel), 128 * 1024) != WD_OK) return WD_ERR_LAUNCH;
  hipLaunchKernelGGL(lds_dma_probe_kernel, dim3(grid), dim3(512), 128 * 1024, static_cast<hipStream_t>(stream),
                     static_cast<const unsigned char*>(src), (long long)window_bytes, iters, pattern, pitch_bytes,
                     static_cast<unsigned*>(sink));
  return wd_launch_status();
}

// grid workgroups of 256 threads (one wave per SIMD; two workgroups per CU co-reside: grid 512 = two waves per SIMD).
// sink: grid * 256 floats; out: 2 x uint64.
extern "C" int wd_probe_issue(int32_t mode, int32_t kind, int32_t nv, int32_t grid, int32_t iters, void* out, void* sink, void* stream) {
  if (!out || !sink || grid <= 0 || iters <= 0) return WD_ERR_BAD_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  auto* o = static_cast<unsigned long long*>(out);
  auto* sk = static_cast<float*>(sink);
#define WD_IP(M, K, N) if (mode == M && kind == K && nv == N) { hipLaunchKernelGGL((issue_probe_kernel<M, K, N>), dim3(grid), dim3(256), 0, st, iters, 1.0f, o, sk); return wd_launch_status(); }
#define WD_IPK(K, N) WD_IP(0, K, N) WD_IP(2, K, N)
  WD_IP(0, 0, 10) WD_IP(2, 0, 10)
  WD_IP(1, 0, 0) WD_IP(3, 0, 0) WD_IP(4, 0, 4) WD_IP(4, 0, 6) WD_IP(4, 0, 8) WD_IP(4, 0, 10)
  WD_IPK(0, 2) WD_IPK(0, 4) WD_IPK(0, 6) WD_IPK(0, 7) WD_IPK(0, 8)
  WD_IPK(1, 4) WD_IPK(1, 7) WD_IPK(2, 2) WD_IPK(2, 4) WD_IPK(3, 7)
  WD_IPK(4, 4) WD_IPK(5, 4) WD_IPK(6, 4) WD_IPK(7, 3) WD_IPK(8, 2) WD_IPK(8, 4) WD_IPK(9, 4) WD_IPK(10, 6) WD_IPK(11, 6) WD_IPK(12, 6) WD_IPK(13, 6) WD_IPK(14, 6) WD_IPK(15, 6) WD_IPK(16, 6) WD_IPK(13, 3)
#undef WD_IPK
#undef WD_IP
  return WD_ERR_UNSUPPORTED;
}
