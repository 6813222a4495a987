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
