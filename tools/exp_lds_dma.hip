// Experiment (run by tools/exp_lds_dma.py on the GPU box): what one CU sustains on the global -> LDS path
// (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction) as a function of the source access shape — the constant
// behind the wide-tile main loop of gemm2.hip, where the segment timing (profiles/r03/i_*) showed ~100 cycles per DMA
// instruction.  One 512-thread block per CU; every wave issues `per_wave` DMA instructions per round into its own LDS
// slice, then s_waitcnt vmcnt(0) + s_barrier (the drain is the point: bytes per cycle per CU at a given queue depth).
//   row_bytes   contiguous bytes per source row touched by one instruction (128: 8 rows x 128 B as the BK = 64 tiles,
//               64: 16 rows x 64 B as the BK = 32 tiles, 1024: one contiguous KiB)
//   ld_bytes    source row stride (the matrix's K * 2)
//   span_bytes  bytes of source the block cycles through (small: L2-resident, large: streaming from HBM / Infinity Cache)
//   swizzle     1: the XOR chunk swizzle of the kernels on the source address
#include <hip/hip_runtime.h>
#include <stdint.h>

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ src, unsigned long long* cyc, uint32_t* sink, int rounds,
                                                  int per_wave, int row_bytes, int ld_bytes, long long span_bytes, int swizzle,
                                                  int waves_active) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long block_base = (long long)blockIdx.x * span_bytes;
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + block_base), 0, (uint32_t)span_bytes, 0x00020000);
  const int cpr = row_bytes >> 4;                 // 16-B chunks per row piece
  const int lr = lane / cpr, lc = lane % cpr;     // row within the instruction, chunk within the row piece
  const int rpi = 64 / cpr;                       // rows per instruction
  const int chunk = swizzle ? (lc ^ (lr & (cpr - 1))) : lc;
  char* my = smem + wave * (per_wave * 1024);
  uint32_t pos = (uint32_t)wave * (uint32_t)(per_wave * rpi) * (uint32_t)ld_bytes;   // this wave's first row
  const uint32_t step = 8u * (uint32_t)(per_wave * rpi) * (uint32_t)ld_bytes;          // all waves advance together
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    if (wave < waves_active) {
      for (int i = 0; i < per_wave; ++i) {
        uint32_t off = pos + (uint32_t)(i * rpi + lr) * (uint32_t)ld_bytes + (uint32_t)chunk * 16u;
        if (off + 16u > (uint32_t)span_bytes) off %= (uint32_t)(span_bytes - 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(my + i * 1024), 16, off & ~15u, 0, 0, 0);
      }
    }
    pos += step;
    if (pos + step > (uint32_t)span_bytes) pos = (uint32_t)wave * (uint32_t)(per_wave * rpi) * (uint32_t)ld_bytes;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (threadIdx.x == 0) sink[blockIdx.x] = *(const uint32_t*)smem;
}

extern "C" int exp_dma(const void* src, unsigned long long* cyc, uint32_t* sink, int blocks, int rounds, int per_wave, int row_bytes,
                       int ld_bytes, long long span_bytes, int swizzle, int waves_active, void* stream) {
  const int lds = 8 * per_wave * 1024;
  if (hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -3;
  hipLaunchKernelGGL(dma_kernel, dim3(blocks), dim3(512), lds, (hipStream_t)stream, (const char*)src, cyc, sink, rounds, per_wave,
                     row_bytes, ld_bytes, span_bytes, swizzle, waves_active);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
