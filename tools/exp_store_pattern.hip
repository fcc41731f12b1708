// Experiment (run by tools/exp_store_pattern.py on the GPU box): HBM write (and read-modify-write) bandwidth of the GEMM
// epilogue's store pattern as a function of the CONTIGUOUS BYTES PER ROW of one store instruction.  The wide-tile epilogue
// writes 16 rows x 64 B per wave instruction (a lane owns 8 consecutive fp16 columns of one row, 4 lanes per row);
// this kernel writes the same 256 x 256 block tiles (8 waves, one block per tile) with R rows x (1024 / R) bytes per
// instruction, R = 16 (64 B: the epilogue), 8 (128 B: one full L2 line), 4, 2 (512 B: a whole row of the block tile).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int R, bool RES>
__global__ __launch_bounds__(512) void store_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ res, int M, int N) {
  constexpr int LPR = 64 / R;                // lanes per row
  constexpr int PC = LPR * 8;                // columns per instruction patch
  constexpr int WTN = PC > 64 ? PC : 64;     // wave tile width (columns)
  constexpr int WNW = 256 / WTN;             // waves along N
  constexpr int WMW = 8 / WNW;               // waves along M
  constexpr int WTM = 256 / WMW;             // wave tile height
  const int nbn = N / 256;
  const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int lr = lane / LPR, lc = (lane % LPR) * 8;
  const int row0 = bm * 256 + wm * WTM, col0 = bn * 256 + wn * WTN;
  u32x4 v = {(unsigned)lane * 0x00010001u, blockIdx.x, (unsigned)wave, 0x3c003c00u};
#pragma unroll 4
  for (int cb = 0; cb < WTN; cb += PC) {
#pragma unroll 4
    for (int rb = 0; rb < WTM; rb += R) {
      const size_t o = (size_t)(row0 + rb + lr) * N + col0 + cb + lc;
      u32x4 x = v;
      if (RES) {
        const u32x4 r = *(const u32x4*)(res + o);
        x.x += r.x; x.y ^= r.y; x.z += r.z; x.w ^= r.w;
      }
      *(u32x4*)(out + o) = x;
    }
  }
}

template <int R>
static void launch(void* out, const void* res, int M, int N, hipStream_t s) {
  const dim3 grid((M / 256) * (N / 256));
  if (res) hipLaunchKernelGGL((store_kernel<R, true>), grid, dim3(512), 0, s, (uint16_t*)out, (const uint16_t*)res, M, N);
  else hipLaunchKernelGGL((store_kernel<R, false>), grid, dim3(512), 0, s, (uint16_t*)out, (const uint16_t*)nullptr, M, N);
}

extern "C" int exp_store(void* out, const void* res, int M, int N, int rows_per_instr, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (rows_per_instr) {
    case 16: launch<16>(out, res, M, N, s); break;
    case 8: launch<8>(out, res, M, N, s); break;
    case 4: launch<4>(out, res, M, N, s); break;
    case 2: launch<2>(out, res, M, N, s); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
