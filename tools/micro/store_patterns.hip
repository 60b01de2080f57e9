// Micro-benchmark: how fast 256 workgroups (one per CU, 8 waves) can write their 256 x 256 bf16 output tiles of a
// row-major [M, N] matrix, by the shape of one wave's 16-byte-per-lane store instruction:
//   p0: 16 rows x 64 bytes  (a wave owns 32 columns: half a 128-byte line per row)
//   p1:  8 rows x 128 bytes (a wave owns 64 columns: whole lines)
//   p2:  4 rows x 256 bytes
//   p3:  2 rows x 512 bytes (whole tile rows)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/store_patterns.hip -o tools/micro/store_patterns.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(512, 1) void store_kernel(unsigned short* C, int M, int N, int tiles_n, int rounds) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  u32x4 val = {(unsigned)tid, (unsigned)blockIdx.x, 3u, 4u};
  for (int r = 0; r < rounds; ++r) {
    const int tile = blockIdx.x + r * gridDim.x;
    const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
    if (P == 0) {
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int qj = 0; qj < 2; ++qj)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int idx = v * 64 + lane, row = idx >> 2, c16 = idx & 3;
            const int m = m0 + qi * 128 + wr * 64 + row, n = n0 + qj * 128 + wc * 32 + c16 * 8;
            *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = val;
          }
    } else if (P == 1) {
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const int idx = v * 64 + lane, row = idx >> 3, c16 = idx & 7;
          const int m = m0 + qi * 128 + wr * 64 + row, n = n0 + wc * 64 + c16 * 8;
          *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = val;
        }
    } else if (P == 2) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int idx = v * 64 + lane, row = idx >> 4, c16 = idx & 15;   // 64 rows x 128 columns per wave
        const int m = m0 + (wid >> 1) * 64 + row, n = n0 + (wid & 1) * 128 + c16 * 8;
        *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = val;
      }
    } else {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int idx = v * 64 + lane, row = idx >> 5, c16 = idx & 31;   // 32 rows x 256 columns per wave
        const int m = m0 + wid * 32 + row, n = n0 + c16 * 8;
        *reinterpret_cast<u32x4*>(C + (long long)m * N + n) = val;
      }
    }
  }
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 3072, rounds = argc > 2 ? atoi(argv[2]) : 12;
  const int nb = argc > 3 ? atoi(argv[3]) : 256;
  const int tiles_n = N / 256, tiles = 256 * rounds, M = (tiles + tiles_n - 1) / tiles_n * 256;
  unsigned short* C;
  if (hipMalloc(&C, (size_t)M * N * 2) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = 160 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipFuncSetAttribute(reinterpret_cast<const void*>(store_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rr : {1, rounds}) {
    for (int p = 0; p < 4; ++p) {
      float best = 1e30f;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        switch (p) {
          case 0: hipLaunchKernelGGL(store_kernel<0>, dim3(nb), dim3(512), lds, 0, C, M, N, tiles_n, rr); break;
          case 1: hipLaunchKernelGGL(store_kernel<1>, dim3(nb), dim3(512), lds, 0, C, M, N, tiles_n, rr); break;
          case 2: hipLaunchKernelGGL(store_kernel<2>, dim3(nb), dim3(512), lds, 0, C, M, N, tiles_n, rr); break;
          default: hipLaunchKernelGGL(store_kernel<3>, dim3(nb), dim3(512), lds, 0, C, M, N, tiles_n, rr); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)nb * rr * 131072.0;
      printf("blocks=%d N=%d rounds=%d pattern p%d: %.2f us per launch, %.2f us per tile round, %.2f TB/s\n", nb, N, rr, p, best * 1e3,
             best * 1e3 / rr, bytes / (best * 1e-3) / 1e12);
    }
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  return 0;
}
