// Small fused elementwise / reduction kernels that surround the GEMMs (gfx950):
// activation backward passes, GELU forward, bias gradient (column sum).
// All move 16 bytes per lane (8 bf16) and are HBM-streaming kernels.
#include "dtf_common.h"

using namespace dtf;

__device__ __forceinline__ void ld8(const unsigned short* p, float (&o)[8]) {
  u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
}
__device__ __forceinline__ void st8(unsigned short* p, const float (&o)[8]) {
  u16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32_to_bf16(o[i]);
  *reinterpret_cast<u16x8*>(p) = a;
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// op 0: dx = g * (y > 0)         (ReLU backward from the OUTPUT y)
// op 1: dx = g * gelu'(x)        (GELU backward from the PRE-activation x)
// op 2: y  = gelu(x)             (a = x, b unused)
template <int OP>
__global__ __launch_bounds__(kBlock) void ew_kernel(const unsigned short* __restrict__ a,
                                                    const unsigned short* __restrict__ b,
                                                    unsigned short* __restrict__ out, long long n) {
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float x[8], y[8];
    ld8(a + i * 8, x);
    if (OP != 2) ld8(b + i * 8, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == 0) x[k] = y[k] > 0.f ? x[k] : 0.f;
      else if (OP == 1) x[k] = x[k] * gelu_df(y[k]);
      else x[k] = gelu_f(x[k]);
    }
    st8(out + i * 8, x);
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
      float x = bf16_to_f32(a[i]);
      const float y = OP != 2 ? bf16_to_f32(b[i]) : 0.f;
      if (OP == 0) x = y > 0.f ? x : 0.f;
      else if (OP == 1) x = x * gelu_df(y);
      else x = gelu_f(x);
      out[i] = f32_to_bf16(x);
    }
}

// out[n] += sum_m g[m][n] for a row-major bf16 [M][N] matrix (N % 8 == 0); out is fp32 and
// zeroed by the caller. Block = 32 column-chunks (256 columns) x 8 row lanes.
__global__ __launch_bounds__(kBlock) void col_sum_kernel(const unsigned short* __restrict__ g, float* __restrict__ out,
                                                         int M, int N, long long ld, int rows_per_block) {
  __shared__ float part[8][33][8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n = blockIdx.x * 256 + tx * 8;
  const int mbeg = blockIdx.y * rows_per_block;
  const int mend = min(M, mbeg + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N)
    for (int m = mbeg + ty; m < mend; m += 8) {
      float v[8];
      ld8(g + (long long)m * ld + n, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
#pragma unroll
  for (int k = 0; k < 8; ++k) part[ty][tx][k] = acc[k];
  __syncthreads();
  if (ty == 0 && n < N) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += part[r][tx][k];
      atomicAdd(out + n + k, s);
    }
  }
}

DTF_API int dtf_ew_bf16(int op, const void* a, const void* b, void* out, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_for((n + 7) / 8, kBlock);
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Bp = (const unsigned short*)b;
  unsigned short* O = (unsigned short*)out;
  switch (op) {
    case 0: hipLaunchKernelGGL(ew_kernel<0>, dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n); break;
    case 1: hipLaunchKernelGGL(ew_kernel<1>, dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n); break;
    case 2: hipLaunchKernelGGL(ew_kernel<2>, dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_col_sum_bf16(const void* g, float* out, int M, int N, long long ld, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  const int gx = (N + 255) / 256;
  int gy = (2048 + gx - 1) / gx;
  int rpb = (M + gy - 1) / gy;
  if (rpb < 64) rpb = 64;
  gy = (M + rpb - 1) / rpb;
  hipLaunchKernelGGL(col_sum_kernel, dim3(gx, gy), dim3(kBlock), 0, stream, (const unsigned short*)g, out, M, N, ld,
                     rpb);
  return (int)hipGetLastError();
}
