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
  store8_bf16(p, o);
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Dropout whose mask is regenerated from the forward pass's (seed, offset): see philox4x32_10.
struct DropArgs {
  unsigned thresh;       // 0 = no dropout
  float scale;           // multiplies every kept element (1/(1-p), or any plain scale when thresh == 0)
  unsigned long long seed;
  const unsigned long long* off;
};

// op 0: dx = g * (y > 0)         (ReLU backward from the OUTPUT y)
// op 1: dx = g * gelu'(x)        (GELU backward from the PRE-activation x)
// op 2: y  = gelu(x)             (a = x, b unused)
// op 3: out = a                  (b unused; with DropArgs: dropout forward or backward on its own)
// With DROP the dropout mask and scale are applied to `a` FIRST (the gradient that arrives from
// above passes the dropout layer before it reaches the activation), then the op.
template <int OP, bool DROP>
__global__ __launch_bounds__(kBlock) void ew_kernel(const unsigned short* __restrict__ a,
                                                    const unsigned short* __restrict__ b,
                                                    unsigned short* __restrict__ out, long long n, DropArgs d) {
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * blockDim.x;
  const unsigned long long off = DROP && d.thresh ? *d.off : 0ull;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float x[8], y[8];
    ld8(a + i * 8, x);
    if (OP == 0 || OP == 1) ld8(b + i * 8, y);
    if (DROP) {
      if (d.thresh) {
        unsigned r0[4], r1[4];
        philox4x32_10(d.seed, off + (unsigned long long)(2 * i), r0);
        philox4x32_10(d.seed, off + (unsigned long long)(2 * i + 1), r1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          x[k] = r0[k] >= d.thresh ? x[k] * d.scale : 0.f;
          x[k + 4] = r1[k] >= d.thresh ? x[k + 4] * d.scale : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] *= d.scale;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (OP == 0) x[k] = y[k] > 0.f ? x[k] : 0.f;
      else if (OP == 1) x[k] = x[k] * gelu_df(y[k]);
      else if (OP == 2) x[k] = gelu_f(x[k]);
    }
    st8(out + i * 8, x);
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
      float x = bf16_to_f32(a[i]);
      const float y = (OP == 0 || OP == 1) ? bf16_to_f32(b[i]) : 0.f;
      if (DROP) {
        if (d.thresh) {
          unsigned r[4];
          philox4x32_10(d.seed, off + (unsigned long long)(i >> 2), r);
          x = r[i & 3] >= d.thresh ? x * d.scale : 0.f;
        } else {
          x *= d.scale;
        }
      }
      if (OP == 0) x = y > 0.f ? x : 0.f;
      else if (OP == 1) x = x * gelu_df(y);
      else if (OP == 2) x = gelu_f(x);
      out[i] = f32_to_bf16(x);
    }
}

// *counter += inc, on the device, so that a captured graph draws a fresh mask on every replay.
__global__ void rng_tick_kernel(unsigned long long* counter, unsigned long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *counter += inc;
}

// Column sums of a row-major bf16 [M][N] matrix (N % 8 == 0), two deterministic stages:
// stage 1: block (bx, by) sums rows [by*rpb, (by+1)*rpb) of 256 columns -> part[by][n]
//          (32 column chunks x 8 row lanes per block, combined through LDS in a fixed order);
// stage 2: out[n] = sum_by part[by][n]. With a single row block stage 1 writes `out` directly.
__global__ __launch_bounds__(kBlock) void col_sum_kernel(const unsigned short* __restrict__ g, float* __restrict__ part,
                                                         int M, int N, long long ld, int rows_per_block) {
  __shared__ float red[8][33][8];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int n = blockIdx.x * 256 + tx * 8;
  const int mbeg = blockIdx.y * rows_per_block;
  const int mend = min(M, mbeg + rows_per_block);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (n < N) {
    int m = mbeg + ty;
    for (; m + 8 < mend; m += 16) {        // two loads in flight
      float v[8], w[8];
      ld8(g + (long long)m * ld + n, v);
      ld8(g + (long long)(m + 8) * ld + n, w);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k] + w[k];
    }
    for (; m < mend; m += 8) {
      float v[8];
      ld8(g + (long long)m * ld + n, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[ty][tx][k] = acc[k];
  __syncthreads();
  if (ty == 0 && n < N) {
    float* dst = part + (long long)blockIdx.y * N + n;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r][tx][k];
      dst[k] = s;
    }
  }
}

__global__ __launch_bounds__(kBlock) void col_sum_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                int n_part, int N) {
  float s;
  if (!partials_col_sum(part, n_part, N, N, s)) return;
  out[blockIdx.x * 32 + (threadIdx.x & 31)] = s;
}

DTF_API int dtf_ew_bf16(int op, const void* a, const void* b, void* out, long long n, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_stream((n + 7) / 8, kBlock);
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Bp = (const unsigned short*)b;
  unsigned short* O = (unsigned short*)out;
  const DropArgs d = {0u, 1.f, 0ull, nullptr};
  switch (op) {
    case 0: hipLaunchKernelGGL((ew_kernel<0, false>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    case 1: hipLaunchKernelGGL((ew_kernel<1, false>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    case 2: hipLaunchKernelGGL((ew_kernel<2, false>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

// The same ops with dropout (probability p, mask from philox(seed, *off + idx/4)) applied to `a`
// first; op 3 is dropout alone. p == 0 with scale != 1 is a plain scaling.
DTF_API int dtf_ew_drop_bf16(int op, const void* a, const void* b, void* out, long long n, float p, float scale,
                             unsigned long long seed, const void* off, hipStream_t stream) {
  if (n <= 0) return 0;
  if (p < 0.f || p >= 1.f || (p > 0.f && !off)) return -1;
  const int grid = grid_stream((n + 7) / 8, kBlock);
  const unsigned short* A = (const unsigned short*)a;
  const unsigned short* Bp = (const unsigned short*)b;
  unsigned short* O = (unsigned short*)out;
  const DropArgs d = {dropout_threshold(p), p > 0.f ? 1.f / (1.f - p) : scale, seed, (const unsigned long long*)off};
  switch (op) {
    case 0: hipLaunchKernelGGL((ew_kernel<0, true>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    case 1: hipLaunchKernelGGL((ew_kernel<1, true>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    case 3: hipLaunchKernelGGL((ew_kernel<3, true>), dim3(grid), dim3(kBlock), 0, stream, A, Bp, O, n, d); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_rng_tick(void* counter, unsigned long long inc, hipStream_t stream) {
  if (!counter) return -1;
  hipLaunchKernelGGL(rng_tick_kernel, dim3(1), dim3(64), 0, stream, (unsigned long long*)counter, inc);
  return (int)hipGetLastError();
}

// Number of row blocks (= fp32 rows of workspace needed) for an [M][N] column sum.
DTF_API int dtf_col_sum_parts(int M, int N) {
  const int gx = (N + 255) / 256;
  int gy = (1024 + gx - 1) / gx;
  const int by_rows = (M + 31) / 32;
  if (gy > by_rows) gy = by_rows;
  if (gy > 256) gy = 256;
  if (gy < 1) gy = 1;
  const int rpb = (M + gy - 1) / gy;
  return (M + rpb - 1) / rpb;
}

// out: fp32 [N]; workspace: fp32 [dtf_col_sum_parts(M, N)][N] (unused when that is 1).
DTF_API int dtf_col_sum_bf16(const void* g, float* out, int M, int N, long long ld, float* workspace,
                             hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (N % 8) return -1;
  const int gx = (N + 255) / 256;
  const int gy = dtf_col_sum_parts(M, N);
  const int rpb = (M + gy - 1) / gy;
  if (gy > 1 && !workspace) return -2;
  hipLaunchKernelGGL(col_sum_kernel, dim3(gx, gy), dim3(kBlock), 0, stream, (const unsigned short*)g,
                     gy > 1 ? workspace : out, M, N, ld, rpb);
  if (gy > 1)
    hipLaunchKernelGGL(col_sum_finish_kernel, dim3((N + 31) / 32), dim3(kBlock), 0, stream, workspace, out, gy, N);
  return (int)hipGetLastError();
}

// out[n] = sum over n_part rows of part ([n_part][N] fp32): the finishing step of column sums whose partial rows another
// kernel wrote (the 256x256 GEMM's store phase)
// Many partial rows (thousands from the attention kernels) are first folded in place: block (x, y) sums chunk y of the
// rows for its 32 columns into the chunk's first row, so that the work is spread over N / 32 x n_chunks blocks instead
// of N / 32 (72 blocks for 2304 columns: 40 us for 6,144 rows); the finish then adds one row per chunk.
__global__ __launch_bounds__(kBlock) void col_sum_fold_kernel(float* __restrict__ part, int n_part, int N, int chunk) {
  const int r0 = blockIdx.y * chunk;
  float s;
  if (!partials_col_sum(part + (long long)r0 * N, min(chunk, n_part - r0), N, N, s)) return;
  part[(long long)r0 * N + blockIdx.x * 32 + (threadIdx.x & 31)] = s;
}

__global__ __launch_bounds__(kBlock) void col_sum_finish_strided_kernel(const float* __restrict__ part,
                                                                        float* __restrict__ out, int n_part,
                                                                        long long stride, int N) {
  float s;
  if (!partials_col_sum(part, n_part, stride, N, s)) return;
  out[blockIdx.x * 32 + (threadIdx.x & 31)] = s;
}

// NOTE: with more than 512 partial rows `part` is used as scratch (its chunk-leading rows are overwritten).
DTF_API int dtf_col_sum_finish(float* part, float* out, int n_part, int N, hipStream_t stream) {
  if (n_part <= 0 || N <= 0) return 0;
  if (n_part > 512) {
    const int chunk = 128, ny = (n_part + chunk - 1) / chunk;
    hipLaunchKernelGGL(col_sum_fold_kernel, dim3((N + 31) / 32, ny), dim3(kBlock), 0, stream, part, n_part, N, chunk);
    hipLaunchKernelGGL(col_sum_finish_strided_kernel, dim3((N + 31) / 32), dim3(kBlock), 0, stream, part, out, ny,
                       (long long)chunk * N, N);
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(col_sum_finish_kernel, dim3((N + 31) / 32), dim3(kBlock), 0, stream, part, out, n_part, N);
  return (int)hipGetLastError();
}

// Channel padding of an NHWC bf16 image: out[p][0..Cp) = in[p][0..C) followed by zeros (Cp = 4 or a multiple of 8,
// C < Cp). A first-layer convolution (1, 3 or 4 input channels) is fed through this so that every pixel (Cp = 8) or
// every pixel pair (Cp = 4) is one aligned 16-byte operand chunk of the implicit-GEMM kernel. A thread writes one
// 16-byte chunk; its (at most 8) source elements are neighbours in memory, so the scalar loads coalesce.
__global__ __launch_bounds__(kBlock) void pad_channels_kernel(const unsigned short* __restrict__ in,
                                                              unsigned short* __restrict__ out, long long n_chunks,
                                                              int C, int Cp) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n_chunks; i += (long long)gridDim.x * kBlock) {
    u16x8 v;
    if (Cp == 4) {                     // two pixels per chunk
      const unsigned short* s = in + i * 2 * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int px = e >> 2, c = e & 3;
        v[e] = c < C ? s[px * C + c] : (unsigned short)0;
      }
    } else {                           // Cp / 8 chunks per pixel
      const int per = Cp >> 3;
      const long long px = i / per;
      const int c0 = (int)(i - px * per) << 3;
      const unsigned short* s = in + px * C;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? s[c0 + e] : (unsigned short)0;
    }
    *reinterpret_cast<u16x8*>(out + i * 8) = v;
  }
}

// in: bf16 [n_pix][C]; out: bf16 [n_pix][Cp], 16-byte aligned. Cp = 4 needs an even pixel count.
DTF_API int dtf_pad_channels_bf16(const void* in, void* out, long long n_pix, int C, int Cp, hipStream_t stream) {
  if (n_pix <= 0) return 0;
  if (C < 1 || C >= Cp || !(Cp == 4 || Cp % 8 == 0) || (Cp == 4 && (n_pix & 1))) return -1;
  const long long n_chunks = n_pix * Cp / 8;
  hipLaunchKernelGGL(pad_channels_kernel, dim3(grid_stream(n_chunks, kBlock)), dim3(kBlock), 0, stream,
                     (const unsigned short*)in, (unsigned short*)out, n_chunks, C, Cp);
  return (int)hipGetLastError();
}

// Streaming-copy variants for bandwidth measurements (tools/bench_stream.py): what a memory-bound kernel of this
// library can expect from HBM, by access pattern. Not used by the training path.
//   variant 0-3: grid-stride loop over `blocks` blocks; 0 = one 16-byte load and store per iteration, 1 = four loads
//                in flight, 2 = as 1 with non-temporal stores, 3 = as 1 with non-temporal loads and stores
//   variant 100 + 10*log2(chunks) + 2*nt_loads + nt_stores: one pass, every thread moves `chunks` 16-byte pieces a
//                block-width apart and the grid covers the buffer (no loop)
template <int VARIANT>
__global__ __launch_bounds__(kBlock) void stream_copy_kernel(const u16x8* __restrict__ src, u16x8* __restrict__ dst,
                                                             long long n16) {
  if (VARIANT == 0) {
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n16; i += (long long)gridDim.x * kBlock)
      dst[i] = src[i];
  } else {
    const long long stride = (long long)gridDim.x * kBlock;
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
      u16x8 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = VARIANT == 3 ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (VARIANT >= 2) __builtin_nontemporal_store(v[k], dst + i + k * stride);
        else dst[i + k * stride] = v[k];
      }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
  }
}

template <int CHUNKS, bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void stream_onepass_kernel(const u16x8* __restrict__ src, u16x8* __restrict__ dst,
                                                                long long n16) {
  const long long base = (long long)blockIdx.x * kBlock * CHUNKS + threadIdx.x;
  u16x8 v[CHUNKS];
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k)
    if (base + k * kBlock < n16) v[k] = NTL ? __builtin_nontemporal_load(src + base + k * kBlock) : src[base + k * kBlock];
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k)
    if (base + k * kBlock < n16) {
      if (NTS) __builtin_nontemporal_store(v[k], dst + base + k * kBlock);
      else dst[base + k * kBlock] = v[k];
    }
}

// One pass with the block index remapped so that each of the 8 XCDs (blocks are dealt to them round-robin) walks its
// own contiguous eighth of the buffer: the mapping the tiled GEMMs use for L2 locality, here to see what eight
// separate fronts instead of one cost in DRAM efficiency.
__global__ __launch_bounds__(kBlock) void stream_xcd_kernel(const u16x8* __restrict__ src, u16x8* __restrict__ dst,
                                                            long long n16, int per_xcd) {
  const long long blk = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const long long i = blk * kBlock + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

template <int CHUNKS>
static void onepass_launch(const u16x8* s, u16x8* d, long long n16, int nt, hipStream_t stream) {
  const unsigned nb = (unsigned)((n16 + kBlock * CHUNKS - 1) / (kBlock * CHUNKS));
  switch (nt) {
    case 0: hipLaunchKernelGGL((stream_onepass_kernel<CHUNKS, false, false>), dim3(nb), dim3(kBlock), 0, stream, s, d, n16); break;
    case 1: hipLaunchKernelGGL((stream_onepass_kernel<CHUNKS, false, true>), dim3(nb), dim3(kBlock), 0, stream, s, d, n16); break;
    case 2: hipLaunchKernelGGL((stream_onepass_kernel<CHUNKS, true, false>), dim3(nb), dim3(kBlock), 0, stream, s, d, n16); break;
    default: hipLaunchKernelGGL((stream_onepass_kernel<CHUNKS, true, true>), dim3(nb), dim3(kBlock), 0, stream, s, d, n16); break;
  }
}

DTF_API int dtf_stream_copy(const void* src, void* dst, long long bytes, int variant, int blocks, hipStream_t stream) {
  const long long n16 = bytes / 16;
  if (n16 <= 0) return 0;
  const u16x8* s = (const u16x8*)src;
  u16x8* d = (u16x8*)dst;
  if (blocks < 1) blocks = kMaxBlocks;
  if (variant == 200) {
    const long long nb = (n16 + kBlock - 1) / kBlock;
    const int per_xcd = (int)((nb + 7) / 8);
    hipLaunchKernelGGL(stream_xcd_kernel, dim3(per_xcd * 8), dim3(kBlock), 0, stream, s, d, n16, per_xcd);
    return (int)hipGetLastError();
  }
  if (variant >= 100) {
    const int lg = (variant - 100) / 10, nt = (variant - 100) % 10;
    if (nt > 3) return -1;
    switch (lg) {
      case 0: onepass_launch<1>(s, d, n16, nt, stream); break;
      case 1: onepass_launch<2>(s, d, n16, nt, stream); break;
      case 2: onepass_launch<4>(s, d, n16, nt, stream); break;
      case 3: onepass_launch<8>(s, d, n16, nt, stream); break;
      default: return -1;
    }
    return (int)hipGetLastError();
  }
  switch (variant) {
    case 0: hipLaunchKernelGGL((stream_copy_kernel<0>), dim3(blocks), dim3(kBlock), 0, stream, s, d, n16); break;
    case 1: hipLaunchKernelGGL((stream_copy_kernel<1>), dim3(blocks), dim3(kBlock), 0, stream, s, d, n16); break;
    case 2: hipLaunchKernelGGL((stream_copy_kernel<2>), dim3(blocks), dim3(kBlock), 0, stream, s, d, n16); break;
    case 3: hipLaunchKernelGGL((stream_copy_kernel<3>), dim3(blocks), dim3(kBlock), 0, stream, s, d, n16); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

// Caps the grid of the streaming kernels (A/B measurements: 2048 restores the looping grid).
DTF_API int dtf_set_stream_blocks(int blocks) {
  if (blocks < 1) return -1;
  stream_block_cap() = blocks;
  return 0;
}
