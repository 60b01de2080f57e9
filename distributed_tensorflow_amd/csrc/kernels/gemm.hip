// bf16 MFMA GEMM for gfx950 with fused epilogues, three operand layouts and split-K.
//
//   C[M,N] = act(A x B + bias)      fp32 accumulate, bf16 or fp32 output
//
// Layout template flags describe how each operand lies in memory relative to
// the contraction index k:
//   A_T = 0 : A[m][k] (k contiguous)        A_T = 1 : A[k][m] (m contiguous)
//   B_T = 0 : B[n][k] (k contiguous)        B_T = 1 : B[k][n] (n contiguous)
// which covers the three products of a dense / 1x1-convolution layer:
//   forward  y  = x  . W^T   : A_T=0, B_T=0   ("NT")
//   dgrad    dx = dy . W     : A_T=0, B_T=1   ("NN")
//   wgrad    dW = dy^T . x   : A_T=1, B_T=1   ("TN")
// These are the MatMul(+BiasAdd+ReLU) ops the reference model executes inside
// TensorFlow (reference initializer.py:16-18, centralized/client.py:12,19).
//
// Structure (tiled for 64-lane wavefronts, not a warp-shaped port):
//   * 128x128 output tile per 256-thread block, BK = 64; 4 waves as 2x2, each
//     wave owns 64x64 = 4x4 MFMA tiles of v_mfma_f32_16x16x32_bf16;
//   * both operand tiles are staged global -> registers -> LDS in ONE image
//     shape, [row][k] with 128-byte rows; operands whose k is the slow index
//     are transposed in registers on the way in (4 k-rows x 8 rows per thread,
//     written as 8-byte pieces), so the MFMA fragment reads are the same
//     ds_read_b128 for every layout;
//   * the 16-byte chunk index of a row is XORed with (row >> 1) & 7, which
//     makes the 16-lane groups of ds_read_b128 hit 16 distinct 16-byte slots of
//     the 256-byte bank row (conflict-free), for writes as well;
//   * global loads for tile t+1 are issued before the MFMAs of tile t and
//     written to the other LDS buffer after them: one barrier per k-tile;
//   * MFMAs are issued with swapped operands (B fragment as the A operand), so
//     each lane ends up with 4 consecutive n of one output row; the tile is
//     then staged through LDS and written as whole 128-byte row segments;
//   * blockIdx is remapped so that the blocks that share an XCD (and its L2)
//     work on neighbouring tiles.
// Requirements checked on the host: 16-byte aligned base pointers and leading
// dimensions that are multiples of 8 elements; M, N, K themselves are free
// (edges are zero-filled / masked) except that a dimension that is the
// contiguous one of a loaded or stored row must be a multiple of 8.
#include "dtf_common.h"

using namespace dtf;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;  // 16 KiB per operand tile

struct GemmArgs {
  const unsigned short* A;
  const unsigned short* B;
  void* C;               // bf16 or fp32 [M][ldc]; fp32 [splits][M][N] workspace when splits > 1
  const void* bias;      // [N] or null
  long long lda, ldb, ldc;
  int M, N, K;
  int k_per_split;       // multiple of BK
  int splits;
  int bias_dtype;        // 0 fp32, 1 bf16
  int act;               // 0 none, 1 relu, 2 gelu(erf)
  int out_fp32;
  int tiles_m, tiles_n;
  // batched GEMM: blockIdx.z = b1 * nb2 + b2; operand offsets b1 * s?1 + b2 * s?2 (elements)
  int nb2;
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  // pixel map of a strided 1x1 convolution: logical row r = (n, oh, ow) of the OUTPUT resolution
  // addresses memory row (n*H + oh*S)*W + ow*S. map_mask bit 0: A's pixel index, bit 1: B's pixel
  // index, bit 2: C's rows (the caller zero-fills C first).
  int map_mask;
  int mOW, mOH, mW, mH, mS;
};

__device__ __forceinline__ long long map_row(const GemmArgs& g, int r) {
  const int hw = g.mOH * g.mOW;
  const int n = r / hw, rem = r - n * hw;
  const int oh = rem / g.mOW, ow = rem - oh * g.mOW;
  return ((long long)n * g.mH + (long long)oh * g.mS) * g.mW + (long long)ow * g.mS;
}

__device__ __forceinline__ int swz_off(int row, int chunk) {
  return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// ---- staging: global -> registers -------------------------------------------------
// k-contiguous operand X[row][k]: thread handles chunk ids tid + 256*i: row = id >> 3, chunk = id & 7.
__device__ __forceinline__ void stage_load_kmajor(const GemmArgs& g, bool mapped, const unsigned short* __restrict__ X,
                                                  long long ld, int rows, int row0, int k0, int kend, u32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + kThreads * i;
    const int row = row0 + (id >> 3);
    const int k = k0 + ((id & 7) << 3);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < rows && k < kend) {
      const long long mr = mapped ? map_row(g, row) : (long long)row;
      v = *reinterpret_cast<const u32x4*>(X + mr * ld + k);
    }
    r[i] = v;
  }
}

__device__ __forceinline__ void stage_store_kmajor(char* lds, const u32x4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = threadIdx.x + kThreads * i;
    *reinterpret_cast<u32x4*>(lds + swz_off(id >> 3, id & 7)) = r[i];
  }
}

// row-contiguous operand X[k][row]: thread handles row-chunk rc = tid & 15 (rows rc*8..+7) and the
// four k-rows kg*4 .. kg*4+3 with kg = tid >> 4.
__device__ __forceinline__ void stage_load_rmajor(const GemmArgs& g, bool mapped, const unsigned short* __restrict__ X,
                                                  long long ld, int rows, int row0, int k0, int kend, u32x4 (&r)[4]) {
  const int rc = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int row = row0 + (rc << 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + kg * 4 + i;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < rows && k < kend) {
      const long long mk = mapped ? map_row(g, k) : (long long)k;   // the pixel index is the contraction index here
      v = *reinterpret_cast<const u32x4*>(X + mk * ld + row);
    }
    r[i] = v;
  }
}

__device__ __forceinline__ void stage_store_rmajor(char* lds, const u32x4 (&r)[4]) {
  const int rc = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int chunk = kg >> 1, half = kg & 1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int w = j >> 1, sh = (j & 1) * 16;
    const unsigned e0 = (r[0][w] >> sh) & 0xffffu, e1 = (r[1][w] >> sh) & 0xffffu;
    const unsigned e2 = (r[2][w] >> sh) & 0xffffu, e3 = (r[3][w] >> sh) & 0xffffu;
    u32x2 o;
    o[0] = e0 | (e1 << 16);
    o[1] = e2 | (e3 << 16);
    const int row = (rc << 3) + j;
    *reinterpret_cast<u32x2*>(lds + swz_off(row, chunk) + (half << 3)) = o;
  }
}

template <int T>
__device__ __forceinline__ void stage_load(const GemmArgs& g, bool mapped, const unsigned short* X, long long ld, int rows,
                                           int row0, int k0, int kend, u32x4 (&r)[4]) {
  if (T == 0) stage_load_kmajor(g, mapped, X, ld, rows, row0, k0, kend, r);
  else stage_load_rmajor(g, mapped, X, ld, rows, row0, k0, kend, r);
}
template <int T>
__device__ __forceinline__ void stage_store(char* lds, const u32x4 (&r)[4]) {
  if (T == 0) stage_store_kmajor(lds, r);
  else stage_store_rmajor(lds, r);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

template <int A_T, int B_T>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 4 tiles: A0 B0 A1 B1 (64 KiB)

  // ---- block -> tile map: XCD-aware, bijective for any grid size --------------------
  const int nwg = g.tiles_m * g.tiles_n;
  const int orig = blockIdx.x;
  const int xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
  const int tile = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int split = blockIdx.y;
  const int b1 = blockIdx.z / g.nb2, b2 = blockIdx.z - b1 * g.nb2;
  const unsigned short* Ap = g.A + b1 * g.sA1 + b2 * g.sA2;
  const unsigned short* Bp = g.B + b1 * g.sB1 + b2 * g.sB2;
  const long long c_off = b1 * g.sC1 + b2 * g.sC2;
  const bool mapA = g.map_mask & 1, mapB = g.map_mask & 2, mapC = g.map_mask & 4;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fq = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ra[4], rb[4];
  if (nt > 0) {
    stage_load<A_T>(g, mapA, Ap, g.lda, g.M, m0, kbeg, kend, ra);
    stage_load<B_T>(g, mapB, Bp, g.ldb, g.N, n0, kbeg, kend, rb);
    stage_store<A_T>(smem, ra);
    stage_store<B_T>(smem + kTileBytes, rb);
  }
  __syncthreads();

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    const bool more = (t + 1) < nt;
    if (more) {
      const int k0 = kbeg + (t + 1) * BK;
      stage_load<A_T>(g, mapA, Ap, g.lda, g.M, m0, k0, kend, ra);
      stage_load<B_T>(g, mapB, Bp, g.ldb, g.N, n0, k0, kend, rb);
    }
    const char* la = smem + cur * 2 * kTileBytes;
    const char* lb = la + kTileBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[4], b[4];
      const int chunk = ks * 4 + fq;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + fr;
        a[i] = *reinterpret_cast<const bf16x8_t*>(la + swz_off(row, chunk));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + fr;
        b[j] = *reinterpret_cast<const bf16x8_t*>(lb + swz_off(row, chunk));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // swapped operands: result register r of lane l is C[m = i*16 + (l&15)][n = j*16 + (l>>4)*4 + r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * 2 * kTileBytes;
      stage_store<A_T>(na, ra);
      stage_store<B_T>(na + kTileBytes, rb);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue ---------------------------------------------------------------------
  const int wm0 = m0 + wm * 64, wn0 = n0 + wn * 64;
  if (g.splits > 1) {
    // fp32 partial slab [split][M][N]; 16 bytes per lane, 4 consecutive n
    float* W = reinterpret_cast<float*>(g.C) + (long long)split * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = wm0 + i * 16 + fr;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn0 + j * 16 + fq * 4;
        if (n < g.N) *reinterpret_cast<f32x4*>(W + (long long)m * g.N + n) = acc[i][j];
      }
    }
    return;
  }

  // bias + activation in registers
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = wn0 + j * 16 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N) {
      if (g.bias_dtype == 0) {
        f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.bias) + n);
        bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
      } else {
        u16x4 t = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(g.bias) + n);
        bv[0] = bf16_to_f32(t[0]); bv[1] = bf16_to_f32(t[1]); bv[2] = bf16_to_f32(t[2]); bv[3] = bf16_to_f32(t[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        if (g.act == 1) v = fmaxf(v, 0.f);
        else if (g.act == 2) v = gelu_erf(v);
        acc[i][j][r] = v;
      }
  }

  if (g.out_fp32) {
    float* C = reinterpret_cast<float*>(g.C) + c_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = wm0 + i * 16 + fr;
      if (m >= g.M) continue;
      const long long mr = mapC ? map_row(g, m) : (long long)m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn0 + j * 16 + fq * 4;
        if (n < g.N) *reinterpret_cast<f32x4*>(C + mr * g.ldc + n) = acc[i][j];
      }
    }
    return;
  }

  // bf16 output: each wave stages its 64x64 tile in its own 8 KiB of LDS ([row][64 n], 128-byte
  // rows, same XOR swizzle), then writes whole 128-byte row segments.
  char* stage = smem + wid * (64 * 128);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 o;
      o[0] = (unsigned)f32_to_bf16(acc[i][j][0]) | ((unsigned)f32_to_bf16(acc[i][j][1]) << 16);
      o[1] = (unsigned)f32_to_bf16(acc[i][j][2]) | ((unsigned)f32_to_bf16(acc[i][j][3]) << 16);
      const int cb = j * 32 + fq * 8;  // byte column inside the 128-byte row
      *reinterpret_cast<u32x2*>(stage + swz_off(row, cb >> 4) + (cb & 8)) = o;
    }
  }
  __syncthreads();
  unsigned short* C = reinterpret_cast<unsigned short*>(g.C) + c_off;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int id = lane + 64 * p;
    const int row = id >> 3, chunk = id & 7;
    const int m = wm0 + row, n = wn0 + chunk * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(stage + swz_off(row, chunk));
    if (m < g.M && n < g.N) {
      const long long mr = mapC ? map_row(g, m) : (long long)m;
      *reinterpret_cast<u32x4*>(C + mr * g.ldc + n) = v;
    }
  }
}

// Sum split-K slabs, add bias, activate, convert. One thread per 4 consecutive n.
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(const float* __restrict__ W, void* __restrict__ C,
                                                               const void* __restrict__ bias, int bias_dtype, int act,
                                                               int out_fp32, int splits, int M, int N, long long ldc) {
  const long long n4 = (long long)M * (N >> 2);
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long slab = (long long)M * N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const int m = (int)(i / (N >> 2));
    const int n = (int)(i % (N >> 2)) << 2;
    f32x4 s = *reinterpret_cast<const f32x4*>(W + (long long)m * N + n);
    for (int k = 1; k < splits; ++k) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(W + k * slab + (long long)m * N + n);
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        s[r] += bias_dtype == 0 ? reinterpret_cast<const float*>(bias)[n + r]
                                : bf16_to_f32(reinterpret_cast<const unsigned short*>(bias)[n + r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (act == 1) s[r] = fmaxf(s[r], 0.f);
      else if (act == 2) s[r] = gelu_erf(s[r]);
    }
    if (out_fp32) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + (long long)m * ldc + n) = s;
    } else {
      u16x4 o;
      o[0] = f32_to_bf16(s[0]); o[1] = f32_to_bf16(s[1]); o[2] = f32_to_bf16(s[2]); o[3] = f32_to_bf16(s[3]);
      *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(C) + (long long)m * ldc + n) = o;
    }
  }
}

// Extra, optional parameters of a launch (zero-initialise for a plain GEMM).
struct GemmExtra {
  int nb1, nb2;                      // batch = nb1 * nb2 (0 or 1 = not batched)
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  int map_mask, mOW, mOH, mW, mH, mS;
};

static int gemm_launch(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                       long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                       int splits, void* workspace, const GemmExtra* ex, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (splits < 1) splits = 1;
  GemmArgs g;
  g.A = (const unsigned short*)A; g.B = (const unsigned short*)B;
  g.bias = bias; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.bias_dtype = bias_dtype; g.act = act; g.out_fp32 = out_fp32;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  int batch = 1;
  g.nb2 = 1; g.sA1 = g.sA2 = g.sB1 = g.sB2 = g.sC1 = g.sC2 = 0;
  g.map_mask = 0; g.mOW = g.mOH = g.mW = g.mH = g.mS = 1;
  if (ex) {
    const int nb1 = ex->nb1 > 0 ? ex->nb1 : 1;
    g.nb2 = ex->nb2 > 0 ? ex->nb2 : 1;
    batch = nb1 * g.nb2;
    g.sA1 = ex->sA1; g.sA2 = ex->sA2; g.sB1 = ex->sB1; g.sB2 = ex->sB2; g.sC1 = ex->sC1; g.sC2 = ex->sC2;
    g.map_mask = ex->map_mask;
    if (g.map_mask) { g.mOW = ex->mOW; g.mOH = ex->mOH; g.mW = ex->mW; g.mH = ex->mH; g.mS = ex->mS; }
  }
  if (batch > 1 || (g.map_mask & 4)) splits = 1;   // split-K slabs are neither batched nor row-mapped
  int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  splits = (K + kps - 1) / kps;
  g.k_per_split = kps; g.splits = splits;
  g.C = splits > 1 ? workspace : C;
  if (splits > 1 && !workspace) return -3;
  const dim3 grid(g.tiles_m * g.tiles_n, splits, batch);
  const size_t lds = 4 * kTileBytes;
  switch (layout) {
    case 0: hipLaunchKernelGGL((gemm_bf16_kernel<0, 0>), grid, dim3(kThreads), lds, stream, g); break;
    case 1: hipLaunchKernelGGL((gemm_bf16_kernel<0, 1>), grid, dim3(kThreads), lds, stream, g); break;
    case 2: hipLaunchKernelGGL((gemm_bf16_kernel<1, 1>), grid, dim3(kThreads), lds, stream, g); break;
    default: return -1;
  }
  if (splits > 1) {
    const long long n4 = (long long)M * (N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n4, kBlock)), dim3(kBlock), 0, stream,
                       (const float*)workspace, C, bias, bias_dtype, act, out_fp32, splits, M, N, ldc);
  }
  return (int)hipGetLastError();
}

// layout: 0 = NT, 1 = NN, 2 = TN (see the header comment). workspace: fp32, >= splits*M*N floats when splits > 1.
DTF_API int dtf_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                          long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                          int splits, void* workspace, hipStream_t stream) {
  return gemm_launch(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, out_fp32, splits, workspace,
                     nullptr, stream);
}

// Batched and/or pixel-mapped variant; `ex` points to a GemmExtra.
DTF_API int dtf_gemm_bf16_ex(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                             long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                             int splits, void* workspace, const void* ex, hipStream_t stream) {
  return gemm_launch(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, out_fp32, splits, workspace,
                     (const GemmExtra*)ex, stream);
}
