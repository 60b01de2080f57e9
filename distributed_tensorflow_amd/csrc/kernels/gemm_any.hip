// General-purpose MFMA GEMM for gfx950: exact fp32 arithmetic, any strides, any sizes, fp32 or bf16 operands.
//
//   C[m][n] = act(sum_k A(m, k) * B(k, n) + bias[n]),   A(m, k) = A[m * sam + k * sak],  B(k, n) = B[k * sbk + n * sbn]
//
// This is the kernel behind everything the bf16 tile kernels (gemm.hip, gemm256.hip) do not take: fp32 models --
// the reference's own configuration is fp32 end to end (reference initializer.py:40, centralized/client.py:80-81) --
// and bf16 products whose rows are not 16-byte aligned, e.g. the 512 -> 10 output layer of the reference model
// (reference initializer.py:19). The matrix instruction is v_mfma_f32_16x16x4_f32: fp32 in, fp32 accumulate, bit for
// bit a k-ordered fmaf chain, at the fp32 vector rate (there is no reduced-precision fp32 matrix path on gfx950).
// bf16 operands are widened to fp32 while they are staged, which is exact.
//
// Structure: 64 x 64 output tile per 256-thread block (4 waves as 2 x 2, each 32 x 32 = 2 x 2 MFMA tiles), BK = 16.
// Both operands are staged through LDS as [k][row] images of 80-float rows: one MFMA operand read is then 16
// consecutive floats per k for 16 consecutive lanes, and the 80-float pitch puts the two k-rows a 32-lane half touches
// on disjoint halves of the 32 banks (conflict-free ds_read_b32). The global side adapts to the operand's layout at
// run time: the thread -> element map walks whichever index has unit stride, so NT, NN and TN products (and the
// im2col / transposed views the convolution front end passes) all load coalesced.
// Split-K (blockIdx.z) writes fp32 slabs that slab_reduce_kernel sums in a fixed order (deterministic).
#include "dtf_common.h"

using namespace dtf;

namespace {

constexpr int TM = 64, TN = 64, TK = 16;
constexpr int kPitch = 80;      // floats per k-row of an LDS image

// Convolution as an implicit product: one operand is not a matrix in memory but the patch matrix of an NHWC image,
// gathered element by element while it is staged (zero outside the image), so that no im2col buffer -- KH * KW times
// the activation -- is ever written or read:
//   GM 1 (forward)          A(m, q)  = x[n][oh * SH - PT + kh][ow * SW - PL + kw][c],   m = (n, oh, ow), q = (kh, kw, c)
//   GM 2 (weight gradient)  B(m, q)  = the same element with the roles swapped: the contraction runs over the pixels m
//   GM 3 (data gradient)    A(p, q') = dy[n][(ih + PT - kh) / SH][(iw + PL - kw) / SW][ko] where the divisions are exact
//                           and in range, p = (n, ih, iw), q' = (kh, kw, ko); B is the weight as [(kh, kw, ko), c]
struct ConvGather {
  int H, W, C;        // the gathered image: height, width, channels (GM 3: the OUTPUT gradient's OH, OW, K)
  int OH, OW;         // the other side's spatial size (GM 1, 2: output; GM 3: the input image's H, W)
  int KH, KW, SH, SW, PT, PL;
};

struct AnyArgs {
  const void* A;
  const void* B;
  void* C;
  const void* bias;
  long long sam, sak, sbk, sbn, ldc;
  int M, N, K;
  int k_per_split, splits;
  int bias_dtype;     // 0 fp32, 1 bf16
  int act;            // 0 none, 1 relu, 2 gelu (erf)
  ConvGather cg;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<unsigned short>(const unsigned short* p) { return bf16_to_f32(*p); }

__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(unsigned short* p, float v) { *p = f32_to_bf16(v); }

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// Stages a [TK][64] tile: element (k, r) comes from base[r * sr + k * sk]. The thread map follows the unit stride.
template <typename T>
__device__ __forceinline__ void stage(float* lds, const T* base, long long sr, long long sk, int r0, int k0, int rows,
                                      int kend) {
  const int t = threadIdx.x;
  if (sk == 1) {          // k contiguous: 16 consecutive threads walk k
    const int k = t & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (t >> 4) + 16 * i;
      const bool ok = r0 + r < rows && k0 + k < kend;
      lds[k * kPitch + r] = ok ? ldf<T>(base + (long long)(r0 + r) * sr + (k0 + k)) : 0.f;
    }
  } else {                // rows contiguous (or a general stride): 64 consecutive threads walk the rows
    const int r = t & 63;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = (t >> 6) + 4 * i;
      const bool ok = r0 + r < rows && k0 + k < kend;
      lds[k * kPitch + r] = ok ? ldf<T>(base + (long long)(r0 + r) * sr + (long long)(k0 + k) * sk) : 0.f;
    }
  }
}

// Element offset of patch entry (pixel, q) in the gathered image, or -1 for padding. DG = false: forward geometry
// (pixel indexes the output), DG = true: data-gradient geometry (pixel indexes the input, the image is dy).
template <bool DG>
__device__ __forceinline__ long long patch_offset(const ConvGather& c, int n, int a, int b, int q) {
  // (a, b): DG false: (oh * SH - PT, ow * SW - PL); DG true: (ih + PT, iw + PL) -- decoded once per row by the caller
  const int ch = q % c.C, tap = q / c.C;
  const int kh = tap / c.KW, kw = tap - kh * c.KW;
  int y, x;
  if (!DG) {
    y = a + kh; x = b + kw;
  } else {
    const int th = a - kh, tw = b - kw;
    if (th < 0 || tw < 0 || th % c.SH || tw % c.SW) return -1;
    y = th / c.SH; x = tw / c.SW;
  }
  if ((unsigned)y >= (unsigned)c.H || (unsigned)x >= (unsigned)c.W) return -1;
  return (((long long)n * c.H + y) * c.W + x) * c.C + ch;
}

// Rows of the tile are pixels, k runs over q (channels fastest in memory): 16 consecutive threads walk q.
// pix[i] = {n, a, b} of row (t >> 4) + 16 i, decoded once before the k-loop (n < 0: row past the end).
template <typename T, bool DG>
__device__ __forceinline__ void stage_patch_rows(float* lds, const T* base, const ConvGather& c, const int (&pix)[4][3],
                                                 int k0, int kend) {
  const int t = threadIdx.x, k = t & 15;
  const bool kok = k0 + k < kend;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (t >> 4) + 16 * i;
    long long off = -1;
    if (kok && pix[i][0] >= 0) off = patch_offset<DG>(c, pix[i][0], pix[i][1], pix[i][2], k0 + k);
    lds[k * kPitch + r] = off >= 0 ? ldf<T>(base + off) : 0.f;
  }
}

// Rows of the tile are patch entries q (fixed per thread, 64 consecutive threads walk them: their tap (kh, kw) and
// channel are decoded once, before the k-loop), k runs over the pixels (the same pixel for a whole wave).
template <typename T>
__device__ __forceinline__ void stage_patch_cols(float* lds, const T* base, const ConvGather& c, int kh, int kw, int ch,
                                                 bool qok, int k0, int kend) {
  const int t = threadIdx.x, r = t & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = (t >> 6) + 4 * i;
    const int m = k0 + k;
    long long off = -1;
    if (qok && m < kend) {
      const int ow = m % c.OW, r2 = m / c.OW;
      const int oh = r2 % c.OH, n = r2 / c.OH;
      const int y = oh * c.SH - c.PT + kh, x = ow * c.SW - c.PL + kw;
      if ((unsigned)y < (unsigned)c.H && (unsigned)x < (unsigned)c.W) off = (((long long)n * c.H + y) * c.W + x) * c.C + ch;
    }
    lds[k * kPitch + r] = off >= 0 ? ldf<T>(base + off) : 0.f;
  }
}

template <typename TA, typename TB, typename TC, int GM = 0>
__global__ __launch_bounds__(256) void gemm_any_kernel(AnyArgs g) {
  __shared__ float As[TK * kPitch];
  __shared__ float Bs[TK * kPitch];
  // tiles folded into grid.x (column tiles fastest): grid.y would cap the rows at 65535 tiles
  const int tiles_n = (g.N + TN - 1) / TN;
  const int tile_m = blockIdx.x / tiles_n;
  const int m0 = tile_m * TM, n0 = (blockIdx.x - tile_m * tiles_n) * TN;
  const int split = blockIdx.z;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fr = lane & 15, fk = lane >> 4;
  const TA* A = reinterpret_cast<const TA*>(g.A);
  const TB* B = reinterpret_cast<const TB*>(g.B);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // gathered operands: what does not change along k is decoded once
  int pix[4][3];
  if (GM == 1 || GM == 3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + (threadIdx.x >> 4) + 16 * i;
      if (m < g.M) {
        // (OH, OW) is the grid the ROW index runs over: the output pixels (GM 1), the input pixels (GM 3)
        const int x = m % g.cg.OW, r2 = m / g.cg.OW;
        const int y = r2 % g.cg.OH;
        pix[i][0] = r2 / g.cg.OH;
        pix[i][1] = GM == 1 ? y * g.cg.SH - g.cg.PT : y + g.cg.PT;
        pix[i][2] = GM == 1 ? x * g.cg.SW - g.cg.PL : x + g.cg.PL;
      } else {
        pix[i][0] = -1; pix[i][1] = pix[i][2] = 0;
      }
    }
  }
  const int qcol = n0 + (threadIdx.x & 63);
  int q_kh = 0, q_kw = 0, q_ch = 0;
  if (GM == 2) {
    const int tap = qcol / g.cg.C;
    q_ch = qcol - tap * g.cg.C;
    q_kh = tap / g.cg.KW;
    q_kw = tap - q_kh * g.cg.KW;
  }

  for (int k0 = kbeg; k0 < kend; k0 += TK) {
    if (GM == 1) stage_patch_rows<TA, false>(As, A, g.cg, pix, k0, kend);
    else if (GM == 3) stage_patch_rows<TA, true>(As, A, g.cg, pix, k0, kend);
    else stage<TA>(As, A, g.sam, g.sak, m0, k0, g.M, kend);
    if (GM == 2) stage_patch_cols<TB>(Bs, B, g.cg, q_kh, q_kw, q_ch, qcol < g.N, k0, kend);
    else stage<TB>(Bs, B, g.sbn, g.sbk, n0, k0, g.N, kend);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(kk + fk) * kPitch + wm * 32 + i * 16 + fr];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(kk + fk) * kPitch + wn * 32 + j * 16 + fr];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // result register r of lane l: row (l >> 4) * 4 + r, column l & 15 of the 16 x 16 tile
  if (g.splits > 1) {
    float* W = reinterpret_cast<float*>(g.C) + (long long)split * g.M * g.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + wm * 32 + i * 16 + fk * 4 + r;
          if (m < g.M && n < g.N) W[(long long)m * g.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  TC* C = reinterpret_cast<TC*>(g.C);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 32 + j * 16 + fr;
    float bv = 0.f;
    if (g.bias && n < g.N)
      bv = g.bias_dtype == 0 ? reinterpret_cast<const float*>(g.bias)[n]
                             : bf16_to_f32(reinterpret_cast<const unsigned short*>(g.bias)[n]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + i * 16 + fk * 4 + r;
        if (m < g.M && n < g.N) {
          float v = acc[i][j][r] + bv;
          if (g.act == 1) v = fmaxf(v, 0.f);
          else if (g.act == 2) v = gelu_erf_f(v);
          stf(C + (long long)m * g.ldc + n, v);
        }
      }
  }
}

// C[m][n] = act(sum_s W[s][m][n] + bias[n]): slabs are summed in slab order, one thread per output element.
template <typename TC>
__global__ __launch_bounds__(kBlock) void slab_reduce_kernel(const float* __restrict__ W, TC* __restrict__ C,
                                                             const void* __restrict__ bias, int bias_dtype, int act,
                                                             int splits, int M, int N, long long ldc) {
  const long long total = (long long)M * N;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += W[(long long)k * total + i];
    const int m = (int)(i / N), n = (int)(i - (long long)m * N);
    if (bias)
      s += bias_dtype == 0 ? reinterpret_cast<const float*>(bias)[n]
                           : bf16_to_f32(reinterpret_cast<const unsigned short*>(bias)[n]);
    if (act == 1) s = fmaxf(s, 0.f);
    else if (act == 2) s = gelu_erf_f(s);
    stf(C + (long long)m * ldc + n, s);
  }
}

template <typename TA, typename TB>
void launch_c(const AnyArgs& g, int c_dtype, dim3 grid, hipStream_t stream) {
  if (c_dtype == 0 || g.splits > 1) hipLaunchKernelGGL((gemm_any_kernel<TA, TB, float>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_any_kernel<TA, TB, unsigned short>), grid, dim3(256), 0, stream, g);
}

// gathered variants: both operands of one element type (a convolution's image, gradient and weight share it)
template <typename T, int GM>
void launch_gather(const AnyArgs& g, int c_dtype, dim3 grid, hipStream_t stream) {
  if (c_dtype == 0 || g.splits > 1)
    hipLaunchKernelGGL((gemm_any_kernel<T, T, float, GM>), grid, dim3(256), 0, stream, g);
  else
    hipLaunchKernelGGL((gemm_any_kernel<T, T, unsigned short, GM>), grid, dim3(256), 0, stream, g);
}

}  // namespace

// dtype codes: 0 = fp32, 1 = bf16. Strides in elements. workspace: fp32, >= splits * M * N floats when splits > 1.
static int gemm_any_launch(int gm, const ConvGather* cg, const void* A, int a_dtype, long long sam, long long sak,
                           const void* B, int b_dtype, long long sbk, long long sbn, void* C, int c_dtype, long long ldc,
                           const void* bias, int bias_dtype, int act, int M, int N, int K, int splits, void* workspace,
                           hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (gm != 0 && (a_dtype != b_dtype || !cg)) return -1;
  if (!A || !B || !C || K < 0 || (unsigned)a_dtype > 1u || (unsigned)b_dtype > 1u || (unsigned)c_dtype > 1u) return -1;
  if (splits < 1) splits = 1;
  int kps = ((K + splits - 1) / splits + TK - 1) / TK * TK;
  if (kps < TK) kps = TK;
  splits = K > 0 ? (K + kps - 1) / kps : 1;
  if (splits > 1 && !workspace) return -3;
  AnyArgs g;
  g.A = A; g.B = B; g.C = splits > 1 ? workspace : C; g.bias = bias;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K; g.k_per_split = kps; g.splits = splits; g.bias_dtype = bias_dtype; g.act = act;
  if (cg) g.cg = *cg; else g.cg = ConvGather{};
  const long long tiles = (long long)((N + TN - 1) / TN) * ((M + TM - 1) / TM);
  if (tiles > 0x7fffffffLL || splits > 65535) return -2;
  const dim3 grid((unsigned)tiles, 1, splits);
  if (gm != 0) {
#define LG(GM_)                                                                       \
  do {                                                                                \
    if (a_dtype == 0) launch_gather<float, GM_>(g, c_dtype, grid, stream);            \
    else launch_gather<unsigned short, GM_>(g, c_dtype, grid, stream);                \
  } while (0)
    if (gm == 1) LG(1); else if (gm == 2) LG(2); else if (gm == 3) LG(3); else return -1;
#undef LG
  } else if (a_dtype == 0 && b_dtype == 0) launch_c<float, float>(g, c_dtype, grid, stream);
  else if (a_dtype == 1 && b_dtype == 1) launch_c<unsigned short, unsigned short>(g, c_dtype, grid, stream);
  else if (a_dtype == 0) launch_c<float, unsigned short>(g, c_dtype, grid, stream);
  else launch_c<unsigned short, float>(g, c_dtype, grid, stream);
  if (splits > 1) {
    const int blocks = grid_for((long long)M * N, kBlock);
    if (c_dtype == 0)
      hipLaunchKernelGGL((slab_reduce_kernel<float>), dim3(blocks), dim3(kBlock), 0, stream, (const float*)workspace,
                         (float*)C, bias, bias_dtype, act, splits, M, N, ldc);
    else
      hipLaunchKernelGGL((slab_reduce_kernel<unsigned short>), dim3(blocks), dim3(kBlock), 0, stream,
                         (const float*)workspace, (unsigned short*)C, bias, bias_dtype, act, splits, M, N, ldc);
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_gemm_any(const void* A, int a_dtype, long long sam, long long sak, const void* B, int b_dtype,
                         long long sbk, long long sbn, void* C, int c_dtype, long long ldc, const void* bias,
                         int bias_dtype, int act, int M, int N, int K, int splits, void* workspace,
                         hipStream_t stream) {
  return gemm_any_launch(0, nullptr, A, a_dtype, sam, sak, B, b_dtype, sbk, sbn, C, c_dtype, ldc, bias, bias_dtype, act,
                         M, N, K, splits, workspace, stream);
}

// Convolution passes as implicit products (see ConvGather). `img` is the gathered NHWC tensor, `mat` the ordinary
// matrix operand with its two strides (k stride, n stride for gm 1 / 3: the B matrix; m stride, k stride for gm 2: the
// A matrix). geom = {H, W, C, OH, OW, KH, KW, SH, SW, PT, PL} as documented at ConvGather. M, N, K: the product's
// sizes (gm 1: pixels x K_out x KH*KW*C; gm 2: K_out x KH*KW*C x pixels; gm 3: input pixels x C_in x KH*KW*K_out).
DTF_API int dtf_gemm_any_conv(int gm, const void* img, const void* mat, int dtype, long long s0, long long s1,
                              const int* geom, void* C, int c_dtype, long long ldc, const void* bias, int bias_dtype,
                              int act, int M, int N, int K, int splits, void* workspace, hipStream_t stream) {
  if (!geom || gm < 1 || gm > 3 || !img || !mat) return -1;
  ConvGather cg = {geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10]};
  if (cg.C <= 0 || cg.KW <= 0 || cg.SH <= 0 || cg.SW <= 0 || cg.OW <= 0 || cg.OH <= 0) return -1;
  if (gm == 2)
    return gemm_any_launch(2, &cg, mat, dtype, s0, s1, img, dtype, 0, 0, C, c_dtype, ldc, bias, bias_dtype, act, M, N, K,
                           splits, workspace, stream);
  return gemm_any_launch(gm, &cg, img, dtype, 0, 0, mat, dtype, s0, s1, C, c_dtype, ldc, bias, bias_dtype, act, M, N, K,
                         splits, workspace, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// im2col / col2im for NHWC images of either dtype: the fp32 convolution path is im2col + the product above.
//   cols[(n, oh, ow)][(kh, kw, c)] = x[n][oh * SH - PT + kh][ow * SW - PL + kw][c]     (zero outside the image)
// col2im is the transpose as a GATHER (one thread per image element sums the taps that read it): no atomics, fixed
// order.
namespace {

struct ColGeom {
  int NB, H, W, C, OH, OW, KH, KW, SH, SW, PT, PL;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void im2col_kernel(const T* __restrict__ x, T* __restrict__ cols, ColGeom g) {
  const long long K = (long long)g.KH * g.KW * g.C;
  const long long total = (long long)g.NB * g.OH * g.OW * K;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long row = i / K;
    const int k = (int)(i - row * K);
    const int c = k % g.C, t = k / g.C;
    const int kw = t % g.KW, kh = t / g.KW;
    const int ow = (int)(row % g.OW);
    const long long r2 = row / g.OW;
    const int oh = (int)(r2 % g.OH), n = (int)(r2 / g.OH);
    const int ih = oh * g.SH - g.PT + kh, iw = ow * g.SW - g.PL + kw;
    T v = T(0);
    if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W)
      v = x[(((long long)n * g.H + ih) * g.W + iw) * g.C + c];
    cols[i] = v;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void col2im_kernel(const T* __restrict__ cols, T* __restrict__ x, ColGeom g) {
  const long long K = (long long)g.KH * g.KW * g.C;
  const long long total = (long long)g.NB * g.H * g.W * g.C;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % g.C);
    const long long p = i / g.C;
    const int iw = (int)(p % g.W);
    const long long p2 = p / g.W;
    const int ih = (int)(p2 % g.H), n = (int)(p2 / g.H);
    float s = 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      const int th = ih + g.PT - kh;
      if (th < 0 || th % g.SH) continue;
      const int oh = th / g.SH;
      if (oh >= g.OH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int tw = iw + g.PL - kw;
        if (tw < 0 || tw % g.SW) continue;
        const int ow = tw / g.SW;
        if (ow >= g.OW) continue;
        s += ldf<T>(cols + (((long long)n * g.OH + oh) * g.OW + ow) * K + ((long long)kh * g.KW + kw) * g.C + c);
      }
    }
    stf(x + i, s);
  }
}

}  // namespace

// dir 0: im2col (x -> cols), dir 1: col2im (cols -> x). dtype 0 fp32, 1 bf16.
DTF_API int dtf_im2col(int dir, int dtype, const void* src, void* dst, int NB, int H, int W, int C, int OH, int OW,
                       int KH, int KW, int SH, int SW, int PT, int PL, hipStream_t stream) {
  if (!src || !dst || NB <= 0 || SH < 1 || SW < 1) return -1;
  const ColGeom g = {NB, H, W, C, OH, OW, KH, KW, SH, SW, PT, PL};
  const long long total = dir == 0 ? (long long)NB * OH * OW * KH * KW * C : (long long)NB * H * W * C;
  const dim3 grid(grid_for(total, kBlock * 4));
  if (dir == 0) {
    if (dtype == 0) hipLaunchKernelGGL((im2col_kernel<float>), grid, dim3(kBlock), 0, stream, (const float*)src, (float*)dst, g);
    else hipLaunchKernelGGL((im2col_kernel<unsigned short>), grid, dim3(kBlock), 0, stream, (const unsigned short*)src,
                            (unsigned short*)dst, g);
  } else {
    if (dtype == 0) hipLaunchKernelGGL((col2im_kernel<float>), grid, dim3(kBlock), 0, stream, (const float*)src, (float*)dst, g);
    else hipLaunchKernelGGL((col2im_kernel<unsigned short>), grid, dim3(kBlock), 0, stream, (const unsigned short*)src,
                            (unsigned short*)dst, g);
  }
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 elementwise companions of the bf16 kernels in elementwise.hip (same op codes, same Philox mask convention:
// the block of 4 random words for linear elements 4q .. 4q+3 is philox(seed, *off + q)).
//   op 0: out = a * (b > 0)     op 1: out = a * gelu'(b)     op 2: out = gelu(a)     op 3: out = a
// thresh != 0 applies dropout (mask + scale) to `a` first; thresh == 0 with scale != 1 is a plain scaling.
namespace {

__device__ __forceinline__ float gelu_df_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

template <int OP>
__global__ __launch_bounds__(kBlock) void ew_f32_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        float* __restrict__ out, long long n, unsigned thresh,
                                                        float scale, unsigned long long seed,
                                                        const unsigned long long* __restrict__ off_p) {
  const unsigned long long off = thresh ? *off_p : 0ull;
  const long long n4 = (n + 3) >> 2;
  for (long long q = (long long)blockIdx.x * kBlock + threadIdx.x; q < n4; q += (long long)gridDim.x * kBlock) {
    unsigned rnd[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (thresh) philox4x32_10(seed, off + (unsigned long long)q, rnd);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long i = q * 4 + e;
      if (i >= n) break;
      float x = a[i];
      x = rnd[e] >= thresh ? x * scale : 0.f;
      if (OP == 0) x = b[i] > 0.f ? x : 0.f;
      else if (OP == 1) x = x * gelu_df_f(b[i]);
      else if (OP == 2) x = gelu_erf_f(x);
      out[i] = x;
    }
  }
}

}  // namespace

DTF_API int dtf_ew_f32(int op, const float* a, const float* b, float* out, long long n, float p, float scale,
                       unsigned long long seed, const void* off, hipStream_t stream) {
  if (n <= 0) return 0;
  if (p < 0.f || p >= 1.f || (p > 0.f && !off) || !a || !out || ((op == 0 || op == 1) && !b)) return -1;
  const unsigned thresh = dropout_threshold(p);
  const float sc = p > 0.f ? 1.f / (1.f - p) : scale;
  const dim3 grid(grid_stream((n + 3) / 4, kBlock));
  const unsigned long long* o = (const unsigned long long*)off;
  switch (op) {
    case 0: hipLaunchKernelGGL((ew_f32_kernel<0>), grid, dim3(kBlock), 0, stream, a, b, out, n, thresh, sc, seed, o); break;
    case 1: hipLaunchKernelGGL((ew_f32_kernel<1>), grid, dim3(kBlock), 0, stream, a, b, out, n, thresh, sc, seed, o); break;
    case 2: hipLaunchKernelGGL((ew_f32_kernel<2>), grid, dim3(kBlock), 0, stream, a, b, out, n, thresh, sc, seed, o); break;
    case 3: hipLaunchKernelGGL((ew_f32_kernel<3>), grid, dim3(kBlock), 0, stream, a, b, out, n, thresh, sc, seed, o); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
