// Normalisation and pooling for EVERY element type and channel count (gfx950): fp32 or bf16 NHWC activations, any C.
//
// norm.hip and pool.hip are tuned for the flagship's bf16 activations with a channel count that is a multiple of 8
// (16-byte chunks, packed mask bits, statistics gathered in the neighbouring convolutions). The reference, though,
// trains in fp32 end to end (reference initializer.py:40, centralized/client.py:80-81) and so does this framework's
// launcher by default; a user model may also have 3, 10 or 20 channels somewhere. Those cases run here, on plain
// one-element-per-lane kernels with fp32 arithmetic, instead of falling through to a vendor library:
//
//   batch norm   gbn_partial  -> gbn_finalize_{fwd,bwd} -> gbn_apply_{fwd,bwd}   (deterministic: per-block partial
//                rows summed in a fixed order, no float atomics)
//   pooling      gpool_fwd / gpool_bwd: max (first maximum wins, NaN propagates, winning tap kept in one byte) and
//                average over rectangular windows and strides; gap_any_{fwd,bwd}: global average
//
// Memory-bound and coalesced along the channel axis: consecutive lanes take consecutive channels of one pixel.
#include "dtf_common.h"

using namespace dtf;

namespace {

template <typename T> __device__ __forceinline__ float ldv(const T* p);
template <> __device__ __forceinline__ float ldv<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldv<unsigned short>(const unsigned short* p) { return bf16_to_f32(*p); }
__device__ __forceinline__ void stv(float* p, float v) { *p = v; }
__device__ __forceinline__ void stv(unsigned short* p, float v) { *p = f32_to_bf16(v); }

constexpr int kGbnMaxParts = 512;

// Partial sums over a row range. A block takes up to 256 channels (blockIdx.x) and, when the tensor is narrower than
// that, 256 / C rows per pass: thread t reads element (row0 + t / C) * C + t % C, so that the threads of a block walk
// consecutive addresses whatever the width (16 channels: 16 rows of 16 per pass) and every thread stays on one
// channel. The row lanes of a channel are then summed in a fixed order.
//   MODE 0: part[p][0][c] = sum x,  part[p][1][c] = sum x^2
//   MODE 1: part[p][0][c] = sum g,  part[p][1][c] = sum g * (x - mean[c]) * invstd[c],  g = dy (* (y > 0) with relu)
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void gbn_partial_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ part,
                                                             long long M, int C, int relu) {
  __shared__ float red[2][kBlock];
  const int cb = blockIdx.x * kBlock;
  const int Cb = min(C - cb, kBlock);          // channels of this block
  const int R = kBlock / Cb;                   // rows per pass
  const int t = threadIdx.x;
  const int cl = t % Cb, rl = t / Cb;
  const int c = cb + cl;
  const int P = gridDim.y, p = blockIdx.y;
  const long long per = (M + P - 1) / P;
  const long long r0 = (long long)p * per, r1 = min(M, r0 + per);
  float s0 = 0.f, s1 = 0.f;
  if (rl < R) {
    const float mu = MODE == 1 ? mean[c] : 0.f, is = MODE == 1 ? invstd[c] : 0.f;
    for (long long r = r0 + rl; r < r1; r += R) {
      const long long i = r * C + c;
      const float xv = ldv<T>(x + i);
      if (MODE == 0) {
        s0 += xv;
        s1 = fmaf(xv, xv, s1);
      } else {
        float g = ldv<T>(dy + i);
        if (relu && !(ldv<T>(y + i) > 0.f)) g = 0.f;
        s0 += g;
        s1 = fmaf(g, (xv - mu) * is, s1);
      }
    }
  }
  red[0][t] = s0;
  red[1][t] = s1;
  __syncthreads();
  if (t < Cb) {
    float a = 0.f, b = 0.f;
    for (int k = 0; k < R; ++k) { a += red[0][k * Cb + t]; b += red[1][k * Cb + t]; }
    part[((long long)p * 2 + 0) * C + c] = a;
    part[((long long)p * 2 + 1) * C + c] = b;
  }
}

// 32 channels per block; the partial rows of a channel are summed by 8 lanes with four loads in flight each and
// combined in a fixed order (dtf_common.h partials_col_sum): hundreds of rows summed by one thread is latency bound.
__global__ __launch_bounds__(kBlock) void gbn_finalize_fwd_kernel(const float* __restrict__ part, int P,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* run_mean,
                                                                  float* run_var, float* __restrict__ mean,
                                                                  float* __restrict__ invstd, float* __restrict__ scale,
                                                                  float* __restrict__ shift, long long M, int C, float eps,
                                                                  float momentum) {
  float s0 = 0.f, s1 = 0.f;
  const bool own = partials_col_sum(part, P, 2LL * C, C, s0);
  __syncthreads();
  partials_col_sum(part + C, P, 2LL * C, C, s1);
  if (!own) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  const float inv_m = 1.f / (float)M;
  const float mu = s0 * inv_m;
  const float var = fmaxf(s1 * inv_m - mu * mu, 0.f);
  const float is = rsqrtf(var + eps);
  mean[c] = mu;
  invstd[c] = is;
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  scale[c] = g * is;
  shift[c] = b - mu * g * is;
  if (run_mean) {
    const float unbiased = M > 1 ? var * (float)M / (float)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
  }
}

// dgamma, dbeta and the coefficients of dx = cA * g - cK * (x - mean) - c0.
__global__ __launch_bounds__(kBlock) void gbn_finalize_bwd_kernel(const float* __restrict__ part, int P,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ invstd,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ cA, float* __restrict__ cK,
                                                                  float* __restrict__ c0, long long M, int C) {
  float s0 = 0.f, s1 = 0.f;
  const bool own = partials_col_sum(part, P, 2LL * C, C, s0);
  __syncthreads();
  partials_col_sum(part + C, P, 2LL * C, C, s1);
  if (!own) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  dbeta[c] = s0;
  dgamma[c] = s1;
  const float inv_m = 1.f / (float)M;
  const float a = (gamma ? gamma[c] : 1.f) * invstd[c];
  cA[c] = a;
  cK[c] = a * invstd[c] * s1 * inv_m;
  c0[c] = a * s0 * inv_m;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gbn_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                               T* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, long long total, int C,
                                                               int relu) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % C);
    float v = fmaf(ldv<T>(x + i), scale[c], shift[c]);
    if (res) v += ldv<T>(res + i);
    if (relu) v = v < 0.f ? 0.f : v;             // (keeps a NaN)
    stv(y + i, v);
  }
}

// dx = cA * g - cK * (x - mean) - c0;  gres (optional) = g: the gradient of the residual input.
template <typename T>
__global__ __launch_bounds__(kBlock) void gbn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const T* __restrict__ y, T* __restrict__ dx,
                                                               T* __restrict__ gres, const float* __restrict__ mean,
                                                               const float* __restrict__ cA, const float* __restrict__ cK,
                                                               const float* __restrict__ c0, long long total, int C,
                                                               int relu) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % C);
    float g = ldv<T>(dy + i);
    if (relu && !(ldv<T>(y + i) > 0.f)) g = 0.f;
    if (gres) stv(gres + i, g);
    stv(dx + i, cA[c] * g - cK[c] * (ldv<T>(x + i) - mean[c]) - c0[c]);
  }
}

// ------------------------------------------------------------------------------------------------------- pooling
struct PoolGeom {
  int N, H, W, C, OH, OW, KH, KW, SH, SW, PH, PW;
};

// MODE 0: max (idx = winning tap kh * KW + kw, first maximum; a NaN wins), MODE 1: average over the taps inside the image
// divided by KH * KW (count_include_pad, the convention of both frameworks' default).
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void gpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           unsigned char* __restrict__ idx, PoolGeom g) {
  const long long total = (long long)g.N * g.OH * g.OW * g.C;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % g.C);
    long long p = i / g.C;
    const int ow = (int)(p % g.OW); p /= g.OW;
    const int oh = (int)(p % g.OH);
    const int n = (int)(p / g.OH);
    float best = -INFINITY, sum = 0.f;
    int tap = 0;
    bool any = false;
    for (int kh = 0; kh < g.KH; ++kh) {
      const int ih = oh * g.SH - g.PH + kh;
      if ((unsigned)ih >= (unsigned)g.H) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int iw = ow * g.SW - g.PW + kw;
        if ((unsigned)iw >= (unsigned)g.W) continue;
        const float v = ldv<T>(x + (((long long)n * g.H + ih) * g.W + iw) * g.C + c);
        if (MODE == 0) {
          if (!any || v > best || (v != v && best == best)) { best = v; tap = kh * g.KW + kw; any = true; }
        } else {
          sum += v;
        }
      }
    }
    if (MODE == 0) {
      stv(y + i, any ? best : 0.f);
      idx[i] = (unsigned char)tap;
    } else {
      stv(y + i, sum / (float)(g.KH * g.KW));
    }
  }
}

// Gather form of the transpose: one thread per INPUT element sums the windows that read it (no atomics, fixed order).
template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void gpool_bwd_kernel(const T* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                           T* __restrict__ gx, PoolGeom g) {
  const long long total = (long long)g.N * g.H * g.W * g.C;
  const float inv = 1.f / (float)(g.KH * g.KW);
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % g.C);
    long long p = i / g.C;
    const int iw = (int)(p % g.W); p /= g.W;
    const int ih = (int)(p % g.H);
    const int n = (int)(p / g.H);
    float s = 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      const int th = ih + g.PH - kh;
      if (th < 0 || th % g.SH) continue;
      const int oh = th / g.SH;
      if (oh >= g.OH) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int tw = iw + g.PW - kw;
        if (tw < 0 || tw % g.SW) continue;
        const int ow = tw / g.SW;
        if (ow >= g.OW) continue;
        const long long o = (((long long)n * g.OH + oh) * g.OW + ow) * g.C + c;
        if (MODE == 0) {
          if (idx[o] == (unsigned char)(kh * g.KW + kw)) s += ldv<T>(gy + o);
        } else {
          s += ldv<T>(gy + o) * inv;
        }
      }
    }
    stv(gx + i, s);
  }
}

// y[n][c] = mean over the HW pixels; 4 pixel-lanes x 64 channel-lanes per block.
template <typename T>
__global__ __launch_bounds__(kBlock) void gap_any_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  float s = 0.f;
  if (c < C)
    for (int p = pl; p < HW; p += 4) s += ldv<T>(x + ((long long)n * HW + p) * C + c);
  red[pl][cl] = s;
  __syncthreads();
  if (pl == 0 && c < C) stv(y + (long long)n * C + c, (red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl]) / (float)HW);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void gap_any_bwd_kernel(const T* __restrict__ gy, T* __restrict__ gx, long long total,
                                                             int HW, int C) {
  const float inv = 1.f / (float)HW;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const int c = (int)(i % C);
    const long long n = i / ((long long)HW * C);
    stv(gx + i, ldv<T>(gy + n * C + c) * inv);
  }
}

// About 8192 elements per block (32 per thread), at most kGbnMaxParts partial rows.
inline int gbn_parts(long long M, int C) {
  const long long cols = C < kBlock ? C : kBlock;
  long long p = (M * cols + 8191) / 8192;
  if (p < 1) p = 1;
  if (p > kGbnMaxParts) p = kGbnMaxParts;
  if (p > M) p = M;
  return (int)p;
}

}  // namespace

// Number of partial rows the two reductions below write for an [M, C] tensor: workspace = parts * 2 * C floats.
DTF_API int dtf_gbn_parts(long long M, int C) { return gbn_parts(M, C > 0 ? C : 1); }

// Training forward statistics of x [M, C] (dtype 0 fp32, 1 bf16): mean, invstd, scale = gamma * invstd,
// shift = beta - mean * scale, running statistics updated with `momentum` (torch convention) when given.
DTF_API int dtf_gbn_fwd_stats(const void* x, int dtype, long long M, int C, const float* gamma, const float* beta,
                              float* run_mean, float* run_var, float* mean, float* invstd, float* scale, float* shift,
                              float eps, float momentum, float* workspace, hipStream_t stream) {
  if (!x || !mean || !invstd || !scale || !shift || !workspace || M <= 0 || C <= 0 || (unsigned)dtype > 1u) return -1;
  const int P = gbn_parts(M, C);
  const dim3 grid((C + kBlock - 1) / kBlock, P);
  if (dtype == 0)
    hipLaunchKernelGGL((gbn_partial_kernel<float, 0>), grid, dim3(kBlock), 0, stream, (const float*)x, nullptr, nullptr,
                       nullptr, nullptr, workspace, M, C, 0);
  else
    hipLaunchKernelGGL((gbn_partial_kernel<unsigned short, 0>), grid, dim3(kBlock), 0, stream, (const unsigned short*)x,
                       nullptr, nullptr, nullptr, nullptr, workspace, M, C, 0);
  hipLaunchKernelGGL(gbn_finalize_fwd_kernel, dim3((C + 31) / 32), dim3(kBlock), 0, stream, workspace, P,
                     gamma, beta, run_mean, run_var, mean, invstd, scale, shift, M, C, eps, momentum);
  return (int)hipGetLastError();
}

// y = act(x * scale + shift (+ res)) over total = M * C elements.
DTF_API int dtf_gbn_apply_fwd(const void* x, const void* res, void* y, int dtype, const float* scale, const float* shift,
                              long long total, int C, int relu, hipStream_t stream) {
  if (!x || !y || !scale || !shift || total <= 0 || C <= 0 || (unsigned)dtype > 1u) return -1;
  const dim3 grid(grid_stream(total, kBlock));
  if (dtype == 0)
    hipLaunchKernelGGL((gbn_apply_fwd_kernel<float>), grid, dim3(kBlock), 0, stream, (const float*)x, (const float*)res,
                       (float*)y, scale, shift, total, C, relu);
  else
    hipLaunchKernelGGL((gbn_apply_fwd_kernel<unsigned short>), grid, dim3(kBlock), 0, stream, (const unsigned short*)x,
                       (const unsigned short*)res, (unsigned short*)y, scale, shift, total, C, relu);
  return (int)hipGetLastError();
}

// Training backward: dx, dgamma, dbeta (and gres = masked dy when asked for). y is only read with relu != 0.
// coef: 3 * C floats of scratch; workspace as for the forward statistics.
DTF_API int dtf_gbn_bwd(const void* dy, const void* x, const void* y, void* dx, void* gres, int dtype, long long M, int C,
                        const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                        float* coef, int relu, float* workspace, hipStream_t stream) {
  if (!dy || !x || !dx || !mean || !invstd || !dgamma || !dbeta || !coef || !workspace || M <= 0 || C <= 0 ||
      (unsigned)dtype > 1u || (relu && !y))
    return -1;
  const int P = gbn_parts(M, C);
  const dim3 grid((C + kBlock - 1) / kBlock, P);
  if (dtype == 0)
    hipLaunchKernelGGL((gbn_partial_kernel<float, 1>), grid, dim3(kBlock), 0, stream, (const float*)x, (const float*)dy,
                       (const float*)y, mean, invstd, workspace, M, C, relu);
  else
    hipLaunchKernelGGL((gbn_partial_kernel<unsigned short, 1>), grid, dim3(kBlock), 0, stream, (const unsigned short*)x,
                       (const unsigned short*)dy, (const unsigned short*)y, mean, invstd, workspace, M, C, relu);
  float *cA = coef, *cK = coef + C, *c0 = coef + 2 * C;
  hipLaunchKernelGGL(gbn_finalize_bwd_kernel, dim3((C + 31) / 32), dim3(kBlock), 0, stream, workspace, P,
                     gamma, invstd, dgamma, dbeta, cA, cK, c0, M, C);
  const long long total = M * C;
  const dim3 g2(grid_stream(total, kBlock));
  if (dtype == 0)
    hipLaunchKernelGGL((gbn_apply_bwd_kernel<float>), g2, dim3(kBlock), 0, stream, (const float*)dy, (const float*)x,
                       (const float*)y, (float*)dx, (float*)gres, mean, cA, cK, c0, total, C, relu);
  else
    hipLaunchKernelGGL((gbn_apply_bwd_kernel<unsigned short>), g2, dim3(kBlock), 0, stream, (const unsigned short*)dy,
                       (const unsigned short*)x, (const unsigned short*)y, (unsigned short*)dx, (unsigned short*)gres,
                       mean, cA, cK, c0, total, C, relu);
  return (int)hipGetLastError();
}

// mode 0 max (idx: one byte per output element, required), 1 average. dir 0 forward (a = x, b = y), 1 backward
// (a = gy, b = gx).
DTF_API int dtf_gpool(int dir, int mode, int dtype, const void* a, void* b, void* idx, int N, int H, int W, int C, int OH,
                      int OW, int KH, int KW, int SH, int SW, int PH, int PW, hipStream_t stream) {
  if (!a || !b || (unsigned)dtype > 1u || (unsigned)mode > 1u || (unsigned)dir > 1u || (mode == 0 && !idx)) return -1;
  if (N <= 0 || C <= 0 || KH < 1 || KW < 1 || SH < 1 || SW < 1 || KH * KW > 255) return -1;
  const PoolGeom g = {N, H, W, C, OH, OW, KH, KW, SH, SW, PH, PW};
  const long long total = dir == 0 ? (long long)N * OH * OW * C : (long long)N * H * W * C;
  if (total <= 0) return 0;
  const dim3 grid(grid_stream(total, kBlock));
  unsigned char* ix = (unsigned char*)idx;
#define GP(T, M_)                                                                                              \
  do {                                                                                                         \
    if (dir == 0) hipLaunchKernelGGL((gpool_fwd_kernel<T, M_>), grid, dim3(kBlock), 0, stream, (const T*)a, (T*)b, ix, g); \
    else hipLaunchKernelGGL((gpool_bwd_kernel<T, M_>), grid, dim3(kBlock), 0, stream, (const T*)a, ix, (T*)b, g);          \
  } while (0)
  if (dtype == 0) { if (mode == 0) GP(float, 0); else GP(float, 1); }
  else { if (mode == 0) GP(unsigned short, 0); else GP(unsigned short, 1); }
#undef GP
  return (int)hipGetLastError();
}

// Global average over HW pixels of [N, HW, C]: dir 0 y[N, C] = mean(x), dir 1 gx = gy / HW broadcast.
DTF_API int dtf_gap_any(int dir, int dtype, const void* a, void* b, int N, int HW, int C, hipStream_t stream) {
  if (!a || !b || (unsigned)dtype > 1u || N <= 0 || HW <= 0 || C <= 0 || N > 65535) return -1;
  if (dir == 0) {
    const dim3 grid((C + 63) / 64, N);
    if (dtype == 0) hipLaunchKernelGGL((gap_any_fwd_kernel<float>), grid, dim3(kBlock), 0, stream, (const float*)a, (float*)b, HW, C);
    else hipLaunchKernelGGL((gap_any_fwd_kernel<unsigned short>), grid, dim3(kBlock), 0, stream, (const unsigned short*)a,
                            (unsigned short*)b, HW, C);
  } else {
    const long long total = (long long)N * HW * C;
    const dim3 grid(grid_stream(total, kBlock));
    if (dtype == 0) hipLaunchKernelGGL((gap_any_bwd_kernel<float>), grid, dim3(kBlock), 0, stream, (const float*)a, (float*)b, total, HW, C);
    else hipLaunchKernelGGL((gap_any_bwd_kernel<unsigned short>), grid, dim3(kBlock), 0, stream, (const unsigned short*)a,
                            (unsigned short*)b, total, HW, C);
  }
  return (int)hipGetLastError();
}
