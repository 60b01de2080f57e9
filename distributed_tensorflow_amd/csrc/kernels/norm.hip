// Batch normalisation for NHWC bf16 activations, fused with the residual add and ReLU (gfx950).
//
// A convolution output x is viewed as a row-major [M = N*H*W, C] matrix. Training forward is
// two memory passes instead of the separate normalise / add / relu kernels of an unfused graph:
//   pass 1  bn_partial_kernel<0>  : per-block partial sum(x), sum(x^2) per channel
//           bn_finalize_fwd       : mean, 1/std, running-stat update, scale/shift vectors
//   pass 2  bn_apply_fwd_kernel   : y = relu(x * scale + shift + residual)
// and backward likewise:
//   pass 1  bn_partial_kernel<1>  : partial sum(dy'), sum(dy' * xhat),  dy' = dy * (y > 0)
//           bn_finalize_bwd       : dgamma, dbeta and the three coefficient vectors of dx
//   pass 2  bn_apply_bwd_kernel   : dx = A*dy' + K*x + C0   (and d_residual = dy')
// Every thread owns 8 consecutive channels (one 16-byte load) of a row; C/8 threads span a row
// and the remaining threads of the 256-thread block take further rows. Partial sums are
// reduced across the block's row lanes through LDS, written per block, and summed in fixed
// order by the finalize kernel, so results are bitwise reproducible (no float atomics).
// These are the BatchNorm + ReLU ops of the ResNet configurations; the reference model has
// no normalisation layer (reference initializer.py:12-21).
#include "dtf_common.h"
#include <algorithm>

using namespace dtf;

constexpr int kBnMaxPartials = 1024;

__device__ __forceinline__ void ld8bf(const unsigned short* p, float (&o)[8]) {
  u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
}
__device__ __forceinline__ void st8bf(unsigned short* p, const float (&o)[8]) {
  store8_bf16(p, o);
}
__device__ __forceinline__ void ld8f(const float* p, float (&o)[8]) {
  f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}

// MODE 0: s0 = sum x, s1 = sum x^2
// MODE 1: s0 = sum dy', s1 = sum dy' * (x - mean) * invstd,  dy' = relu ? dy * (y > 0) : dy
// grid = (n_partials, channel_chunks); a channel chunk is up to 2048 channels (256 threads x 8).
// RL (MODE 1): the ReLU form, a template parameter so that each form keeps only its own operands in registers.
template <int MODE, int RL = 0>
__global__ __launch_bounds__(kBlock) void bn_partial_kernel(const unsigned short* __restrict__ x,
                                                            const unsigned short* __restrict__ dy,
                                                            const unsigned short* __restrict__ y,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ part,
                                                            long long M, int C) {
  constexpr int relu = RL;
  __shared__ float red[2][kBlock][8];
  const int cchunk = min(C, 2048);
  const int tpr = cchunk >> 3;                 // threads per row
  const int rpb = kBlock / tpr;                // rows per block pass
  const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
  const int c0 = blockIdx.y * 2048 + tc * 8;
  const bool live = tr < rpb && c0 < C;
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mu[8], is[8], sa[RL == 2 ? 8 : 1], sb[RL == 2 ? 8 : 1];
  if (MODE == 1 && live) {
    ld8f(mean + c0, mu); ld8f(invstd + c0, is);
    if constexpr (RL == 2) { ld8f(scale + c0, sa); ld8f(shift + c0, sb); }
  }
  if (live) {
    const long long rstride = (long long)gridDim.x * rpb;
    for (long long r = (long long)blockIdx.x * rpb + tr; r < M; r += rstride) {
      float v[8];
      ld8bf(x + r * C + c0, v);
      if (MODE == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] += v[k]; s1[k] = fmaf(v[k], v[k], s1[k]); }
      } else {
        float g[8];
        ld8bf(dy + r * C + c0, g);
        if constexpr (RL == 1) {
          float o[8];
          ld8bf(y + r * C + c0, o);
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] : 0.f;
        } else if constexpr (RL == 3) {
          // one mask byte per 8 channels, written by the forward pass (y holds the byte array)
          const unsigned mb = reinterpret_cast<const unsigned char*>(y)[(r * C + c0) >> 3];
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = (mb >> k) & 1u ? g[k] : 0.f;
        } else if constexpr (RL == 2) {
          // no residual: the forward output was relu(fma(x, scale, shift)); recompute its sign from x
#pragma unroll
          for (int k = 0; k < 8; ++k) g[k] = fmaf(v[k], sa[k], sb[k]) > 0.f ? g[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] += g[k]; s1[k] = fmaf(g[k], (v[k] - mu[k]) * is[k], s1[k]); }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; }
  __syncthreads();
  if (tr == 0 && c0 < C) {
    for (int r = 1; r < rpb; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0[k] += red[0][r * tpr + tc][k]; s1[k] += red[1][r * tpr + tc][k]; }
    }
    float* p0 = part + ((long long)blockIdx.x * 2) * C + c0;
    float* p1 = p0 + C;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p0[k] = s0[k]; p1[k] = s1[k]; }
  }
}

// Sum of the per-block partials for kFinCh channels per 256-thread block: lane (c = tid % kFinCh) x
// partial-lane (tid / kFinCh) strides over the partials with 4 loads in flight, then the
// partial-lanes are combined through LDS in a fixed order (bitwise reproducible). Few channels per
// block means many blocks: the kernel is latency-bound, so it wants the whole chip.
constexpr int kFinCh = 8;
constexpr int kFinLanes = kBlock / kFinCh;   // 32

__device__ __forceinline__ bool bn_sum_partials(const float* __restrict__ part, int n_part, int C, float& s0, float& s1) {
  __shared__ float red[2][kFinLanes][kFinCh];
  const int cl = threadIdx.x % kFinCh, pl = threadIdx.x / kFinCh;
  const int c = blockIdx.x * kFinCh + cl;
  float a0 = 0.f, a1 = 0.f;
  if (c < C) {
    int p = pl;
    const long long st = (long long)2 * kFinLanes * C;   // kFinLanes partials ahead
    for (; p + 7 * kFinLanes < n_part; p += 8 * kFinLanes) {      // 16 loads in flight (1024 partials: one round)
      const float* q = part + (long long)p * 2 * C + c;
      float x[8], y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { x[i] = q[i * st]; y[i] = q[i * st + C]; }
      a0 += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
      a1 += ((y[0] + y[1]) + (y[2] + y[3])) + ((y[4] + y[5]) + (y[6] + y[7]));
    }
    for (; p + 3 * kFinLanes < n_part; p += 4 * kFinLanes) {
      const float* q = part + (long long)p * 2 * C + c;
      float x0 = q[0], y0 = q[C], x1 = q[st], y1 = q[st + C], x2 = q[2 * st], y2 = q[2 * st + C], x3 = q[3 * st],
            y3 = q[3 * st + C];
      a0 += (x0 + x1) + (x2 + x3);
      a1 += (y0 + y1) + (y2 + y3);
    }
    for (; p < n_part; p += kFinLanes) {
      a0 += part[(long long)p * 2 * C + c];
      a1 += part[(long long)p * 2 * C + C + c];
    }
  }
  red[0][pl][cl] = a0; red[1][pl][cl] = a1;
  __syncthreads();
  if (pl != 0 || c >= C) return false;
  s0 = 0.f; s1 = 0.f;
#pragma unroll
  for (int r = 0; r < kFinLanes; ++r) { s0 += red[0][r][cl]; s1 += red[1][r][cl]; }
  return true;
}

// Writes mean, invstd, scale = gamma*invstd, shift = beta - mean*scale, and folds the batch
// statistics into the running ones (unbiased variance, torch convention). grid = ceil(C / kFinCh).
__global__ __launch_bounds__(kBlock) void bn_finalize_fwd_kernel(const float* __restrict__ part, int n_part,
                                       const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ run_mean,
                                       float* __restrict__ run_var, float* __restrict__ mean,
                                       float* __restrict__ invstd, float* __restrict__ scale,
                                       float* __restrict__ shift, long long M, int C, float eps, float momentum) {
  float s0, s1;
  if (!bn_sum_partials(part, n_part, C, s0, s1)) return;
  const int c = blockIdx.x * kFinCh + (threadIdx.x % kFinCh);
  const float inv_m = 1.f / (float)M;
  const float mu = s0 * inv_m;
  const float var = fmaxf(s1 * inv_m - mu * mu, 0.f);
  const float is = rsqrtf(var + eps);
  mean[c] = mu; invstd[c] = is;
  const float a = gamma[c] * is;
  scale[c] = a; shift[c] = beta[c] - mu * a;
  if (run_mean) {
    const float unbiased = M > 1 ? var * (float)M / (float)(M - 1) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(kBlock) void bn_finalize_bwd_kernel(const float* __restrict__ part, int n_part,
                                       const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ cA,
                                       float* __restrict__ cK, float* __restrict__ c0, long long M, int C,
                                       int s1_centered_only) {
  float s0, s1;
  if (!bn_sum_partials(part, n_part, C, s0, s1)) return;
  const int c = blockIdx.x * kFinCh + (threadIdx.x % kFinCh);
  if (s1_centered_only) s1 *= invstd[c];     // the partials hold sum dy' * (x - mean): the producer had no invstd
  dbeta[c] = s0; dgamma[c] = s1;
  const float inv_m = 1.f / (float)M;
  const float a = gamma[c] * invstd[c];
  const float k = a * s1 * inv_m * invstd[c];     // dx = a*dy' - a*dbeta/M - (x - mean)*k
  cA[c] = a; cK[c] = -k; c0[c] = mean[c] * k - a * s0 * inv_m;
}

// The apply kernels: a block covers kBlock * R consecutive 16-byte items in one pass (no block-level loop at the
// default grid, see grid_stream) and a thread takes R of them a block-width apart, all loads issued before the
// first use. When C/8 divides the block width every item of a thread lies in the same 8 channels, so the
// per-channel coefficient vectors (2-3 x 32 bytes against 16 bytes of activation: most of the kernel's cache traffic)
// are fetched once per thread instead of once per item.
__device__ __forceinline__ void cvt8(const u16x8& a, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
}
template <bool NT>
__device__ __forceinline__ u16x8 ldraw(const unsigned short* p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p)) : *reinterpret_cast<const u16x8*>(p);
}

// y = act(x*scale + shift (+ res)). RN: the residual is itself a batch norm whose apply pass never ran (the norm of a
// down-sampling branch): res holds that norm's INPUT and rscale / rshift its coefficients, so the branch's normalised
// tensor is neither written nor read back (4 bytes per element of the widest activations of a stage).
template <int R, bool NT, bool RN = false>
__global__ __launch_bounds__(kBlock) void bn_apply_fwd_kernel(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ res,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              unsigned short* __restrict__ y,
                                                              unsigned char* __restrict__ mask, long long M, int C,
                                                              int relu, const float* __restrict__ rscale = nullptr,
                                                              const float* __restrict__ rshift = nullptr) {
  const int c8 = C >> 3;
  const long long n8 = M * c8;
  const bool fixed = (kBlock % c8) == 0;
  for (long long base = (long long)blockIdx.x * (kBlock * R); base < n8; base += (long long)gridDim.x * (kBlock * R)) {
    const long long i0 = base + threadIdx.x;
    u16x8 xr[R], rr[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i < n8) {
        xr[k] = ldraw<NT>(x + i * 8);
        if (res) rr[k] = ldraw<NT>(res + i * 8);
      }
    }
    float a[8], b[8], ra[RN ? 8 : 1];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i >= n8) break;
      if (k == 0 || !fixed) {
        const int c = (int)(i % c8) << 3;
        ld8f(scale + c, a); ld8f(shift + c, b);
        if constexpr (RN) {
          // the two shifts add up front: one vector less to keep
          float rb[8];
          ld8f(rscale + c, ra); ld8f(rshift + c, rb);
#pragma unroll
          for (int e = 0; e < 8; ++e) b[e] += rb[e];
        }
      }
      float v[8];
      cvt8(xr[k], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], a[e], b[e]);
      if (RN || res) {
        float r[8];
        cvt8(rr[k], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = RN ? fmaf(r[e], ra[RN ? e : 0], v[e]) : v[e] + r[e];
      }
      if (mask) {      // sign bits of the pre-activation: the backward pass reads 1 byte instead of 16
        unsigned mb = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) mb |= (v[e] > 0.f ? 1u : 0u) << e;
        mask[i] = (unsigned char)mb;
      }
      if (relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      st8bf(y + i * 8, v);
    }
  }
}

// dx = A*dy' + K*x + C0 ; dres = dy' (only written when dres != null). The ReLU form (RL = the `relu` argument of
// dtf_bn_bwd) is a template parameter: each form then keeps only its own operands in registers (the recomputed mask of
// form 2 needs 16 for scale / shift that the mask-bit form 3 does not), which decides how many waves fit a SIMD.
template <int R, bool NT, int RL>
__global__ __launch_bounds__(kBlock) void bn_apply_bwd_kernel(const unsigned short* __restrict__ x,
                                                              const unsigned short* __restrict__ dy,
                                                              const unsigned short* __restrict__ y,
                                                              const float* __restrict__ cA, const float* __restrict__ cK,
                                                              const float* __restrict__ c0,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              unsigned short* __restrict__ dx,
                                                              unsigned short* __restrict__ dres, long long M, int C) {
  constexpr int relu = RL;
  const int c8 = C >> 3;
  const long long n8 = M * c8;
  const bool fixed = (kBlock % c8) == 0;
  for (long long base = (long long)blockIdx.x * (kBlock * R); base < n8; base += (long long)gridDim.x * (kBlock * R)) {
    const long long i0 = base + threadIdx.x;
    u16x8 xr[R], gr[R], yr[RL == 1 ? R : 1];
    unsigned mbits[RL == 3 ? R : 1];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i < n8) {
        xr[k] = ldraw<NT>(x + i * 8);
        gr[k] = ldraw<NT>(dy + i * 8);
        if (relu == 1) yr[RL == 1 ? k : 0] = ldraw<NT>(y + i * 8);
        else if (relu == 3) mbits[RL == 3 ? k : 0] = reinterpret_cast<const unsigned char*>(y)[i];
      }
    }
    float a[8], kk[8], z[8], sa[RL == 2 ? 8 : 1], sb[RL == 2 ? 8 : 1];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i >= n8) break;
      if (k == 0 || !fixed) {
        const int c = (int)(i % c8) << 3;
        ld8f(cA + c, a); ld8f(cK + c, kk); ld8f(c0 + c, z);
        if constexpr (RL == 2) { ld8f(scale + c, sa); ld8f(shift + c, sb); }
      }
      float v[8], g[8];
      cvt8(xr[k], v); cvt8(gr[k], g);
      if constexpr (RL == 1) {
        float o[8];
        cvt8(yr[k], o);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
      } else if constexpr (RL == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = (mbits[k] >> e) & 1u ? g[e] : 0.f;
      } else if constexpr (RL == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = fmaf(v[e], sa[e], sb[e]) > 0.f ? g[e] : 0.f;
      }
      if (dres) st8bf(dres + i * 8, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaf(a[e], g[e], fmaf(kk[e], v[e], z[e]));
      st8bf(dx + i * 8, v);
    }
  }
}

// Two norms that share their incoming gradient -- the closing norm of a down-sampling block (input x) and the norm of
// its down-sampling branch (input z), whose outputs were added before the ReLU: dx = A dy' + K x + C0 and
// dz = A2 dy' + K2 z + C02 with dy' = dy masked by the block's mask bits. dy and the bits are read once for both.
template <int R, bool NT>
__global__ __launch_bounds__(kBlock) void bn_apply_bwd2_kernel(const unsigned short* __restrict__ x,
                                                               const unsigned short* __restrict__ z,
                                                               const unsigned short* __restrict__ dy,
                                                               const unsigned char* __restrict__ mask,
                                                               const float* __restrict__ coef,
                                                               const float* __restrict__ coef2,
                                                               unsigned short* __restrict__ dx,
                                                               unsigned short* __restrict__ dz, long long M, int C) {
  const int c8 = C >> 3;
  const long long n8 = M * c8;
  const bool fixed = (kBlock % c8) == 0;
  for (long long base = (long long)blockIdx.x * (kBlock * R); base < n8; base += (long long)gridDim.x * (kBlock * R)) {
    const long long i0 = base + threadIdx.x;
    u16x8 xr[R], zr[R], gr[R];
    unsigned mbits[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i < n8) {
        xr[k] = ldraw<NT>(x + i * 8);
        zr[k] = ldraw<NT>(z + i * 8);
        gr[k] = ldraw<NT>(dy + i * 8);
        mbits[k] = mask[i];
      }
    }
    float a[8], kk[8], c0[8], a2[8], kk2[8], c02[8];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * kBlock;
      if (i >= n8) break;
      if (k == 0 || !fixed) {
        const int c = (int)(i % c8) << 3;
        ld8f(coef + c, a); ld8f(coef + C + c, kk); ld8f(coef + 2 * C + c, c0);
        ld8f(coef2 + c, a2); ld8f(coef2 + C + c, kk2); ld8f(coef2 + 2 * C + c, c02);
      }
      float v[8], w[8], g[8];
      cvt8(xr[k], v); cvt8(zr[k], w); cvt8(gr[k], g);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (mbits[k] >> e) & 1u ? g[e] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = fmaf(a[e], g[e], fmaf(kk[e], v[e], c0[e]));
        w[e] = fmaf(a2[e], g[e], fmaf(kk2[e], w[e], c02[e]));
      }
      st8bf(dx + i * 8, v);
      st8bf(dz + i * 8, w);
    }
  }
}

static int g_apply_rows = 2;   // items per thread of the apply kernels: 1, 2 or 4
static int g_apply2_rows = 2;  // ... of the two-norm backward apply kernel: 1 or 2 (measurement hook)
DTF_API void dtf_set_bn_apply2_rows(int r) { g_apply2_rows = r == 1 ? 1 : 2; }
static int g_apply_nt = 1;     // non-temporal activation loads

DTF_API int dtf_set_bn_apply_rows(int r, int nt) {
  if (r != 1 && r != 2 && r != 4) return -1;
  g_apply_rows = r;
  g_apply_nt = nt != 0;
  return 0;
}

static void launch_apply_fwd(const void* x, const void* res, const float* scale, const float* shift, void* y,
                             void* mask, long long M, int C, int relu, hipStream_t stream,
                             const float* rscale = nullptr, const float* rshift = nullptr) {
#define AFN(R, NT, RN)                                                                                                  \
  hipLaunchKernelGGL((bn_apply_fwd_kernel<R, NT, RN>), dim3(grid_stream(M * (C / 8), kBlock * R)), dim3(kBlock), 0, stream, \
                     (const unsigned short*)x, (const unsigned short*)res, scale, shift, (unsigned short*)y,       \
                     (unsigned char*)mask, M, C, relu, rscale, rshift)
#define AF(R, NT)                                                                                                       \
  do {                                                                                                                  \
    if (rscale && res) AFN(R, NT, true); else AFN(R, NT, false);                                                        \
  } while (0)
  if (g_apply_nt) { if (g_apply_rows == 4) AF(4, true); else if (g_apply_rows == 2) AF(2, true); else AF(1, true); }
  else { if (g_apply_rows == 4) AF(4, false); else if (g_apply_rows == 2) AF(2, false); else AF(1, false); }
#undef AF
#undef AFN
}

static void launch_apply_bwd(const void* x, const void* dy, const void* y, const float* coef, const float* scale,
                             const float* shift, void* dx, void* dres, long long M, int C, int relu,
                             hipStream_t stream) {
#define AB3(R, NT, RL)                                                                                                  \
  hipLaunchKernelGGL((bn_apply_bwd_kernel<R, NT, RL>), dim3(grid_stream(M * (C / 8), kBlock * R)), dim3(kBlock), 0, stream, \
                     (const unsigned short*)x, (const unsigned short*)dy, (const unsigned short*)y, coef, coef + C, \
                     coef + 2 * C, scale, shift, (unsigned short*)dx, (unsigned short*)dres, M, C)
#define AB(R, NT)                                                                                                       \
  do {                                                                                                                  \
    if (relu == 3) AB3(R, NT, 3); else if (relu == 2) AB3(R, NT, 2); else if (relu == 1) AB3(R, NT, 1); else AB3(R, NT, 0); \
  } while (0)
  if (g_apply_nt) { if (g_apply_rows == 4) AB(4, true); else if (g_apply_rows == 2) AB(2, true); else AB(1, true); }
  else { if (g_apply_rows == 4) AB(4, false); else if (g_apply_rows == 2) AB(2, false); else AB(1, false); }
#undef AB
#undef AB3
}

static int bn_partials_for(long long M, int C) {
  const int cchunk = C < 2048 ? C : 2048;
  const int rpb = kBlock / (cchunk / 8);
  long long nb = (M + (long long)rpb * 8 - 1) / ((long long)rpb * 8);   // >= 8 passes per block
  const int cy = (C + 2047) / 2048;
  long long cap = kBnMaxPartials / cy;
  if (nb > cap) nb = cap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

DTF_API int dtf_bn_num_partials(long long M, int C) { return bn_partials_for(M, C); }

// workspace: fp32 [n_partials * 2 * C]. stats: mean, invstd, scale, shift each [C] fp32.
DTF_API int dtf_bn_fwd_train(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                             float* run_mean, float* run_var, float* mean, float* invstd, float* scale, float* shift,
                             float* workspace, void* mask_out, long long M, int C, float eps, float momentum, int relu,
                             hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 8) return -1;
  const int nb = bn_partials_for(M, C);
  const dim3 grid(nb, (C + 2047) / 2048);
  hipLaunchKernelGGL(bn_partial_kernel<0>, grid, dim3(kBlock), 0, stream, (const unsigned short*)x, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, workspace, M, C);
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, workspace, nb, gamma, beta,
                     run_mean, run_var, mean, invstd, scale, shift, M, C, eps, momentum);
  launch_apply_fwd(x, res, scale, shift, y, mask_out, M, C, relu, stream);
  return (int)hipGetLastError();
}

// Folds n_rows partial rows ([n_rows][2][C] fp32, e.g. the per-wave-row column statistics a GEMM epilogue wrote)
// into n_out rows: block (g, chunk) sums rows [g * per, (g + 1) * per) of 256 consecutive floats of the 2 * C wide
// row, four loads in flight, in a fixed order.
__global__ __launch_bounds__(kBlock) void bn_fold_partials_kernel(const float* __restrict__ part, int n_rows, int per,
                                                                  int width, float* __restrict__ out) {
  const int e = blockIdx.y * kBlock + threadIdx.x;
  if (e >= width) return;
  const int r0 = blockIdx.x * per;
  const int r1 = min(n_rows, r0 + per);
  const float* q = part + (long long)r0 * width + e;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  int r = r0;
  // eight loads in flight: the kernel is a chain of dependent round trips (98 rows per block behind a 56x56 layer)
  for (; r + 7 < r1; r += 8, q += 8LL * width) {
    const float x0 = q[0], x1 = q[width], x2 = q[2LL * width], x3 = q[3LL * width], x4 = q[4LL * width],
                x5 = q[5LL * width], x6 = q[6LL * width], x7 = q[7LL * width];
    a += x0 + x4; b += x1 + x5; c += x2 + x6; d += x3 + x7;
  }
  for (; r + 3 < r1; r += 4, q += 4LL * width) {
    a += q[0]; b += q[width]; c += q[2LL * width]; d += q[3LL * width];
  }
  for (; r < r1; ++r, q += width) a += q[0];
  out[(long long)blockIdx.x * width + e] = (a + b) + (c + d);
}

// Fold + finalize in ONE launch, for statistics that arrive as more than kFoldAbove partial rows (the per-tile rows
// a convolution's store phase wrote: 25088 of them behind a 56x56 layer at batch 1024). Block (g, j) sums rows
// [g * per, (g + 1) * per) of 32 channels (both statistics: 64 columns, one per lane of a wave, 16 waves = 16 row
// lanes) and leaves one row in `ws`; the block that arrives last at the channel group's counter sums those rows in a
// fixed order and finishes the statistics, so no float atomics and bitwise reproducible results. The counter is
// back at zero when the kernel ends. MODE 0: forward (mean, 1/std, scale, shift, running statistics), MODE 1:
// backward (dgamma, dbeta, dx coefficients; the partials hold sum dy' * (x - mean)).
constexpr int kRfThreads = 1024, kRfCh = 32, kRfLanes = kRfThreads / (2 * kRfCh), kRfGroups = 64;

struct BnFin {
  const float* gamma; const float* beta; float* run_mean; float* run_var;      // forward
  float* mean; float* invstd; float* scale; float* shift; float eps, momentum;
  float* dgamma; float* dbeta; float* cA; float* cK; float* c0;                // backward (reads mean, invstd)
  long long M; int C;
};

template <int MODE>
__global__ __launch_bounds__(kRfThreads) void bn_reduce_finalize_kernel(const float* __restrict__ part, int n_rows,
                                                                        int per, float* __restrict__ ws,
                                                                        unsigned* __restrict__ counters, BnFin f) {
  __shared__ float red[kRfLanes][2 * kRfCh];
  __shared__ unsigned is_last;
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int ch = j * kRfCh + (col & 31);
  const long long width = 2LL * f.C;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  if (ch < f.C) {
    const int r1 = min(n_rows, (g + 1) * per);
    int r = g * per + rl;
    const float* q = part + (long long)r * width + (col >> 5) * f.C + ch;
    const long long st = kRfLanes * width;
    for (; r + 3 * kRfLanes < r1; r += 4 * kRfLanes, q += 4 * st) { a += q[0]; b += q[st]; c += q[2 * st]; d += q[3 * st]; }
    for (; r < r1; r += kRfLanes, q += st) a += q[0];
  }
  red[rl][col] = (a + b) + (c + d);
  __syncthreads();
  float* mine = ws + ((long long)j * G + g) * 64;
  if (rl == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kRfLanes; ++k) s += red[k][col];
    // Device-scope stores go through to the level all XCDs share, and the wait below returns once they are
    // acknowledged there; the counter (device scope as well) moves only after that. A release fence instead would
    // write back the whole L2 of this XCD, dirty with the activation the producer has just stored: measured 275 us
    // per launch against 10.
    __hip_atomic_store(mine + col, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (threadIdx.x == 0)
    is_last = __hip_atomic_fetch_add(counters + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1);
  __syncthreads();
  if (!is_last) return;
  // G <= kRfGroups rows, kRfGroups / kRfLanes per row lane: all loads in flight before the first add
  float v[kRfGroups / kRfLanes];
#pragma unroll
  for (int i = 0; i < kRfGroups / kRfLanes; ++i) {
    const int k = rl + i * kRfLanes;
    v[i] = __hip_atomic_load(ws + ((long long)j * G + min(k, G - 1)) * 64 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k >= G) v[i] = 0.f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kRfGroups / kRfLanes; ++i) s += v[i];
  red[rl][col] = s;
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(counters + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int cc = j * kRfCh + threadIdx.x;
  if (threadIdx.x >= kRfCh || cc >= f.C) return;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int k = 0; k < kRfLanes; ++k) { s0 += red[k][threadIdx.x]; s1 += red[k][kRfCh + threadIdx.x]; }
  const float inv_m = 1.f / (float)f.M;
  if (MODE == 0) {
    const float mu = s0 * inv_m;
    const float var = fmaxf(s1 * inv_m - mu * mu, 0.f);
    const float is = rsqrtf(var + f.eps);
    f.mean[cc] = mu; f.invstd[cc] = is;
    const float sc = f.gamma[cc] * is;
    f.scale[cc] = sc; f.shift[cc] = f.beta[cc] - mu * sc;
    if (f.run_mean) {
      const float unbiased = f.M > 1 ? var * (float)f.M / (float)(f.M - 1) : var;
      f.run_mean[cc] = (1.f - f.momentum) * f.run_mean[cc] + f.momentum * mu;
      f.run_var[cc] = (1.f - f.momentum) * f.run_var[cc] + f.momentum * unbiased;
    }
  } else {
    const float is = f.invstd[cc];
    s1 *= is;
    f.dbeta[cc] = s0; f.dgamma[cc] = s1;
    const float sc = f.gamma[cc] * is;
    const float k = sc * s1 * inv_m * is;
    f.cA[cc] = sc; f.cK[cc] = -k; f.c0[cc] = f.mean[cc] * k - sc * s0 * inv_m;
  }
}

template <int MODE>
static void launch_reduce_finalize(const float* part, int n_rows, float* ws, unsigned* counters, const BnFin& f,
                                   hipStream_t stream) {
  const int per = (n_rows + kRfGroups - 1) / kRfGroups;
  const int G = (n_rows + per - 1) / per;
  hipLaunchKernelGGL(bn_reduce_finalize_kernel<MODE>, dim3(G, (f.C + kRfCh - 1) / kRfCh), dim3(kRfThreads), 0, stream,
                     part, n_rows, per, ws, counters, f);
}

// counters a launch of the fused reduction needs (one per 32 channels), all zero before and after
DTF_API int dtf_bn_reduce_counters(int C) { return (C + kRfCh - 1) / kRfCh; }

constexpr int kFoldRows = 512;            // rows a fold leaves
constexpr int kFoldAbove = kBnMaxPartials; // the finalize kernels take up to this many rows directly

// dtf_bn_fwd_train with the batch statistics already reduced to partial rows by the producer of x: col_stats is
// fp32 [n_rows][2][C] (sum, sum of squares per channel over disjoint row sets that cover all M rows), so x is read
// once (by the apply pass) instead of twice. workspace: fp32 [kFoldRows * 2 * C] (used when n_rows > kFoldAbove).
static int bn_fwd_train_stats_impl(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                   float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                   float* shift, const float* col_stats, int n_rows, float* workspace, void* mask_out,
                                   long long M, int C, float eps, float momentum, int relu, void* counters,
                                   const float* rscale, const float* rshift, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 8 || !col_stats || n_rows < 1) return -1;
  const float* part = col_stats;
  int np = n_rows;
  if (n_rows > kFoldAbove && counters) {
    BnFin f = {};
    f.gamma = gamma; f.beta = beta; f.run_mean = run_mean; f.run_var = run_var; f.mean = mean; f.invstd = invstd;
    f.scale = scale; f.shift = shift; f.eps = eps; f.momentum = momentum; f.M = M; f.C = C;
    launch_reduce_finalize<0>(col_stats, n_rows, workspace, (unsigned*)counters, f, stream);
    if (y) launch_apply_fwd(x, res, scale, shift, y, mask_out, M, C, relu, stream, rscale, rshift);
    return (int)hipGetLastError();
  }
  if (n_rows > kFoldAbove) {
    const int per = (n_rows + kFoldRows - 1) / kFoldRows;
    np = (n_rows + per - 1) / per;
    hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(np, (2 * C + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       col_stats, n_rows, per, 2 * C, workspace);
    part = workspace;
  }
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, part, np, gamma,
                     beta, run_mean, run_var, mean, invstd, scale, shift, M, C, eps, momentum);
  // y == null: the consumer of the normalised activation applies scale / shift (/ ReLU) itself while it stages its
  // operand (gemm.hip, operand transform); only the statistics are finished here
  if (y)
    launch_apply_fwd(x, res, scale, shift, y, mask_out, M, C, relu, stream, rscale, rshift);
  return (int)hipGetLastError();
}

DTF_API int dtf_bn_fwd_train_stats(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                   float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                   float* shift, const float* col_stats, int n_rows, float* workspace, void* mask_out,
                                   long long M, int C, float eps, float momentum, int relu, void* counters,
                                   hipStream_t stream) {
  return bn_fwd_train_stats_impl(x, res, y, gamma, beta, run_mean, run_var, mean, invstd, scale, shift, col_stats, n_rows,
                                 workspace, mask_out, M, C, eps, momentum, relu, counters, nullptr, nullptr, stream);
}

// The same with a residual that is the INPUT of another batch norm (rscale, rshift: that norm's fp32 coefficients):
// y = act(x * scale + shift + res * rscale + rshift)
DTF_API int dtf_bn_fwd_train_stats_rn(const void* x, const void* res, void* y, const float* gamma, const float* beta,
                                      float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                      float* shift, const float* col_stats, int n_rows, float* workspace,
                                      void* mask_out, long long M, int C, float eps, float momentum, int relu,
                                      void* counters, const float* rscale, const float* rshift, hipStream_t stream) {
  if (!res || !rscale || !rshift || !y) return -1;
  return bn_fwd_train_stats_impl(x, res, y, gamma, beta, run_mean, run_var, mean, invstd, scale, shift, col_stats, n_rows,
                                 workspace, mask_out, M, C, eps, momentum, relu, counters, rscale, rshift, stream);
}

DTF_API int dtf_bn_fwd_apply(const void* x, const void* res, void* y, const float* scale, const float* shift,
                             long long M, int C, int relu, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 8) return -1;
  launch_apply_fwd(x, res, scale, shift, y, nullptr, M, C, relu, stream);
  return (int)hipGetLastError();
}

// coef: fp32 [3 * C] scratch (A, K, C0). relu: 0 none, 1 mask from the saved output y, 2 mask recomputed
// from x with the forward's scale/shift (layers without a residual input; y is not read at all), 3 mask
// bits written by dtf_bn_fwd_train (y points to that byte array: M * C / 8 bytes).
// workspace (fp32 partial sums) and mask_out: see dtf_bn_num_partials.
DTF_API int dtf_bn_bwd(const void* x, const void* dy, const void* y, const float* gamma, const float* mean,
                       const float* invstd, const float* scale, const float* shift, void* dx, void* dres,
                       float* dgamma, float* dbeta, float* coef, float* workspace, long long M, int C, int relu,
                       hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (C % 8) return -1;
  const int nb = bn_partials_for(M, C);
  const dim3 grid(nb, (C + 2047) / 2048);
#define BP(RL)                                                                                                 \
  hipLaunchKernelGGL((bn_partial_kernel<1, RL>), grid, dim3(kBlock), 0, stream, (const unsigned short*)x,       \
                     (const unsigned short*)dy, (const unsigned short*)y, mean, invstd, scale, shift, workspace, M, C)
  if (relu == 3) BP(3); else if (relu == 2) BP(2); else if (relu == 1) BP(1); else BP(0);
#undef BP
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, workspace, nb, gamma, mean,
                     invstd, dgamma, dbeta, coef, coef + C, coef + 2 * C, M, C, 0);
  launch_apply_bwd(x, dy, y, coef, scale, shift, dx, dres, M, C, relu, stream);
  return (int)hipGetLastError();
}

// dtf_bn_bwd with the reductions over the incoming gradient already done by its producer (the data-gradient GEMM
// of the following convolution): col_stats is fp32 [n_rows][2][C] holding sum dy' and sum dy' * (x - mean) over
// disjoint row sets, dy' being the gradient after the ReLU mask. x and dy are then read once (by the apply pass).
// workspace: fp32 [kFoldRows * 2 * C].
static int bn_bwd_stats_coef(const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                             float* coef, const float* col_stats, int n_rows, float* workspace, long long M, int C,
                             void* counters, hipStream_t stream) {
  if (C % 8 || !col_stats || n_rows < 1) return -1;
  const float* part = col_stats;
  int np = n_rows;
  if (n_rows > kFoldAbove && counters) {
    BnFin f = {};
    f.gamma = gamma; f.mean = const_cast<float*>(mean); f.invstd = const_cast<float*>(invstd); f.dgamma = dgamma;
    f.dbeta = dbeta; f.cA = coef; f.cK = coef + C; f.c0 = coef + 2 * C; f.M = M; f.C = C;
    launch_reduce_finalize<1>(col_stats, n_rows, workspace, (unsigned*)counters, f, stream);
    return 0;
  }
  if (n_rows > kFoldAbove) {
    const int per = (n_rows + kFoldRows - 1) / kFoldRows;
    np = (n_rows + per - 1) / per;
    hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(np, (2 * C + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                       col_stats, n_rows, per, 2 * C, workspace);
    part = workspace;
  }
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, part, np, gamma,
                     mean, invstd, dgamma, dbeta, coef, coef + C, coef + 2 * C, M, C, 1);
  return 0;
}

DTF_API int dtf_bn_bwd_stats(const void* x, const void* dy, const void* y, const float* gamma, const float* mean,
                             const float* invstd, const float* scale, const float* shift, void* dx, void* dres,
                             float* dgamma, float* dbeta, float* coef, const float* col_stats, int n_rows,
                             float* workspace, long long M, int C, int relu, void* counters, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (bn_bwd_stats_coef(gamma, mean, invstd, dgamma, dbeta, coef, col_stats, n_rows, workspace, M, C, counters, stream))
    return -1;
  launch_apply_bwd(x, dy, y, coef, scale, shift, dx, dres, M, C, relu, stream);
  return (int)hipGetLastError();
}

// The closing norm of a down-sampling block (input x, sums of its gradient already in col_stats as for
// dtf_bn_bwd_stats) together with the plain norm of the down-sampling branch (input z; `2` arguments), which receives
// the same gradient dy under the same mask bits: the branch norm's two sums are reduced here (one pass over dy, the
// bits and z), then ONE apply pass writes dx and dz. workspace2: fp32 [dtf_bn_num_partials * 2 * C].
DTF_API int dtf_bn_bwd_stats_dual(const void* x, const void* dy, const void* mask, const float* gamma, const float* mean,
                                  const float* invstd, void* dx, float* dgamma, float* dbeta, float* coef,
                                  const float* col_stats, int n_rows, float* workspace, const void* z,
                                  const float* gamma2, const float* mean2, const float* invstd2, void* dz,
                                  float* dgamma2, float* dbeta2, float* coef2, float* workspace2, long long M, int C,
                                  void* counters, hipStream_t stream) {
  if (M <= 0 || C <= 0) return 0;
  if (!mask || !z || !dz || !dx) return -1;
  if (bn_bwd_stats_coef(gamma, mean, invstd, dgamma, dbeta, coef, col_stats, n_rows, workspace, M, C, counters, stream))
    return -1;
  const int nb = bn_partials_for(M, C);
  hipLaunchKernelGGL((bn_partial_kernel<1, 3>), dim3(nb, (C + 2047) / 2048), dim3(kBlock), 0, stream,
                     (const unsigned short*)z, (const unsigned short*)dy, (const unsigned short*)mask, mean2, invstd2,
                     (const float*)nullptr, (const float*)nullptr, workspace2, M, C);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, workspace2, nb,
                     gamma2, mean2, invstd2, dgamma2, dbeta2, coef2, coef2 + C, coef2 + 2 * C, M, C, 0);
#define AB2(R, NT)                                                                                                       \
  hipLaunchKernelGGL((bn_apply_bwd2_kernel<R, NT>), dim3(grid_stream(M * (C / 8), kBlock * R)), dim3(kBlock), 0, stream, \
                     (const unsigned short*)x, (const unsigned short*)z, (const unsigned short*)dy,                     \
                     (const unsigned char*)mask, coef, coef2, (unsigned short*)dx, (unsigned short*)dz, M, C)
  if (g_apply_nt) { if (g_apply2_rows == 1) AB2(1, true); else AB2(2, true); }
  else { if (g_apply2_rows == 1) AB2(1, false); else AB2(2, false); }
#undef AB2
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------ norm + ReLU + max pool in one pass
// The ResNet stem ends in batch norm, ReLU and a 3x3 / stride-2 max pool over the widest activation of the network.
// Fused, the normalised activation is never written: the forward pass normalises each tap while it takes the window
// maximum, and the backward kernels gather the gradient of an input pixel from the (at most ceil(K/S)^2) windows
// that chose it -- the pool's gather backward -- while they do the norm's two backward passes. Tap values are rounded
// to bf16 before they are compared, so maxima and winning taps are those of the unfused kernels.
struct BnPoolGeom {
  int N, H, W, C, OH, OW, K, S, P;
};
typedef unsigned char bp_u8x8 __attribute__((ext_vector_type(8)));

// All three kernels walk image rows: blockIdx.x picks (n, row) with one scalar division per row, and a thread keeps
// its 8-channel chunk and its pixel slot inside the row for the whole kernel, so the inner loops contain no integer
// division. KS = 1 compiles the 3x3 / stride-2 / pad-1 geometry of the ResNet stem with constant bounds; KS = 0
// takes the geometry from the arguments.
template <int KS>
__device__ __forceinline__ void bp_geom(const BnPoolGeom& g, int& K, int& S, int& P) {
  if (KS == 1) { K = 3; S = 2; P = 1; } else { K = g.K; S = g.S; P = g.P; }
}

template <int KS>
__global__ __launch_bounds__(kBlock) void bn_relu_maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     unsigned short* __restrict__ y,
                                                                     unsigned char* __restrict__ idx, BnPoolGeom g) {
  int K, S, P;
  bp_geom<KS>(g, K, S, P);
  const int c8 = g.C >> 3;
  const int tc = threadIdx.x % c8, slot = threadIdx.x / c8, slots = kBlock / c8;
  const int c = tc << 3;
  float sa[8], sb[8];
  ld8f(scale + c, sa); ld8f(shift + c, sb);
  const int rows = g.N * g.OH;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / g.OH, oh = row - n * g.OH;
    for (int ow = slot; ow < g.OW; ow += slots) {
      // One 32-bit key per element: the rounded, rectified value's bf16 bits above 255 - tap. The value is not
      // negative, so its bits order like unsigned integers, and among equal values the larger low byte -- the earlier
      // tap -- wins: an unsigned maximum per tap replaces the compare / select chain (first maximum wins, as in
      // the unfused pool). Any tap beats the initial key 0.
      unsigned key[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) key[k] = 0u;
#pragma unroll 3
      for (int kh = 0; kh < K; ++kh) {
        const int ih = oh * S - P + kh;
        if (ih < 0 || ih >= g.H) continue;
#pragma unroll 3
        for (int kw = 0; kw < K; ++kw) {
          const int iw = ow * S - P + kw;
          if (iw < 0 || iw >= g.W) continue;
          float v[8];
          ld8bf(x + (((long long)n * g.H + ih) * g.W + iw) * g.C + c, v);
          const unsigned code = 255u - (unsigned)(kh * K + kw);
#pragma unroll
          for (int k = 0; k < 8; k += 2) {
            // ReLU as a signed-integer maximum of the float's bits: negative values (sign bit set) become 0 and a
            // NaN keeps its bits, so it wins the key comparison below and reaches the output like in the unfused
            // norm + ReLU + pool (a float maximum would turn it into 0); one instruction either way
            const float lo = __int_as_float(max(__float_as_int(fmaf(v[k], sa[k], sb[k])), 0));
            const float hi = __int_as_float(max(__float_as_int(fmaf(v[k + 1], sa[k + 1], sb[k + 1])), 0));
            const unsigned pk = pack_bf16x2(lo, hi) & 0x7fff7fffu;       // (a negative zero must not outrank anything)
            key[k] = max(key[k], (pk << 16) | code);
            key[k + 1] = max(key[k + 1], (pk & 0xffff0000u) | code);
          }
        }
      }
      bp_u8x8 a;
      u32x4_t out;
#pragma unroll
      for (int k = 0; k < 8; ++k) a[k] = (unsigned char)(255u - (key[k] & 0xffu));
#pragma unroll
      for (int k = 0; k < 8; k += 2) out[k >> 1] = (key[k] >> 16) | (key[k + 1] & 0xffff0000u);
      const long long o = ((long long)row * g.OW + ow) * g.C + c;
      *reinterpret_cast<u32x4_t*>(y + o) = out;
      *reinterpret_cast<bp_u8x8*>(idx + o) = a;
    }
  }
}

// gradient of input pixel (n, ih, iw), channels c..c+7, of the max pool: sum over the windows that chose it
template <int KS>
__device__ __forceinline__ void pool_gather_grad(const unsigned short* __restrict__ gy, const unsigned char* __restrict__ idx,
                                                 const BnPoolGeom& g, int n, int ih, int iw, int c, float (&acc)[8]) {
  int K, S, P;
  bp_geom<KS>(g, K, S, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  const int oh_hi = min(g.OH - 1, (ih + P) / S);
  const int ow_hi = min(g.OW - 1, (iw + P) / S);
  const int oh_lo = max(0, (ih + P - K + S) / S);      // ceil((ih + P - K + 1) / S) where it is non-negative
  const int ow_lo = max(0, (iw + P - K + S) / S);
  for (int oh = oh_lo; oh <= oh_hi; ++oh) {
    const int kh = ih - (oh * S - P);
    if (kh < 0 || kh >= K) continue;
    for (int ow = ow_lo; ow <= ow_hi; ++ow) {
      const int kw = iw - (ow * S - P);
      if (kw < 0 || kw >= K) continue;
      const long long o = (((long long)n * g.OH + oh) * g.OW + ow) * g.C + c;
      const bp_u8x8 a = *reinterpret_cast<const bp_u8x8*>(idx + o);
      float v[8];
      ld8bf(gy + o, v);
      const unsigned char tap = (unsigned char)(kh * K + kw);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += a[k] == tap ? v[k] : 0.f;
    }
  }
}

// pass 1 of the backward: per-block partial sum dy', sum dy' * xhat with dy' = pool gradient masked by the ReLU
// (partial layout of bn_partial_kernel<1>: part[block][2][C]; C <= 2048 and C / 8 divides the block)
template <int KS>
__global__ __launch_bounds__(kBlock) void bn_pool_partial_bwd_kernel(const unsigned short* __restrict__ x,
                                                                     const unsigned short* __restrict__ gy,
                                                                     const unsigned char* __restrict__ idx,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     float* __restrict__ part, BnPoolGeom g) {
  __shared__ float red[2][kBlock][8];
  const int C = g.C, c8 = C >> 3;
  const int tc = threadIdx.x % c8, slot = threadIdx.x / c8, slots = kBlock / c8;
  const int c0 = tc * 8;
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mu[8], is[8], sa[8], sb[8];
  ld8f(mean + c0, mu); ld8f(invstd + c0, is); ld8f(scale + c0, sa); ld8f(shift + c0, sb);
  const int rows = g.N * g.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / g.H, ih = row - n * g.H;
    for (int iw = slot; iw < g.W; iw += slots) {
      float v[8], d[8];
      ld8bf(x + ((long long)row * g.W + iw) * C + c0, v);
      pool_gather_grad<KS>(gy, idx, g, n, ih, iw, c0, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dk = fmaf(v[k], sa[k], sb[k]) > 0.f ? d[k] : 0.f;
        s0[k] += dk;
        s1[k] = fmaf(dk, (v[k] - mu[k]) * is[k], s1[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; }
  __syncthreads();
  if (slot == 0) {
    for (int r = 1; r < slots; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0[k] += red[0][r * c8 + tc][k]; s1[k] += red[1][r * c8 + tc][k]; }
    }
    float* p0 = part + ((long long)blockIdx.x * 2) * C + c0;
    float* p1 = p0 + C;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p0[k] = s0[k]; p1[k] = s1[k]; }
  }
}

// pass 2: dx = A * dy' + K * x + C0
template <int KS>
__global__ __launch_bounds__(kBlock) void bn_pool_apply_bwd_kernel(const unsigned short* __restrict__ x,
                                                                   const unsigned short* __restrict__ gy,
                                                                   const unsigned char* __restrict__ idx,
                                                                   const float* __restrict__ cA, const float* __restrict__ cK,
                                                                   const float* __restrict__ c0, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift,
                                                                   unsigned short* __restrict__ dx, BnPoolGeom g) {
  const int c8 = g.C >> 3;
  const int tc = threadIdx.x % c8, slot = threadIdx.x / c8, slots = kBlock / c8;
  const int c = tc << 3;
  float a[8], kk[8], z[8], sa[8], sb[8];
  ld8f(cA + c, a); ld8f(cK + c, kk); ld8f(c0 + c, z); ld8f(scale + c, sa); ld8f(shift + c, sb);
  const int rows = g.N * g.H;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / g.H, ih = row - n * g.H;
    for (int iw = slot; iw < g.W; iw += slots) {
      const long long o = ((long long)row * g.W + iw) * g.C + c;
      float v[8], d[8];
      ld8bf(x + o, v);
      pool_gather_grad<KS>(gy, idx, g, n, ih, iw, c, d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float dk = fmaf(v[k], sa[k], sb[k]) > 0.f ? d[k] : 0.f;
        v[k] = fmaf(a[k], dk, fmaf(kk[k], v[k], z[k]));
      }
      st8bf(dx + o, v);
    }
  }
}

// ---- 3x3 / stride 2 / pad 1 backward on 2x2 input patches -------------------------------------------------------------
// The patch of input pixels (2m..2m+1, 2j..2j+1) lies in exactly the four windows (m..m+1, j..j+1), so a thread that
// owns a patch loads each window's gradient and tap once (instead of once per covered pixel) and no lane diverges:
//   pixel (2m  , 2j  ): window (m  , j  ) tap 4
//   pixel (2m  , 2j+1): window (m  , j  ) tap 5, window (m  , j+1) tap 3
//   pixel (2m+1, 2j  ): window (m  , j  ) tap 7, window (m+1, j  ) tap 1
//   pixel (2m+1, 2j+1): window (m  , j  ) tap 8, window (m  , j+1) tap 6, window (m+1, j) tap 2, window (m+1, j+1) tap 0
// The patch stays packed as loaded (bf16 pairs, four taps per word) and is unpacked one channel at a time, which
// keeps the kernels near 100 registers; unpacked to fp32 up front they needed over 200 and ran two waves per SIMD.
typedef unsigned bp_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned bp_u32x2 __attribute__((ext_vector_type(2)));

struct BpPatch {
  bp_u32x4 x[2][2];     // input pixels (zero where outside the image)
  bp_u32x4 g[2][2];     // gradients of the four windows (zero where outside the pooled image)
  bp_u32x2 t[2][2];     // winning taps of the four windows (0xff where outside)
  bool in[2][2];
};

__device__ __forceinline__ float bp_bf(const bp_u32x4& v, int k) {
  const unsigned w = v[k >> 1];
  return __builtin_bit_cast(float, (k & 1) ? (w & 0xffff0000u) : (w << 16));
}
__device__ __forceinline__ unsigned bp_tap(const bp_u32x2& v, int k) {
  const unsigned w = v[k >> 2];
  return (w >> (8 * (k & 3))) & 0xffu;
}

__device__ __forceinline__ void bp_load_patch(const unsigned short* __restrict__ x, const unsigned short* __restrict__ gy,
                                              const unsigned char* __restrict__ idx, const BnPoolGeom& g, int n, int m,
                                              int j, int c, BpPatch& p) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int oh = m + a, ow = j + b;
      p.g[a][b] = bp_u32x4{0u, 0u, 0u, 0u};
      p.t[a][b] = bp_u32x2{0xffffffffu, 0xffffffffu};
      if (oh < g.OH && ow < g.OW) {
        const long long o = (((long long)n * g.OH + oh) * g.OW + ow) * g.C + c;
        p.t[a][b] = *reinterpret_cast<const bp_u32x2*>(idx + o);
        p.g[a][b] = *reinterpret_cast<const bp_u32x4*>(gy + o);
      }
      const int ih = 2 * m + a, iw = 2 * j + b;
      p.in[a][b] = ih < g.H && iw < g.W;
      p.x[a][b] = bp_u32x4{0u, 0u, 0u, 0u};
      if (p.in[a][b]) p.x[a][b] = *reinterpret_cast<const bp_u32x4*>(x + (((long long)n * g.H + ih) * g.W + iw) * g.C + c);
    }
}

// pool gradient of channel k of the four patch pixels, before the ReLU mask
__device__ __forceinline__ void bp_grads(const BpPatch& p, int k, float (&d)[2][2]) {
  const unsigned t00 = bp_tap(p.t[0][0], k), t01 = bp_tap(p.t[0][1], k), t10 = bp_tap(p.t[1][0], k),
                 t11 = bp_tap(p.t[1][1], k);
  const float g00 = bp_bf(p.g[0][0], k), g01 = bp_bf(p.g[0][1], k), g10 = bp_bf(p.g[1][0], k), g11 = bp_bf(p.g[1][1], k);
  d[0][0] = t00 == 4 ? g00 : 0.f;
  d[0][1] = (t00 == 5 ? g00 : 0.f) + (t01 == 3 ? g01 : 0.f);
  d[1][0] = (t00 == 7 ? g00 : 0.f) + (t10 == 1 ? g10 : 0.f);
  d[1][1] = ((t00 == 8 ? g00 : 0.f) + (t01 == 6 ? g01 : 0.f)) + ((t10 == 2 ? g10 : 0.f) + (t11 == 0 ? g11 : 0.f));
}

__global__ __launch_bounds__(kBlock, 4) void bn_pool322_partial_bwd_kernel(const unsigned short* __restrict__ x,
                                                                        const unsigned short* __restrict__ gy,
                                                                        const unsigned char* __restrict__ idx,
                                                                        const float* __restrict__ mean,
                                                                        const float* __restrict__ invstd,
                                                                        const float* __restrict__ scale,
                                                                        const float* __restrict__ shift,
                                                                        float* __restrict__ part, BnPoolGeom g) {
  __shared__ float red[2][kBlock][8];
  const int C = g.C, c8 = C >> 3;
  const int tc = threadIdx.x % c8, slot = threadIdx.x / c8, slots = kBlock / c8;
  const int c0 = tc * 8;
  float s0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float mu[8], sa[8], sb[8];
  ld8f(mean + c0, mu); ld8f(scale + c0, sa); ld8f(shift + c0, sb);
  const int PH = (g.H + 1) >> 1, PW = (g.W + 1) >> 1;
  const int rows = g.N * PH;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / PH, m = row - n * PH;
    for (int j = slot; j < PW; j += slots) {
      BpPatch p;
      bp_load_patch(x, gy, idx, g, n, m, j, c0, p);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float d[2][2];
        bp_grads(p, k, d);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float v = bp_bf(p.x[a][b], k);
            const float dk = fmaf(v, sa[k], sb[k]) > 0.f ? d[a][b] : 0.f;   // d is zero outside the image
            s0[k] += dk;
            s1[k] = fmaf(dk, v - mu[k], s1[k]);              // times invstd once, below
          }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { red[0][threadIdx.x][k] = s0[k]; red[1][threadIdx.x][k] = s1[k]; }
  __syncthreads();
  if (slot == 0) {
    for (int r = 1; r < slots; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0[k] += red[0][r * c8 + tc][k]; s1[k] += red[1][r * c8 + tc][k]; }
    }
    float is[8];
    ld8f(invstd + c0, is);
    float* p0 = part + ((long long)blockIdx.x * 2) * C + c0;
    float* p1 = p0 + C;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p0[k] = s0[k]; p1[k] = s1[k] * is[k]; }
  }
}

__global__ __launch_bounds__(kBlock, 4) void bn_pool322_apply_bwd_kernel(const unsigned short* __restrict__ x,
                                                                      const unsigned short* __restrict__ gy,
                                                                      const unsigned char* __restrict__ idx,
                                                                      const float* __restrict__ cA, const float* __restrict__ cK,
                                                                      const float* __restrict__ c0, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift,
                                                                      unsigned short* __restrict__ dx, BnPoolGeom g) {
  const int c8 = g.C >> 3;
  const int tc = threadIdx.x % c8, slot = threadIdx.x / c8, slots = kBlock / c8;
  const int c = tc << 3;
  float a8[8], kk[8], z[8], sa[8], sb[8];
  ld8f(cA + c, a8); ld8f(cK + c, kk); ld8f(c0 + c, z); ld8f(scale + c, sa); ld8f(shift + c, sb);
  const int PH = (g.H + 1) >> 1, PW = (g.W + 1) >> 1;
  const int rows = g.N * PH;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int n = row / PH, m = row - n * PH;
    for (int j = slot; j < PW; j += slots) {
      BpPatch p;
      bp_load_patch(x, gy, idx, g, n, m, j, c, p);
      bp_u32x4 o[2][2];
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        float d0[2][2], d1[2][2];
        bp_grads(p, k, d0);
        bp_grads(p, k + 1, d1);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float v0 = bp_bf(p.x[a][b], k), v1 = bp_bf(p.x[a][b], k + 1);
            const float e0 = fmaf(v0, sa[k], sb[k]) > 0.f ? d0[a][b] : 0.f;
            const float e1 = fmaf(v1, sa[k + 1], sb[k + 1]) > 0.f ? d1[a][b] : 0.f;
            const float r0 = fmaf(a8[k], e0, fmaf(kk[k], v0, z[k]));
            const float r1 = fmaf(a8[k + 1], e1, fmaf(kk[k + 1], v1, z[k + 1]));
            o[a][b][k >> 1] = pack_bf16x2(r0, r1);
          }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (p.in[a][b])
            *reinterpret_cast<bp_u32x4*>(dx + (((long long)n * g.H + 2 * m + a) * g.W + 2 * j + b) * g.C + c) = o[a][b];
    }
  }
}

static bool bn_pool_geom_ok(const BnPoolGeom& g) {
  return g.N > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.C % 8 == 0 && g.C <= 2048 && kBlock % (g.C / 8) == 0 && g.K >= 1 &&
         g.K <= 15 && g.S >= 1 && g.P >= 0 && g.P < g.K && g.OH == (g.H + 2 * g.P - g.K) / g.S + 1 &&
         g.OW == (g.W + 2 * g.P - g.K) / g.S + 1;
}

DTF_API int dtf_bn_relu_maxpool_fwd_stats(const void* x, void* y, void* idx, const float* gamma, const float* beta,
                                          float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                          float* shift, const float* col_stats, int n_rows, float* workspace, int N,
                                          int H, int W, int C, int OH, int OW, int K, int S, int P, float eps,
                                          float momentum, hipStream_t stream);

// y [N,OH,OW,C] = maxpool(relu(bn(x))), idx = winning taps (kh * K + kw); statistics as dtf_bn_fwd_train.
// workspace: fp32 [dtf_bn_num_partials(N*H*W, C) * 2 * C]. Returns -1 for an unsupported geometry.
DTF_API int dtf_bn_relu_maxpool_fwd(const void* x, void* y, void* idx, const float* gamma, const float* beta,
                                    float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                    float* shift, float* workspace, int N, int H, int W, int C, int OH, int OW, int K,
                                    int S, int P, float eps, float momentum, hipStream_t stream) {
  return dtf_bn_relu_maxpool_fwd_stats(x, y, idx, gamma, beta, run_mean, run_var, mean, invstd, scale, shift, nullptr, 0,
                                       workspace, N, H, W, C, OH, OW, K, S, P, eps, momentum, stream);
}

// The same with the batch statistics of x already reduced to partial rows by its producer (col_stats: fp32
// [n_rows][2][C], as for dtf_bn_fwd_train_stats; null: reduced here). workspace: fp32
// [max(dtf_bn_num_partials(N*H*W, C), 256) * 2 * C].
DTF_API int dtf_bn_relu_maxpool_fwd_stats(const void* x, void* y, void* idx, const float* gamma, const float* beta,
                                          float* run_mean, float* run_var, float* mean, float* invstd, float* scale,
                                          float* shift, const float* col_stats, int n_rows, float* workspace, int N,
                                          int H, int W, int C, int OH, int OW, int K, int S, int P, float eps,
                                          float momentum, hipStream_t stream) {
  const BnPoolGeom g = {N, H, W, C, OH, OW, K, S, P};
  if (!bn_pool_geom_ok(g)) return -1;
  const long long M = (long long)N * H * W;
  const float* part = workspace;
  int np;
  if (col_stats) {
    if (n_rows < 1) return -1;
    part = col_stats;
    np = n_rows;
    if (n_rows > kFoldAbove) {
      const int per = (n_rows + kFoldRows - 1) / kFoldRows;
      np = (n_rows + per - 1) / per;
      hipLaunchKernelGGL(bn_fold_partials_kernel, dim3(np, (2 * C + kBlock - 1) / kBlock), dim3(kBlock), 0, stream,
                         col_stats, n_rows, per, 2 * C, workspace);
      part = workspace;
    }
  } else {
    np = bn_partials_for(M, C);
    hipLaunchKernelGGL(bn_partial_kernel<0>, dim3(np, 1), dim3(kBlock), 0, stream, (const unsigned short*)x, nullptr,
                       nullptr, nullptr, nullptr, nullptr, nullptr, workspace, M, C);
  }
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, part, np,
                     gamma, beta, run_mean, run_var, mean, invstd, scale, shift, M, C, eps, momentum);
  const int grid = grid_stream((long long)N * OH, 1);
  if (K == 3 && S == 2 && P == 1)
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<1>, dim3(grid), dim3(kBlock), 0, stream, (const unsigned short*)x, scale,
                       shift, (unsigned short*)y, (unsigned char*)idx, g);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel<0>, dim3(grid), dim3(kBlock), 0, stream, (const unsigned short*)x, scale,
                       shift, (unsigned short*)y, (unsigned char*)idx, g);
  return (int)hipGetLastError();
}

// dx [N,H,W,C], dgamma, dbeta from gy [N,OH,OW,C]; coef: fp32 [3 * C] scratch; workspace as in the forward.
DTF_API int dtf_bn_relu_maxpool_bwd(const void* x, const void* gy, const void* idx, const float* gamma, const float* mean,
                                    const float* invstd, const float* scale, const float* shift, void* dx, float* dgamma,
                                    float* dbeta, float* coef, float* workspace, int N, int H, int W, int C, int OH,
                                    int OW, int K, int S, int P, hipStream_t stream) {
  const BnPoolGeom g = {N, H, W, C, OH, OW, K, S, P};
  if (!bn_pool_geom_ok(g)) return -1;
  const long long M = (long long)N * H * W;
  const bool stem = K == 3 && S == 2 && P == 1;
  // one partial row per block of the row-walking reduction; never more than the workspace of the forward holds
  const long long rows = stem ? (long long)N * ((H + 1) / 2) : (long long)N * H;
  const int nb = (int)std::min<long long>(bn_partials_for(M, C), rows);
  const int grid = grid_stream(rows, 1);
  const unsigned short *xp = (const unsigned short*)x, *gp = (const unsigned short*)gy;
  const unsigned char* ip = (const unsigned char*)idx;
  if (stem)
    hipLaunchKernelGGL(bn_pool322_partial_bwd_kernel, dim3(nb), dim3(kBlock), 0, stream, xp, gp, ip, mean, invstd, scale,
                       shift, workspace, g);
  else
    hipLaunchKernelGGL(bn_pool_partial_bwd_kernel<0>, dim3(nb), dim3(kBlock), 0, stream, xp, gp, ip, mean, invstd, scale,
                       shift, workspace, g);
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + kFinCh - 1) / kFinCh), dim3(kBlock), 0, stream, workspace, nb,
                     gamma, mean, invstd, dgamma, dbeta, coef, coef + C, coef + 2 * C, M, C, 0);
  if (stem)
    hipLaunchKernelGGL(bn_pool322_apply_bwd_kernel, dim3(grid), dim3(kBlock), 0, stream, xp, gp, ip, coef, coef + C,
                       coef + 2 * C, scale, shift, (unsigned short*)dx, g);
  else
    hipLaunchKernelGGL(bn_pool_apply_bwd_kernel<0>, dim3(grid), dim3(kBlock), 0, stream, xp, gp, ip, coef, coef + C,
                       coef + 2 * C, scale, shift, (unsigned short*)dx, g);
  return (int)hipGetLastError();
}
