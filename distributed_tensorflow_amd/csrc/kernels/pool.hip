// Max pooling for NHWC bf16 activations (gfx950), forward and backward.
//
// A thread owns 8 consecutive channels (one 16-byte access) of one pixel. Forward scans the
// window in (kh, kw) order with a strict '>' so that the FIRST maximum wins (the convention of
// the vendor library, which matters after a ReLU where whole windows are zero) and records the
// winning tap in one byte per element. Backward is a gather, not a scatter: every input pixel
// looks at the (at most ceil(k/s)^2) windows that contain it and adds the gradients of those
// that chose it, so there are no atomics and the result is bitwise reproducible.
// Used by the ResNet-50 stem (3x3, stride 2, pad 1) and the MNIST CNN (2x2, stride 2); the
// reference model has no pooling layer (reference initializer.py:12-21).
#include "dtf_common.h"

using namespace dtf;

namespace {

struct PoolGeom {
  int N, H, W, C, OH, OW, K, S, P;
};

typedef unsigned char u8x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(kBlock) void maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                             unsigned short* __restrict__ y,
                                                             unsigned char* __restrict__ idx, PoolGeom g) {
  const int c8 = g.C >> 3;
  const long long total = (long long)g.N * g.OH * g.OW * c8;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % c8) << 3;
    long long pix = i / c8;
    const int ow = (int)(pix % g.OW); pix /= g.OW;
    const int oh = (int)(pix % g.OH);
    const int n = (int)(pix / g.OH);
    float best[8];
    unsigned char arg[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; arg[k] = 0; }
    for (int kh = 0; kh < g.K; ++kh) {
      const int ih = oh * g.S - g.P + kh;
      if (ih < 0 || ih >= g.H) continue;
      for (int kw = 0; kw < g.K; ++kw) {
        const int iw = ow * g.S - g.P + kw;
        if (iw < 0 || iw >= g.W) continue;
        const u16x8 v = *reinterpret_cast<const u16x8*>(x + (((long long)n * g.H + ih) * g.W + iw) * g.C + c);
        const unsigned char tap = (unsigned char)(kh * g.K + kw);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float f = bf16_to_f32(v[k]);
          if (f > best[k] || (f != f && best[k] == best[k])) { best[k] = f; arg[k] = tap; }   // NaN propagates
        }
      }
    }
    u8x8 a;
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = arg[k];
    store8_bf16(y + i * 8, best);
    *reinterpret_cast<u8x8*>(idx + i * 8) = a;
  }
}

__global__ __launch_bounds__(kBlock) void maxpool_bwd_kernel(const unsigned short* __restrict__ gy,
                                                             const unsigned char* __restrict__ idx,
                                                             unsigned short* __restrict__ gx, PoolGeom g) {
  const int c8 = g.C >> 3;
  const long long total = (long long)g.N * g.H * g.W * c8;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % c8) << 3;
    long long pix = i / c8;
    const int iw = (int)(pix % g.W); pix /= g.W;
    const int ih = (int)(pix % g.H);
    const int n = (int)(pix / g.H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // windows (oh, ow) with oh*S - P <= ih <= oh*S - P + K - 1
    const int oh_hi = min(g.OH - 1, (ih + g.P) / g.S);
    const int ow_hi = min(g.OW - 1, (iw + g.P) / g.S);
    const int oh_lo = max(0, (ih + g.P - g.K + g.S) / g.S);      // ceil((ih + P - K + 1) / S) for the non-negative case
    const int ow_lo = max(0, (iw + g.P - g.K + g.S) / g.S);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int kh = ih - (oh * g.S - g.P);
      if (kh < 0 || kh >= g.K) continue;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int kw = iw - (ow * g.S - g.P);
        if (kw < 0 || kw >= g.K) continue;
        const long long o = (((long long)n * g.OH + oh) * g.OW + ow) * g.C + c;
        const u8x8 a = *reinterpret_cast<const u8x8*>(idx + o);
        const u16x8 v = *reinterpret_cast<const u16x8*>(gy + o);
        const unsigned char tap = (unsigned char)(kh * g.K + kw);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += a[k] == tap ? bf16_to_f32(v[k]) : 0.f;
      }
    }
    store8_bf16(gx + i * 8, acc);
  }
}

bool geom_ok(const PoolGeom& g) {
  return g.N > 0 && g.H > 0 && g.W > 0 && g.C > 0 && g.C % 8 == 0 && g.K >= 1 && g.K <= 15 && g.S >= 1 && g.P >= 0 &&
         g.P < g.K && g.OH == (g.H + 2 * g.P - g.K) / g.S + 1 && g.OW == (g.W + 2 * g.P - g.K) / g.S + 1;
}

}  // namespace

// x [N,H,W,C] bf16 -> y [N,OH,OW,C] bf16, idx [N,OH,OW,C] uint8 (winning tap kh*K + kw)
DTF_API int dtf_maxpool_fwd(const void* x, void* y, void* idx, int N, int H, int W, int C, int OH, int OW, int K, int S,
                            int P, hipStream_t stream) {
  const PoolGeom g = {N, H, W, C, OH, OW, K, S, P};
  if (!geom_ok(g)) return -1;
  const long long total = (long long)N * OH * OW * (C / 8);
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_stream(total, kBlock)), dim3(kBlock), 0, stream,
                     (const unsigned short*)x, (unsigned short*)y, (unsigned char*)idx, g);
  return (int)hipGetLastError();
}

DTF_API int dtf_maxpool_bwd(const void* gy, const void* idx, void* gx, int N, int H, int W, int C, int OH, int OW, int K,
                            int S, int P, hipStream_t stream) {
  const PoolGeom g = {N, H, W, C, OH, OW, K, S, P};
  if (!geom_ok(g)) return -1;
  const long long total = (long long)N * H * W * (C / 8);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_stream(total, kBlock)), dim3(kBlock), 0, stream,
                     (const unsigned short*)gy, (const unsigned char*)idx, (unsigned short*)gx, g);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------- global average pool (the ResNet head)
// y[n][c] = mean over the HW pixels of x[n][.][c], fp32 accumulation; a thread owns 8 channels of one image and walks
// the pixels (consecutive lanes read consecutive channels, so every access of a wave is one contiguous row piece).
// Backward is one streaming pass: dx[n][p][c] = dy[n][c] / HW.
namespace {

__global__ __launch_bounds__(kBlock) void gap_fwd_kernel(const unsigned short* __restrict__ x,
                                                         unsigned short* __restrict__ y, int N, int HW, int C,
                                                         float inv) {
  const int c8 = C >> 3;
  const long long total = (long long)N * c8;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long n = i / c8;
    const int c = (int)(i - n * c8) << 3;
    const unsigned short* p = x + (n * HW) * C + c;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = 0; q < HW; ++q, p += C) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += bf16_to_f32(v[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= inv;
    store8_bf16(y + n * C + c, acc);
  }
}

__global__ __launch_bounds__(kBlock) void gap_bwd_kernel(const unsigned short* __restrict__ dy,
                                                         unsigned short* __restrict__ dx, long long total, int HW,
                                                         int C, float inv) {
  const int c8 = C >> 3;
  const long long per_image = (long long)HW * c8;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long long)gridDim.x * kBlock) {
    const long long n = i / per_image;
    const int c = (int)(i % c8) << 3;
    const u16x8 v = *reinterpret_cast<const u16x8*>(dy + n * C + c);
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = bf16_to_f32(v[k]) * inv;
    store8_bf16(dx + i * 8, g);
  }
}

}  // namespace

DTF_API int dtf_gap_fwd(const void* x, void* y, int N, int HW, int C, hipStream_t stream) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  if (C % 8) return -1;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(grid_stream((long long)N * (C / 8), kBlock)), dim3(kBlock), 0, stream,
                     (const unsigned short*)x, (unsigned short*)y, N, HW, C, 1.f / (float)HW);
  return (int)hipGetLastError();
}

DTF_API int dtf_gap_bwd(const void* dy, void* dx, int N, int HW, int C, hipStream_t stream) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  if (C % 8) return -1;
  const long long total = (long long)N * HW * (C / 8);
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_stream(total, kBlock)), dim3(kBlock), 0, stream,
                     (const unsigned short*)dy, (unsigned short*)dx, total, HW, C, 1.f / (float)HW);
  return (int)hipGetLastError();
}
