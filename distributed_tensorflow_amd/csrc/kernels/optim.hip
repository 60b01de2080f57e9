// Fused optimizer apply kernels over flat parameter arenas (gfx950).
//
// The parameter server of the reference applies Keras Adam one variable at a
// time under a Python lock (reference centralized/server.py:92,100). Here the
// whole parameter shard is one contiguous fp32 arena, and one launch updates
// param, first and second moment (and an optional bf16 compute copy) at
// 16 bytes per lane. The kernels are pure streaming: per element they read
// 16 B (p, g, m, v) and write 12 B (+2 B for the bf16 copy), so the target is
// the HBM streaming rate.
//
// Hyper-parameters live in a small device-resident block (`OptHyper`) so that
// a captured hipGraph replays with the current learning rate / step instead
// of values frozen at capture time. `dtf_opt_tick` advances the step and the
// bias corrections on the device.
#include "dtf_common.h"

using namespace dtf;

// Layout shared with python (ops/optim.py): 16 floats.
struct OptHyper {
  float lr;            // 0
  float beta1;         // 1
  float beta2;         // 2
  float eps;           // 3
  float weight_decay;  // 4
  float grad_scale;    // 5  multiplies every gradient (1/world, 1/loss_scale)
  float momentum;      // 6
  float step;          // 7
  float bc1;           // 8  1 - beta1^step
  float bc2;           // 9  1 - beta2^step
  float pad[6];
};

__global__ void opt_tick_kernel(OptHyper* h) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = h->step + 1.f;
    h->step = s;
    h->bc1 = 1.f - powf(h->beta1, s);
    h->bc2 = 1.f - powf(h->beta2, s);
  }
}

template <typename G>
__device__ __forceinline__ f32x4 load_grad4(const G* g, long long i4);

template <>
__device__ __forceinline__ f32x4 load_grad4<float>(const float* g, long long i4) {
  return reinterpret_cast<const f32x4*>(g)[i4];
}

template <>
__device__ __forceinline__ f32x4 load_grad4<unsigned short>(const unsigned short* g, long long i4) {
  u16x4 r = reinterpret_cast<const u16x4*>(g)[i4];
  f32x4 o;
  o[0] = bf16_to_f32(r[0]); o[1] = bf16_to_f32(r[1]);
  o[2] = bf16_to_f32(r[2]); o[3] = bf16_to_f32(r[3]);
  return o;
}

template <typename G>
__device__ __forceinline__ float load_grad1(const G* g, long long i);
template <>
__device__ __forceinline__ float load_grad1<float>(const float* g, long long i) { return g[i]; }
template <>
__device__ __forceinline__ float load_grad1<unsigned short>(const unsigned short* g, long long i) {
  return bf16_to_f32(g[i]);
}

// mode bit 0: 0 = Keras formula  p -= lr*sqrt(bc2)/bc1 * m / (sqrt(v) + eps)
//             1 = torch formula  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// mode bit 1: decoupled weight decay (AdamW) instead of L2-in-gradient.
struct AdamCoef {
  float b1, b2, omb1, omb2, eps, wd, gs, step_size, inv_sqrt_bc2, lr;
  int torch_form, decoupled;
};

__device__ __forceinline__ AdamCoef adam_coef(const OptHyper* h, int mode) {
  AdamCoef c;
  c.b1 = h->beta1; c.b2 = h->beta2; c.omb1 = 1.f - c.b1; c.omb2 = 1.f - c.b2;
  c.eps = h->eps; c.wd = h->weight_decay; c.gs = h->grad_scale; c.lr = h->lr;
  c.torch_form = mode & 1; c.decoupled = (mode >> 1) & 1;
  const float bc1 = h->bc1, bc2 = h->bc2;
  if (c.torch_form) {
    c.step_size = h->lr / bc1;
    c.inv_sqrt_bc2 = rsqrtf(bc2);
  } else {
    c.step_size = h->lr * sqrtf(bc2) / bc1;
    c.inv_sqrt_bc2 = 1.f;
  }
  return c;
}

__device__ __forceinline__ void adam_one(const AdamCoef& c, float& p, float g, float& m, float& v) {
  g *= c.gs;
  if (!c.decoupled) g = fmaf(c.wd, p, g);
  m = fmaf(c.b1, m, c.omb1 * g);
  v = fmaf(c.b2, v, c.omb2 * g * g);
  const float denom = sqrtf(v) * c.inv_sqrt_bc2 + c.eps;
  float np = p - c.step_size * (m / denom);
  if (c.decoupled) np -= c.lr * c.wd * p;
  p = np;
}

template <typename G, bool SHADOW>
__global__ __launch_bounds__(kBlock) void adam_kernel(float* __restrict__ p, const G* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      unsigned short* __restrict__ shadow,
                                                      const OptHyper* __restrict__ hyp, long long n, int mode) {
  const AdamCoef c = adam_coef(hyp, mode);
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* m4 = reinterpret_cast<f32x4*>(m);
  f32x4* v4 = reinterpret_cast<f32x4*>(v);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = p4[i], mm = m4[i], vv = v4[i];
    f32x4 gg = load_grad4<G>(g, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = pp[k], b = mm[k], d = vv[k];
      adam_one(c, a, gg[k], b, d);
      pp[k] = a; mm[k] = b; vv[k] = d;
    }
    p4[i] = pp; m4[i] = mm; v4[i] = vv;
    if (SHADOW) {
      store4_bf16(shadow + 4 * i, pp[0], pp[1], pp[2], pp[3]);
    }
  }
  // scalar tail (n % 4 elements), handled by the first threads of block 0
  const long long tail0 = n4 << 2;
  if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
    const long long i = tail0 + threadIdx.x;
    float a = p[i], b = m[i], d = v[i];
    adam_one(c, a, load_grad1<G>(g, i), b, d);
    p[i] = a; m[i] = b; v[i] = d;
    if (SHADOW) shadow[i] = f32_to_bf16(a);
  }
}

// mode bit 0: 0 = Keras SGD   vel = mu*vel - lr*g ; p += vel   (nesterov: p += mu*vel - lr*g)
//             1 = torch SGD   buf = mu*buf + g    ; p -= lr*buf (nesterov: p -= lr*(g + mu*buf))
// mode bit 1: nesterov.
template <typename G, bool SHADOW>
__global__ __launch_bounds__(kBlock) void sgdm_kernel(float* __restrict__ p, const G* __restrict__ g,
                                                      float* __restrict__ vel,
                                                      unsigned short* __restrict__ shadow,
                                                      const OptHyper* __restrict__ hyp, long long n, int mode) {
  const float lr = hyp->lr, mu = hyp->momentum, wd = hyp->weight_decay, gs = hyp->grad_scale;
  const int torch_form = mode & 1, nesterov = (mode >> 1) & 1;
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * blockDim.x;
  f32x4* p4 = reinterpret_cast<f32x4*>(p);
  f32x4* v4 = reinterpret_cast<f32x4*>(vel);
  auto one = [&](float& pp, float gg, float& vv) {
    gg = fmaf(wd, pp, gg * gs);
    if (torch_form) {
      vv = fmaf(mu, vv, gg);
      pp -= lr * (nesterov ? fmaf(mu, vv, gg) : vv);
    } else {
      vv = fmaf(mu, vv, -lr * gg);
      pp += nesterov ? fmaf(mu, vv, -lr * gg) : vv;
    }
  };
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    f32x4 pp = p4[i], vv = v4[i];
    f32x4 gg = load_grad4<G>(g, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = pp[k], b = vv[k];
      one(a, gg[k], b);
      pp[k] = a; vv[k] = b;
    }
    p4[i] = pp; v4[i] = vv;
    if (SHADOW) {
      store4_bf16(shadow + 4 * i, pp[0], pp[1], pp[2], pp[3]);
    }
  }
  const long long tail0 = n4 << 2;
  if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
    const long long i = tail0 + threadIdx.x;
    float a = p[i], b = vel[i];
    one(a, load_grad1<G>(g, i), b);
    p[i] = a; vel[i] = b;
    if (SHADOW) shadow[i] = f32_to_bf16(a);
  }
}

DTF_API int dtf_opt_tick(void* hyper, hipStream_t stream) {
  hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(64), 0, stream, (OptHyper*)hyper);
  return (int)hipGetLastError();
}

// grad_dtype: 0 = fp32, 1 = bf16. shadow may be null.
DTF_API int dtf_adam_apply(void* p, const void* g, void* m, void* v, void* shadow, const void* hyper,
                           long long n, int grad_dtype, int mode, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_stream((n + 3) / 4, kBlock);
  const OptHyper* h = (const OptHyper*)hyper;
#define LAUNCH(G, S)                                                                              \
  hipLaunchKernelGGL((adam_kernel<G, S>), dim3(grid), dim3(kBlock), 0, stream, (float*)p,         \
                     (const G*)g, (float*)m, (float*)v, (unsigned short*)shadow, h, n, mode)
  if (grad_dtype == 0) { if (shadow) LAUNCH(float, true); else LAUNCH(float, false); }
  else { if (shadow) LAUNCH(unsigned short, true); else LAUNCH(unsigned short, false); }
#undef LAUNCH
  return (int)hipGetLastError();
}

DTF_API int dtf_sgdm_apply(void* p, const void* g, void* vel, void* shadow, const void* hyper,
                           long long n, int grad_dtype, int mode, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_stream((n + 3) / 4, kBlock);
  const OptHyper* h = (const OptHyper*)hyper;
#define LAUNCH(G, S)                                                                              \
  hipLaunchKernelGGL((sgdm_kernel<G, S>), dim3(grid), dim3(kBlock), 0, stream, (float*)p,         \
                     (const G*)g, (float*)vel, (unsigned short*)shadow, h, n, mode)
  if (grad_dtype == 0) { if (shadow) LAUNCH(float, true); else LAUNCH(float, false); }
  else { if (shadow) LAUNCH(unsigned short, true); else LAUNCH(unsigned short, false); }
#undef LAUNCH
  return (int)hipGetLastError();
}
