// Fused softmax + sparse categorical cross-entropy, forward and backward (gfx950).
//
// Replaces the Dense->softmax->sparse_categorical_crossentropy chain of the
// reference loss (reference centralized/client.py:11-13, server.py:13-15):
// the kernel consumes raw logits, never materialises the probabilities, and
// returns the per-example loss vector (no reduction, as the reference does)
// together with the row log-sum-exp that the backward pass reuses.
//
//   forward : loss[b] = lse[b] - x[b, y[b]]     (0 when y[b] < 0 = ignored row)
//             optional correct[b] = (argmax_j x[b, j] == y[b])
//   backward: dx[b, j] = g[b] * (exp(x[b, j] - lse[b]) - [j == y[b]])
//
// One row is owned by LPR lanes (8 for C <= 64, else a whole 64-lane wave);
// lanes walk the row in 16-byte vectors after peeling the unaligned head, so
// rows whose length is not a multiple of the vector width (BERT's 30522)
// still stream at full width. Max and sum are tracked online, one rescale per
// vector.
#include "dtf_common.h"

using namespace dtf;

template <typename T> struct Elt;
template <> struct Elt<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ void ldv(const float* p, float (&o)[4]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  }
  static __device__ __forceinline__ void stv(float* p, const float (&o)[4]) {
    f32x4 a; a[0] = o[0]; a[1] = o[1]; a[2] = o[2]; a[3] = o[3];
    *reinterpret_cast<f32x4*>(p) = a;
  }
};
template <> struct Elt<unsigned short> {
  static constexpr int V = 8;
  static __device__ __forceinline__ float ld(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(unsigned short* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ void ldv(const unsigned short* p, float (&o)[8]) {
    u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
  }
  static __device__ __forceinline__ void stv(unsigned short* p, const float (&o)[8]) {
    u16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f32_to_bf16(o[i]);
    *reinterpret_cast<u16x8*>(p) = a;
  }
};

template <int LPR>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Online (max, sum-exp, argmax) update with one element.
__device__ __forceinline__ void online1(float x, int idx, float& m, float& s, int& am) {
  if (x > m) { s = s * __expf(m - x) + 1.f; m = x; am = idx; }
  else s += __expf(x - m);
}

template <typename T, typename L, int LPR>
__global__ __launch_bounds__(kBlock) void softmax_ce_fwd_kernel(const T* __restrict__ x, const L* __restrict__ y,
                                                                float* __restrict__ loss, float* __restrict__ lse_out,
                                                                unsigned char* __restrict__ correct, int B, int C,
                                                                long long ld) {
  constexpr int V = Elt<T>::V;
  constexpr int RPB = kBlock / LPR;  // rows per block
  const int sub = threadIdx.x % LPR;
  const long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
  const bool live = row < B;
  const T* xr = x + (live ? row : 0) * ld;

  float m = -INFINITY, s = 0.f;
  int am = 0;
  if (live) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(xr);
    int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
    if (head > C) head = C;
    for (int i = sub; i < head; i += LPR) online1(Elt<T>::ld(xr + i), i, m, s, am);
    const int nvec = (C - head) / V;
    const T* body = xr + head;
    for (int j = sub; j < nvec; j += LPR) {
      float v[V];
      Elt<T>::ldv(body + j * V, v);
      float cm = v[0]; int ci = 0;
#pragma unroll
      for (int k = 1; k < V; ++k) if (v[k] > cm) { cm = v[k]; ci = k; }
      if (cm > m) { s *= __expf(m - cm); m = cm; am = head + j * V + ci; }
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < V; ++k) acc += __expf(v[k] - m);
      s += acc;
    }
    for (int i = head + nvec * V + sub; i < C; i += LPR) online1(Elt<T>::ld(xr + i), i, m, s, am);
  }
  // merge the LPR partial (m, s, am) triples; ties resolve to the lowest index
  const float gm = group_max<LPR>(m);
  const float sc = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
  const float gs = group_sum<LPR>(sc);
  int cand = (m == gm) ? am : 0x7fffffff;
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  if (live && sub == 0) {
    const float lse = gm + __logf(gs);
    const long long lab = (long long)y[row];
    float l = 0.f;
    if (lab >= 0 && lab < C) l = lse - Elt<T>::ld(xr + lab);
    loss[row] = l;
    lse_out[row] = lse;
    if (correct) correct[row] = (unsigned char)(cand == (int)lab);
  }
}

template <typename T, typename L, int LPR>
__global__ __launch_bounds__(kBlock) void softmax_ce_bwd_kernel(const T* __restrict__ x, const L* __restrict__ y,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ gout, T* __restrict__ dx,
                                                                int B, int C, long long ld) {
  constexpr int V = Elt<T>::V;
  constexpr int RPB = kBlock / LPR;
  const int sub = threadIdx.x % LPR;
  const long long row = (long long)blockIdx.x * RPB + threadIdx.x / LPR;
  if (row >= B) return;
  const T* xr = x + row * ld;
  T* dr = dx + row * ld;
  const long long lab = (long long)y[row];
  const bool ignored = !(lab >= 0 && lab < C);
  const float g = ignored ? 0.f : gout[row];
  const float l = lse[row];
  const int labi = ignored ? -1 : (int)lab;

  const uintptr_t addr = reinterpret_cast<uintptr_t>(xr);
  // x and dx rows share the same element offset, so they share alignment when
  // the two base pointers are equally aligned (checked on the host).
  int head = (int)(((16 - (addr & 15)) & 15) / sizeof(T));
  if (head > C) head = C;
  for (int i = sub; i < head; i += LPR)
    Elt<T>::st(dr + i, g * (__expf(Elt<T>::ld(xr + i) - l) - (i == labi ? 1.f : 0.f)));
  const int nvec = (C - head) / V;
  for (int j = sub; j < nvec; j += LPR) {
    float v[V];
    const int base = head + j * V;
    Elt<T>::ldv(xr + base, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = g * (__expf(v[k] - l) - ((base + k) == labi ? 1.f : 0.f));
    Elt<T>::stv(dr + base, v);
  }
  for (int i = head + nvec * V + sub; i < C; i += LPR)
    Elt<T>::st(dr + i, g * (__expf(Elt<T>::ld(xr + i) - l) - (i == labi ? 1.f : 0.f)));
  // padded rows (ld > C): the padding of the gradient is zeroed, so that a GEMM may read whole 16-byte chunks
  for (long long i = C + sub; i < ld; i += LPR) Elt<T>::st(dr + i, 0.f);
}

// x_dtype: 0 fp32, 1 bf16. label_dtype: 0 int32, 1 int64. ld: row stride of x (and dx) in elements, >= C.
DTF_API int dtf_softmax_ce_fwd_ld(const void* x, const void* y, float* loss, float* lse, unsigned char* correct,
                                  int B, int C, long long ld, int x_dtype, int label_dtype, hipStream_t stream) {
  if (B <= 0 || C <= 0) return 0;
  if (ld < C) return -1;
#define FWD(T, L, LPR)                                                                                        \
  hipLaunchKernelGGL((softmax_ce_fwd_kernel<T, L, LPR>), dim3((B + (kBlock / LPR) - 1) / (kBlock / LPR)),    \
                     dim3(kBlock), 0, stream, (const T*)x, (const L*)y, loss, lse, correct, B, C, ld)
  const bool small = C <= 64;
  if (x_dtype == 0 && label_dtype == 0) { if (small) FWD(float, int, 8); else FWD(float, int, 64); }
  else if (x_dtype == 0) { if (small) FWD(float, long long, 8); else FWD(float, long long, 64); }
  else if (label_dtype == 0) { if (small) FWD(unsigned short, int, 8); else FWD(unsigned short, int, 64); }
  else { if (small) FWD(unsigned short, long long, 8); else FWD(unsigned short, long long, 64); }
#undef FWD
  return (int)hipGetLastError();
}

DTF_API int dtf_softmax_ce_fwd(const void* x, const void* y, float* loss, float* lse, unsigned char* correct,
                               int B, int C, int x_dtype, int label_dtype, hipStream_t stream) {
  return dtf_softmax_ce_fwd_ld(x, y, loss, lse, correct, B, C, C, x_dtype, label_dtype, stream);
}

DTF_API int dtf_softmax_ce_bwd_ld(const void* x, const void* y, const float* lse, const float* gout, void* dx,
                                  int B, int C, long long ld, int x_dtype, int label_dtype, hipStream_t stream) {
  if (B <= 0 || C <= 0) return 0;
  if (ld < C) return -1;
  if ((reinterpret_cast<uintptr_t>(x) & 15) != (reinterpret_cast<uintptr_t>(dx) & 15)) return -2;
#define BWD(T, L, LPR)                                                                                        \
  hipLaunchKernelGGL((softmax_ce_bwd_kernel<T, L, LPR>), dim3((B + (kBlock / LPR) - 1) / (kBlock / LPR)),    \
                     dim3(kBlock), 0, stream, (const T*)x, (const L*)y, lse, gout, (T*)dx, B, C, ld)
  const bool small = C <= 64;
  if (x_dtype == 0 && label_dtype == 0) { if (small) BWD(float, int, 8); else BWD(float, int, 64); }
  else if (x_dtype == 0) { if (small) BWD(float, long long, 8); else BWD(float, long long, 64); }
  else if (label_dtype == 0) { if (small) BWD(unsigned short, int, 8); else BWD(unsigned short, int, 64); }
  else { if (small) BWD(unsigned short, long long, 8); else BWD(unsigned short, long long, 64); }
#undef BWD
  return (int)hipGetLastError();
}

DTF_API int dtf_softmax_ce_bwd(const void* x, const void* y, const float* lse, const float* gout, void* dx,
                               int B, int C, int x_dtype, int label_dtype, hipStream_t stream) {
  return dtf_softmax_ce_bwd_ld(x, y, lse, gout, dx, B, C, C, x_dtype, label_dtype, stream);
}
