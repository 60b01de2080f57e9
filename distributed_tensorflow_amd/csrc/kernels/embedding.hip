// Embedding kernels for gfx950: row gather (with the BERT word + position + type sum fused),
// dense scatter-add of row gradients, and the sparse path for very large tables -- segment
// sums over id-sorted gradients followed by an optimizer update of only the touched rows.
//
// The reference has no embedding layer; these serve the BERT configuration (vocabulary
// 30522 x 768) and the Wide&Deep configuration, whose tables are row-sharded across the
// parameter-server shards (each GPU holds rows [r*V/W, (r+1)*V/W) in its own HBM).
//
// A row is handled by LPR lanes moving 16 bytes each; rows are D elements, D % 8 == 0 for
// bf16 tables and D % 4 == 0 for fp32 tables.
#include "dtf_common.h"

using namespace dtf;

template <typename T> struct Row;
template <> struct Row<unsigned short> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void ld(const unsigned short* p, float (&o)[8]) {
    u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
  }
  static __device__ __forceinline__ void st(unsigned short* p, const float (&o)[8]) {
    u16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f32_to_bf16(o[i]);
    *reinterpret_cast<u16x8*>(p) = a;
  }
};
template <> struct Row<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
  }
  static __device__ __forceinline__ void st(float* p, const float (&o)[8]) {
    f32x4 a; a[0] = o[0]; a[1] = o[1]; a[2] = o[2]; a[3] = o[3];
    *reinterpret_cast<f32x4*>(p) = a;
  }
};

// out[i, :] = table[ids[i] - row_base, :] (+ pos[i % period, :]) (+ typ[tt[i], :]); ids outside
// [row_base, row_base + n_rows) produce zeros (the row lives on another shard).
template <typename T, typename O>
__global__ __launch_bounds__(kBlock) void emb_gather_kernel(const T* __restrict__ table, const long long* __restrict__ ids,
                                                            const T* __restrict__ pos, int period,
                                                            const T* __restrict__ typ, const long long* __restrict__ tt,
                                                            O* __restrict__ out, long long n, int D, long long row_base,
                                                            long long n_rows) {
  constexpr int V = Row<T>::V;
  const int cpr = D / V;                                 // chunks per row
  const long long total = n * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * V;
    const long long id = ids[r] - row_base;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (id >= 0 && id < n_rows) Row<T>::ld(table + id * D + c, v);
    if (pos) {
      float p[8];
      Row<T>::ld(pos + (r % period) * D + c, p);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] += p[k];
    }
    if (typ) {
      float p[8];
      Row<T>::ld(typ + tt[r] * D + c, p);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] += p[k];
    }
    if (sizeof(O) == sizeof(T)) {
      Row<T>::st(reinterpret_cast<T*>(out) + r * D + c, v);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) reinterpret_cast<unsigned short*>(out)[r * D + c + k] = f32_to_bf16(v[k]);
    }
  }
}

// dtable[ids[i] - row_base, :] += g[i, :]  (fp32 accumulation with hardware float atomics)
template <typename G>
__global__ __launch_bounds__(kBlock) void emb_scatter_add_kernel(const G* __restrict__ g, const long long* __restrict__ ids,
                                                                 float* __restrict__ dtable, long long n, int D,
                                                                 long long row_base, long long n_rows, int period_rows) {
  constexpr int V = Row<G>::V;
  const int cpr = D / V;
  const long long total = n * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * V;
    long long id = ids ? ids[r] - row_base : (r % period_rows);
    if (id < 0 || id >= n_rows) continue;
    float v[8];
    Row<G>::ld(g + r * D + c, v);
    float* dst = dtable + id * D + c;
#pragma unroll
    for (int k = 0; k < V; ++k) atomicAdd(dst + k, v[k]);
  }
}

// Rows sorted by id: segment u covers sorted positions [seg[u], seg[u+1]); perm maps a sorted
// position to its row of g. out[u, :] = sum of those rows, in sorted order (reproducible).
template <typename G>
__global__ __launch_bounds__(kBlock) void emb_segment_sum_kernel(const G* __restrict__ g, const long long* __restrict__ perm,
                                                                 const long long* __restrict__ seg, float* __restrict__ out,
                                                                 long long n_seg, int D) {
  constexpr int V = Row<G>::V;
  const int cpr = D / V;
  const long long total = n_seg * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long u = i / cpr;
    const int c = (int)(i - u * cpr) * V;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long j = seg[u]; j < seg[u + 1]; ++j) {
      float v[8];
      Row<G>::ld(g + perm[j] * D + c, v);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
    float* dst = out + u * D + c;
#pragma unroll
    for (int k = 0; k < V; ++k) dst[k] = acc[k];
  }
}

// Dense table gradient from rows SORTED by id (sorted_ids ascending, perm[j] = row of g at sorted
// position j). Equal ids are adjacent, so a run of them is summed in registers by one thread group
// and written with a plain store: no atomic per element, which is what bounds the unsorted
// scatter-add (about one atomic per gradient element, contended when an id repeats). A run is cut
// at every kRunCut-th sorted position so that a very frequent id (padding, token type) is still
// summed in parallel; only the pieces of such cut runs are combined with atomics.
constexpr int kRunCut = 128;

template <typename G>
__global__ __launch_bounds__(kBlock) void emb_sorted_grad_kernel(const G* __restrict__ g,
                                                                 const long long* __restrict__ sorted_ids,
                                                                 const long long* __restrict__ perm,
                                                                 float* __restrict__ dtable, long long n, int D,
                                                                 long long row_base, long long n_rows) {
  constexpr int V = Row<G>::V;
  const int cpr = D / V;
  const long long total = n * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long j = i / cpr;
    const int c = (int)(i - j * cpr) * V;
    const long long id = sorted_ids[j];
    const bool head = j == 0 || sorted_ids[j - 1] != id;
    if (!head && (j % kRunCut) != 0) continue;            // an earlier position owns this row
    const long long loc = id - row_base;
    if (loc < 0 || loc >= n_rows) continue;
    long long end = (j / kRunCut + 1) * kRunCut;
    if (end > n) end = n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long k = j;
    for (; k < end && sorted_ids[k] == id; ++k) {
      float v[8];
      Row<G>::ld(g + perm[k] * D + c, v);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] += v[e];
    }
    const bool whole = head && (k == n || sorted_ids[k] != id);
    float* dst = dtable + loc * D + c;
    if (whole) {
#pragma unroll
      for (int e = 0; e < V; ++e) dst[e] = acc[e];
    } else {
#pragma unroll
      for (int e = 0; e < V; ++e) atomicAdd(dst + e, acc[e]);
    }
  }
}

// Optimizer update of the touched rows only. kind 0: SGD (p -= lr*g); 1: Adagrad (v += g^2,
// p -= lr*g/(sqrt(v)+eps)); 2: Adam with per-row lazy moments (bias correction from hyper[1..2]).
// hyper = {lr, bc1, bc2, eps, beta1, beta2, grad_scale, _}.
__global__ __launch_bounds__(kBlock) void emb_sparse_apply_kernel(float* __restrict__ table, float* __restrict__ m,
                                                                  float* __restrict__ v, const float* __restrict__ grad,
                                                                  const long long* __restrict__ uids,
                                                                  const float* __restrict__ hyper, long long n_u, int D,
                                                                  long long row_base, long long n_rows, int kind) {
  const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2], eps = hyper[3], b1 = hyper[4], b2 = hyper[5], gs = hyper[6];
  const int cpr = D / 4;
  const long long total = n_u * cpr;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const long long u = i / cpr;
    const int c = (int)(i - u * cpr) * 4;
    const long long id = uids[u] - row_base;
    if (id < 0 || id >= n_rows) continue;
    f32x4 g = *reinterpret_cast<const f32x4*>(grad + u * D + c);
    f32x4 p = *reinterpret_cast<f32x4*>(table + id * D + c);
    if (kind == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) p[k] -= lr * gs * g[k];
    } else if (kind == 1) {
      f32x4 vv = *reinterpret_cast<f32x4*>(v + id * D + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float gg = gs * g[k]; vv[k] = fmaf(gg, gg, vv[k]); p[k] -= lr * gg / (sqrtf(vv[k]) + eps); }
      *reinterpret_cast<f32x4*>(v + id * D + c) = vv;
    } else {
      f32x4 mm = *reinterpret_cast<f32x4*>(m + id * D + c);
      f32x4 vv = *reinterpret_cast<f32x4*>(v + id * D + c);
      const float step = lr * sqrtf(bc2) / bc1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = gs * g[k];
        mm[k] = fmaf(b1, mm[k], (1.f - b1) * gg);
        vv[k] = fmaf(b2, vv[k], (1.f - b2) * gg * gg);
        p[k] -= step * mm[k] / (sqrtf(vv[k]) + eps);
      }
      *reinterpret_cast<f32x4*>(m + id * D + c) = mm;
      *reinterpret_cast<f32x4*>(v + id * D + c) = vv;
    }
    *reinterpret_cast<f32x4*>(table + id * D + c) = p;
  }
}

// dtype codes: 0 fp32, 1 bf16
DTF_API int dtf_emb_gather(const void* table, const long long* ids, const void* pos, int period, const void* typ,
                           const long long* tt, void* out, long long n, int D, long long row_base, long long n_rows,
                           int table_dtype, int out_dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  if (table_dtype == 1) {
    if (D % 8 || out_dtype != 1) return -1;
    hipLaunchKernelGGL((emb_gather_kernel<unsigned short, unsigned short>), dim3(grid_stream(n * (D / 8), kBlock)),
                       dim3(kBlock), 0, stream, (const unsigned short*)table, ids, (const unsigned short*)pos, period,
                       (const unsigned short*)typ, tt, (unsigned short*)out, n, D, row_base, n_rows);
  } else {
    if (D % 4) return -1;
    if (out_dtype == 0)
      hipLaunchKernelGGL((emb_gather_kernel<float, float>), dim3(grid_stream(n * (D / 4), kBlock)), dim3(kBlock), 0, stream,
                         (const float*)table, ids, (const float*)pos, period, (const float*)typ, tt, (float*)out, n, D,
                         row_base, n_rows);
    else
      hipLaunchKernelGGL((emb_gather_kernel<float, unsigned short>), dim3(grid_stream(n * (D / 4), kBlock)), dim3(kBlock), 0,
                         stream, (const float*)table, ids, (const float*)pos, period, (const float*)typ, tt,
                         (unsigned short*)out, n, D, row_base, n_rows);
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_emb_scatter_add(const void* g, const long long* ids, float* dtable, long long n, int D,
                                long long row_base, long long n_rows, int period_rows, int g_dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  if (g_dtype == 1) {
    if (D % 8) return -1;
    hipLaunchKernelGGL((emb_scatter_add_kernel<unsigned short>), dim3(grid_stream(n * (D / 8), kBlock)), dim3(kBlock), 0,
                       stream, (const unsigned short*)g, ids, dtable, n, D, row_base, n_rows, period_rows);
  } else {
    if (D % 4) return -1;
    hipLaunchKernelGGL((emb_scatter_add_kernel<float>), dim3(grid_stream(n * (D / 4), kBlock)), dim3(kBlock), 0, stream,
                       (const float*)g, ids, dtable, n, D, row_base, n_rows, period_rows);
  }
  return (int)hipGetLastError();
}

// dtable (fp32 [n_rows, D]) must be zero on entry; sorted_ids / perm as produced by a stable sort of ids.
DTF_API int dtf_emb_sorted_grad(const void* g, const long long* sorted_ids, const long long* perm, float* dtable,
                                long long n, int D, long long row_base, long long n_rows, int g_dtype,
                                hipStream_t stream) {
  if (n <= 0) return 0;
  if (g_dtype == 1) {
    if (D % 8) return -1;
    hipLaunchKernelGGL((emb_sorted_grad_kernel<unsigned short>), dim3(grid_stream(n * (D / 8), kBlock)), dim3(kBlock), 0,
                       stream, (const unsigned short*)g, sorted_ids, perm, dtable, n, D, row_base, n_rows);
  } else {
    if (D % 4) return -1;
    hipLaunchKernelGGL((emb_sorted_grad_kernel<float>), dim3(grid_stream(n * (D / 4), kBlock)), dim3(kBlock), 0, stream,
                       (const float*)g, sorted_ids, perm, dtable, n, D, row_base, n_rows);
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_emb_segment_sum(const void* g, const long long* perm, const long long* seg, float* out, long long n_seg,
                                int D, int g_dtype, hipStream_t stream) {
  if (n_seg <= 0) return 0;
  if (g_dtype == 1) {
    if (D % 8) return -1;
    hipLaunchKernelGGL((emb_segment_sum_kernel<unsigned short>), dim3(grid_stream(n_seg * (D / 8), kBlock)), dim3(kBlock),
                       0, stream, (const unsigned short*)g, perm, seg, out, n_seg, D);
  } else {
    if (D % 4) return -1;
    hipLaunchKernelGGL((emb_segment_sum_kernel<float>), dim3(grid_stream(n_seg * (D / 4), kBlock)), dim3(kBlock), 0, stream,
                       (const float*)g, perm, seg, out, n_seg, D);
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_emb_sparse_apply(float* table, float* m, float* v, const float* grad, const long long* uids,
                                 const float* hyper, long long n_u, int D, long long row_base, long long n_rows, int kind,
                                 hipStream_t stream) {
  if (n_u <= 0) return 0;
  if (D % 4) return -1;
  hipLaunchKernelGGL(emb_sparse_apply_kernel, dim3(grid_stream(n_u * (D / 4), kBlock)), dim3(kBlock), 0, stream, table, m, v,
                     grad, uids, hyper, n_u, D, row_base, n_rows, kind);
  return (int)hipGetLastError();
}
