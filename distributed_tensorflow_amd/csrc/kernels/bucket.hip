// Gradient-bucket kernels (gfx950): multi-tensor flatten / unflatten with
// fused scale and dtype cast, flat scale/cast, and the n-ary reductions used
// by the hand-written all-reduce algorithms and the graph (gossip) strategy.
//
// The reference moves gradients as a pickled list of per-variable arrays
// (reference centralized/client.py:85, centralized/server.py:104). Here a
// bucket is one contiguous device buffer; one launch packs up to kMtMax
// tensors into it (pointer table passed by value in the kernel arguments, so
// nothing is uploaded and the launch is graph-capturable).
#include "dtf_common.h"

using namespace dtf;

constexpr int kMtMax = 48;           // tensors per launch
constexpr int kMtChunk = 8192;       // elements per block-chunk

struct MtArgs {
  void* tensor[kMtMax];              // per-tensor base pointer
  long long flat_off[kMtMax];        // element offset of the tensor inside the flat buffer
  long long numel[kMtMax];
  int chunk_start[kMtMax + 1];       // prefix sum of per-tensor chunk counts
  int n_tensors;
};

template <typename T> struct Vec8;
template <> struct Vec8<float> {
  static __device__ __forceinline__ void load(const float* p, float (&o)[8]) {
    f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  }
  static __device__ __forceinline__ void store(float* p, const float (&o)[8]) {
    f32x4 a, b;
    a[0] = o[0]; a[1] = o[1]; a[2] = o[2]; a[3] = o[3];
    b[0] = o[4]; b[1] = o[5]; b[2] = o[6]; b[3] = o[7];
    reinterpret_cast<f32x4*>(p)[0] = a; reinterpret_cast<f32x4*>(p)[1] = b;
  }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Vec8<unsigned short> {
  static __device__ __forceinline__ void load(const unsigned short* p, float (&o)[8]) {
    u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
  }
  static __device__ __forceinline__ void store(unsigned short* p, const float (&o)[8]) {
    store8_bf16(p, o);
  }
  static __device__ __forceinline__ float ld1(const unsigned short* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st1(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

// Copies `n` elements src -> dst with scale; vector path when both are 16-B
// (bf16) / 32-B (f32 pair of 16-B) aligned, scalar otherwise.
template <typename S, typename D>
__device__ __forceinline__ void copy_scale_chunk(const S* __restrict__ src, D* __restrict__ dst, long long n,
                                                 float scale) {
  const bool aligned = ((reinterpret_cast<uintptr_t>(src) & 15) == 0) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  if (aligned) {
    const long long n8 = n >> 3;
    for (long long i = threadIdx.x; i < n8; i += blockDim.x) {
      float v[8];
      Vec8<S>::load(src + i * 8, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= scale;
      Vec8<D>::store(dst + i * 8, v);
    }
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x)
      Vec8<D>::st1(dst + i, Vec8<S>::ld1(src + i) * scale);
  } else {
    for (long long i = threadIdx.x; i < n; i += blockDim.x)
      Vec8<D>::st1(dst + i, Vec8<S>::ld1(src + i) * scale);
  }
}

// DIR 0: tensors -> flat (flatten); DIR 1: flat -> tensors (unflatten).
template <typename TT, typename FT, int DIR>
__global__ __launch_bounds__(kBlock) void mt_copy_kernel(MtArgs a, FT* __restrict__ flat, const float* scale_ptr,
                                                         float scale_imm) {
  const int b = blockIdx.x;
  int t = 0;
  while (t + 1 < a.n_tensors && a.chunk_start[t + 1] <= b) ++t;
  const long long c0 = (long long)(b - a.chunk_start[t]) * kMtChunk;
  long long n = a.numel[t] - c0;
  if (n <= 0) return;
  if (n > kMtChunk) n = kMtChunk;
  const float scale = scale_ptr ? scale_imm * (*scale_ptr) : scale_imm;
  TT* tp = reinterpret_cast<TT*>(a.tensor[t]) + c0;
  FT* fp = flat + a.flat_off[t] + c0;
  if (DIR == 0) copy_scale_chunk<TT, FT>(tp, fp, n, scale);
  else copy_scale_chunk<FT, TT>(fp, tp, n, scale);
}

template <typename S, typename D>
__global__ __launch_bounds__(kBlock) void flat_scale_cast_kernel(const S* __restrict__ src, D* __restrict__ dst,
                                                                 long long n, const float* scale_ptr, float scale_imm) {
  const float scale = scale_ptr ? scale_imm * (*scale_ptr) : scale_imm;
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float v[8];
    Vec8<S>::load(src + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= scale;
    Vec8<D>::store(dst + i * 8, v);
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x)
      Vec8<D>::st1(dst + i, Vec8<S>::ld1(src + i) * scale);
}

// dst = scale * (w0 * dst_in + sum_k w[k] * src[k]); up to 8 sources. Used for
//   * ring / tree / direct all-reduce partial sums (weights 1),
//   * gossip averaging over graph neighbours (weights 1/(d+1)).
constexpr int kNaryMax = 8;
struct NaryArgs {
  const void* src[kNaryMax];
  float w[kNaryMax];
  int n_src;
};

template <typename T>
__global__ __launch_bounds__(kBlock) void nary_axpy_kernel(T* __restrict__ dst, NaryArgs a, float w_dst, long long n) {
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float acc[8];
    Vec8<T>::load(dst + i * 8, acc);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= w_dst;
    for (int s = 0; s < a.n_src; ++s) {
      float v[8];
      Vec8<T>::load(reinterpret_cast<const T*>(a.src[s]) + i * 8, v);
      const float w = a.w[s];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(w, v[k], acc[k]);
    }
    Vec8<T>::store(dst + i * 8, acc);
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
      float acc = Vec8<T>::ld1(dst + i) * w_dst;
      for (int s = 0; s < a.n_src; ++s) acc = fmaf(a.w[s], Vec8<T>::ld1(reinterpret_cast<const T*>(a.src[s]) + i), acc);
      Vec8<T>::st1(dst + i, acc);
    }
}

// Sum of squares of a flat buffer into out[0] (fp32, atomically accumulated;
// caller zeroes out). Used for gradient-norm clipping and shard checksums.
template <typename T>
__global__ __launch_bounds__(kBlock) void sumsq_kernel(const T* __restrict__ x, long long n, float* out) {
  __shared__ float scratch[16];
  float acc = 0.f;
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
    float v[8];
    Vec8<T>::load(x + i * 8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = fmaf(v[k], v[k], acc);
  }
  if (blockIdx.x == 0)
    for (long long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) {
      const float v = Vec8<T>::ld1(x + i);
      acc = fmaf(v, v, acc);
    }
  const float s = block_sum(acc, scratch);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// ---------------------------------------------------------------------------
// C ABI. dtype codes: 0 = fp32, 1 = bf16.

static int mt_launch(int dir, int n_tensors, void* const* tensors, const long long* flat_off,
                     const long long* numel, void* flat, int tensor_dtype, int flat_dtype,
                     const float* scale_ptr, float scale_imm, hipStream_t stream) {
  int done = 0;
  while (done < n_tensors) {
    MtArgs a;
    int cnt = 0, chunks = 0;
    while (done + cnt < n_tensors && cnt < kMtMax) {
      const long long n = numel[done + cnt];
      a.tensor[cnt] = tensors[done + cnt];
      a.flat_off[cnt] = flat_off[done + cnt];
      a.numel[cnt] = n;
      a.chunk_start[cnt] = chunks;
      chunks += (int)((n + kMtChunk - 1) / kMtChunk);
      ++cnt;
    }
    a.chunk_start[cnt] = chunks;
    a.n_tensors = cnt;
    done += cnt;
    if (chunks == 0) continue;
#define MT(TT, FT, DIR)                                                                               \
  hipLaunchKernelGGL((mt_copy_kernel<TT, FT, DIR>), dim3(chunks), dim3(kBlock), 0, stream, a, (FT*)flat, \
                     scale_ptr, scale_imm)
    const int key = (dir << 2) | (tensor_dtype << 1) | flat_dtype;
    switch (key) {
      case 0: MT(float, float, 0); break;
      case 1: MT(float, unsigned short, 0); break;
      case 2: MT(unsigned short, float, 0); break;
      case 3: MT(unsigned short, unsigned short, 0); break;
      case 4: MT(float, float, 1); break;
      case 5: MT(float, unsigned short, 1); break;
      case 6: MT(unsigned short, float, 1); break;
      case 7: MT(unsigned short, unsigned short, 1); break;
      default: return -1;
    }
#undef MT
  }
  return (int)hipGetLastError();
}

DTF_API int dtf_bucket_flatten(int n_tensors, void* const* tensors, const long long* flat_off,
                               const long long* numel, void* flat, int tensor_dtype, int flat_dtype,
                               const float* scale_ptr, float scale_imm, hipStream_t stream) {
  return mt_launch(0, n_tensors, tensors, flat_off, numel, flat, tensor_dtype, flat_dtype, scale_ptr, scale_imm, stream);
}

DTF_API int dtf_bucket_unflatten(int n_tensors, void* const* tensors, const long long* flat_off,
                                 const long long* numel, void* flat, int tensor_dtype, int flat_dtype,
                                 const float* scale_ptr, float scale_imm, hipStream_t stream) {
  return mt_launch(1, n_tensors, tensors, flat_off, numel, flat, tensor_dtype, flat_dtype, scale_ptr, scale_imm, stream);
}

DTF_API int dtf_flat_scale_cast(const void* src, void* dst, long long n, int src_dtype, int dst_dtype,
                                const float* scale_ptr, float scale_imm, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_stream((n + 7) / 8, kBlock);
#define FS(S, D)                                                                                     \
  hipLaunchKernelGGL((flat_scale_cast_kernel<S, D>), dim3(grid), dim3(kBlock), 0, stream, (const S*)src, \
                     (D*)dst, n, scale_ptr, scale_imm)
  switch ((src_dtype << 1) | dst_dtype) {
    case 0: FS(float, float); break;
    case 1: FS(float, unsigned short); break;
    case 2: FS(unsigned short, float); break;
    case 3: FS(unsigned short, unsigned short); break;
    default: return -1;
  }
#undef FS
  return (int)hipGetLastError();
}

DTF_API int dtf_nary_axpy(void* dst, float w_dst, int n_src, const void* const* srcs, const float* weights,
                          long long n, int dtype, hipStream_t stream) {
  if (n <= 0) return 0;
  if (n_src > kNaryMax || n_src < 0) return -1;
  NaryArgs a;
  a.n_src = n_src;
  for (int i = 0; i < n_src; ++i) { a.src[i] = srcs[i]; a.w[i] = weights[i]; }
  const int grid = grid_stream((n + 7) / 8, kBlock);
  if (dtype == 0)
    hipLaunchKernelGGL((nary_axpy_kernel<float>), dim3(grid), dim3(kBlock), 0, stream, (float*)dst, a, w_dst, n);
  else
    hipLaunchKernelGGL((nary_axpy_kernel<unsigned short>), dim3(grid), dim3(kBlock), 0, stream,
                       (unsigned short*)dst, a, w_dst, n);
  return (int)hipGetLastError();
}

DTF_API int dtf_sumsq(const void* x, long long n, int dtype, float* out, hipStream_t stream) {
  if (n <= 0) return 0;
  const int grid = grid_for((n + 7) / 8, kBlock * 4);
  if (dtype == 0)
    hipLaunchKernelGGL((sumsq_kernel<float>), dim3(grid), dim3(kBlock), 0, stream, (const float*)x, n, out);
  else
    hipLaunchKernelGGL((sumsq_kernel<unsigned short>), dim3(grid), dim3(kBlock), 0, stream,
                       (const unsigned short*)x, n, out);
  return (int)hipGetLastError();
}
