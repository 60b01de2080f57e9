// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, blocks are multiples of 64 threads;
//   * memory-bound kernels move 16 bytes per lane per access and use a
//     grid-stride loop capped at kMaxBlocks so that the launch is a few
//     waves deep on all 256 CUs;
//   * every exported entry point is `extern "C"`, takes raw device pointers
//     and a hipStream_t, never allocates and never synchronises, so a caller
//     can capture it into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#define DTF_API extern "C" __attribute__((visibility("default")))

namespace dtf {

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kMaxBlocks = 2048;  // 256 CUs x 8 blocks

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned)h) << 16);
}

// Round-to-nearest-even through the hardware convert; keeps NaN a NaN.
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __hip_bfloat16 b = __float2bfloat16(f);
  return *reinterpret_cast<unsigned short*>(&b);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; `scratch` holds >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

__device__ __forceinline__ float block_max(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = -INFINITY;
  for (int i = 0; i < nw; ++i) r = fmaxf(r, scratch[i]);
  return r;
}

inline int grid_for(long long n_items, int per_block) {
  long long b = (n_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

}  // namespace dtf
