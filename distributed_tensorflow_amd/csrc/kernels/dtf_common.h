// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//   * wavefront = 64 lanes, blocks are multiples of 64 threads;
//   * memory-bound kernels move 16 bytes per lane per access in a grid-stride
//     loop. Streaming kernels are launched with a grid that covers the buffer
//     in ONE pass (grid_stream): measured with tools/bench_stream.py, a copy
//     moves 6.1-6.5 TB/s that way against 4.6 TB/s for 2048 resident blocks
//     looping (the blocks of a looping grid drift apart and scatter their
//     accesses over DRAM pages). Kernels whose blocks write partial results
//     or synchronise with each other keep the capped grid (grid_for);
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

// Two floats to one packed bf16 pair (lo in bits 0-15) with ONE v_cvt_pk_bf16_f32; converting the halves separately
// costs two converts, a shift and an or.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// Eight (four) floats to consecutive bf16 values at a 16-byte (8-byte) aligned address.
__device__ __forceinline__ void store8_bf16(unsigned short* p, const float (&o)[8]) {
  const u32x4_t a = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7])};
  *reinterpret_cast<u32x4_t*>(p) = a;
}
__device__ __forceinline__ void store4_bf16(unsigned short* p, float a, float b, float c, float d) {
  const u32x2_t v = {pack_bf16x2(a, b), pack_bf16x2(c, d)};
  *reinterpret_cast<u32x2_t*>(p) = v;
}

// 16-byte store that goes through the caches to memory (system scope): the bytes are where another GPU that maps
// this buffer reads them once the storing wave's `s_waitcnt vmcnt(0)` has returned, with no cache write-back
// (a release fence writes back the whole L2 of the XCD). Costs the same as a plain 16-byte store.
__device__ __forceinline__ void store16_through(void* p, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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

// Philox4x32-10 counter-based generator: 4 x 32 random bits for (seed, counter). Dropout draws one
// block per 4 consecutive elements of the output: counter = offset + (linear element index >> 2),
// so the backward pass regenerates the forward mask from (seed, offset) instead of storing it.
__device__ __forceinline__ void philox4x32_10(unsigned long long seed, unsigned long long ctr, unsigned (&out)[4]) {
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// An element is dropped when its 32 random bits are below p * 2^32.
inline unsigned dropout_threshold(float p) {
  if (p <= 0.f) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}

// Sum over p of part[p * stride + c] for the 32 columns c = blockIdx.x * 32 + (tid & 31) of a
// 256-thread block: the 8 partial-lanes (tid >> 5) each walk every 8th partial with four loads in
// flight and are combined through LDS in a fixed order (bitwise reproducible). A one-thread-per-
// column loop over hundreds of partials is latency-bound; this spreads it over the chip.
// Returns true on the threads (partial-lane 0, column in range) that hold the sum in `s`.
__device__ __forceinline__ bool partials_col_sum(const float* __restrict__ part, int n_part, long long stride, int n_cols,
                                                 float& s) {
  __shared__ float red[8][33];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float a = 0.f;
  if (c < n_cols) {
    int p = pl;
    for (; p + 24 < n_part; p += 32) {
      const float* q = part + (long long)p * stride + c;
      const float x0 = q[0], x1 = q[8 * stride], x2 = q[16 * stride], x3 = q[24 * stride];
      a += (x0 + x1) + (x2 + x3);
    }
    for (; p < n_part; p += 8) a += part[(long long)p * stride + c];
  }
  red[pl][cl] = a;
  __syncthreads();
  if (pl != 0 || c >= n_cols) return false;
  s = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) s += red[r][cl];
  return true;
}

// Upper bound of a streaming grid; 2^23 blocks of 256 threads stay below 2^32 threads per launch. Settable for
// A/B measurements (dtf_set_stream_blocks).
inline int& stream_block_cap() {
  static int cap = 1 << 23;
  return cap;
}
inline int grid_stream(long long n_items, int per_block) {
  long long b = (n_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > stream_block_cap()) b = stream_block_cap();
  return (int)b;
}

inline int grid_for(long long n_items, int per_block) {
  long long b = (n_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

}  // namespace dtf
