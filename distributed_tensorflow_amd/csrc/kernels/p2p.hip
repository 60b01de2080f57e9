// Collectives on peer pointers: every rank of a node maps the gradient arena of every other rank
// (HIP IPC, csrc/runtime/ipc.cpp) and the kernels below read peer memory with plain 16-byte
// loads. On an 8-GPU MI355X node each peer is one xGMI hop away on its own link, so a kernel
// that keeps loads to all 7 peers in flight drives all 7 links at once; that is what a single
// ring cannot do.
//
//   dtf_p2p_reduce   out[i] = scale * sum_j peer_j[off + i]       (pull half of a one-hop
//                    reduce-scatter, or a one-shot all-reduce when off = 0, n = everything;
//                    summation order is j = 0..world-1 on every rank, so all ranks that reduce
//                    the same elements get the same bits)
//   dtf_p2p_gather   dst[shard_j] = peer_j[shard_j] for j != rank (pull all-gather)
//   dtf_p2p_barrier  device-side barrier over fine-grained flag arrays; every wait is bounded
//                    by a wall-clock budget, after which the kernel records a timeout in
//                    *status and ends, so the grid always drains
//
// Ring and tree schedules are sequences of dtf_p2p_reduce calls on two pointers
// (comm/ipc.py). The reference has no counterpart: its only collective is TensorFlow's CPU
// ring over gRPC (reference decentralized/native/dist_keras.py:77-78).
#include "dtf_common.h"
#include <stdlib.h>

using namespace dtf;

namespace {

constexpr int kMaxPeers = 8;

struct PeerPtrs {
  const void* p[kMaxPeers];
};
struct FlagPtrs {
  unsigned* p[kMaxPeers];
};

template <int W>
__global__ __launch_bounds__(kBlock) void p2p_reduce_f32(PeerPtrs pp, float* __restrict__ out, long long off,
                                                         long long n, float scale) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
    f32x4 v[W];
#pragma unroll
    for (int j = 0; j < W; ++j)
      v[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(pp.p[j]) + off) + i);
    f32x4 s = v[0];
#pragma unroll
    for (int j = 1; j < W; ++j) s += v[j];
    reinterpret_cast<f32x4*>(out)[i] = s * scale;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float s = reinterpret_cast<const float*>(pp.p[0])[off + i];
    for (int j = 1; j < W; ++j) s += reinterpret_cast<const float*>(pp.p[j])[off + i];
    out[i] = s * scale;
  }
}

// bf16 buffers, fp32 accumulation, one rounding at the end.
template <int W>
__global__ __launch_bounds__(kBlock) void p2p_reduce_bf16(PeerPtrs pp, unsigned short* __restrict__ out,
                                                          long long off, long long n, float scale) {
  const long long n8 = n >> 3;
  const long long stride = (long long)gridDim.x * kBlock;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n8; i += stride) {
    u16x8 v[W];
#pragma unroll
    for (int j = 0; j < W; ++j)
      v[j] = __builtin_nontemporal_load(
          reinterpret_cast<const u16x8*>(reinterpret_cast<const unsigned short*>(pp.p[j]) + off) + i);
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = bf16_to_f32(v[0][e]);
#pragma unroll
      for (int j = 1; j < W; ++j) s += bf16_to_f32(v[j][e]);
      o[e] = f32_to_bf16(s * scale);
    }
    reinterpret_cast<u16x8*>(out)[i] = o;
  }
  for (long long i = (n8 << 3) + (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
    float s = bf16_to_f32(reinterpret_cast<const unsigned short*>(pp.p[0])[off + i]);
    for (int j = 1; j < W; ++j) s += bf16_to_f32(reinterpret_cast<const unsigned short*>(pp.p[j])[off + i]);
    out[i] = f32_to_bf16(s * scale);
  }
}

// 16-byte units; blockIdx.y = source peer, so the copies from all peers are in flight together.
__global__ __launch_bounds__(kBlock) void p2p_gather_kernel(PeerPtrs pp, char* __restrict__ dst, long long shard_bytes,
                                                            long long total_bytes, int rank) {
  const int j = blockIdx.y;
  if (j == rank) return;
  const long long lo = (long long)j * shard_bytes;
  long long hi = lo + shard_bytes;
  if (hi > total_bytes) hi = total_bytes;
  if (lo >= hi) return;
  const char* src = reinterpret_cast<const char*>(pp.p[j]);
  const long long n16 = (hi - lo) >> 4;
  const long long stride = (long long)gridDim.x * kBlock;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n16; i += stride)
    reinterpret_cast<u32x4*>(dst + lo)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + lo) + i);
  for (long long b = lo + (n16 << 4) + (long long)blockIdx.x * kBlock + threadIdx.x; b < hi; b += stride)
    dst[b] = src[b];
}

// One block. Lane j < world announces this rank's arrival in peer j's flag array, then waits for
// peer j's announcement in its own. Flags only grow (epoch numbers), so nothing is ever reset.
__global__ __launch_bounds__(kWave) void p2p_barrier_kernel(FlagPtrs fp, int rank, int world, unsigned epoch,
                                                            long long budget_ticks, unsigned* __restrict__ status) {
  const int j = threadIdx.x;
  if (j >= world) return;
  __threadfence_system();
  __hip_atomic_store(fp.p[j] + rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  const long long t0 = wall_clock64();
  bool ok = false;
  while (true) {
    const unsigned seen = __hip_atomic_load(fp.p[rank] + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)(seen - epoch) >= 0) { ok = true; break; }
    if (wall_clock64() - t0 > budget_ticks) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (!ok) atomicOr(status, 1u << j);   // bit j: peer j never arrived
  __threadfence_system();
}

template <int W>
void launch_reduce(const PeerPtrs& pp, void* out, long long off, long long n, int dtype, float scale, int blocks,
                   hipStream_t stream) {
  if (dtype == 0)
    hipLaunchKernelGGL(p2p_reduce_f32<W>, dim3(blocks), dim3(kBlock), 0, stream, pp, (float*)out, off, n, scale);
  else
    hipLaunchKernelGGL(p2p_reduce_bf16<W>, dim3(blocks), dim3(kBlock), 0, stream, pp, (unsigned short*)out, off, n,
                       scale);
}

}  // namespace

// out[i] = scale * sum_{j < world} peers[j][off + i], i in [0, n). `out` may alias peers[rank] + off
// (each element is read before it is written by the same lane). dtype 0 = fp32, 1 = bf16.
// max_blocks caps the grid (0 = default) so that a collective leaves CUs to the backward pass.
DTF_API int dtf_p2p_reduce(const void* const* peers, int world, void* out, long long off, long long n, int dtype,
                           float scale, int max_blocks, hipStream_t stream) {
  if (!peers || !out || world < 1 || world > kMaxPeers || (dtype != 0 && dtype != 1) || off < 0) return -1;
  if (n <= 0) return 0;
  const int esz = dtype == 0 ? 4 : 2;
  PeerPtrs pp;
  for (int j = 0; j < kMaxPeers; ++j) pp.p[j] = j < world ? peers[j] : nullptr;
  for (int j = 0; j < world; ++j)
    if (!pp.p[j] || ((uintptr_t)pp.p[j] + (uintptr_t)off * esz) % 16) return -2;
  if ((uintptr_t)out % 16) return -2;
  int blocks = grid_for(n / (16 / esz) + 1, kBlock);
  if (max_blocks > 0 && blocks > max_blocks) blocks = max_blocks;
  switch (world) {
    case 1: launch_reduce<1>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 2: launch_reduce<2>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 3: launch_reduce<3>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 4: launch_reduce<4>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 5: launch_reduce<5>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 6: launch_reduce<6>(pp, out, off, n, dtype, scale, blocks, stream); break;
    case 7: launch_reduce<7>(pp, out, off, n, dtype, scale, blocks, stream); break;
    default: launch_reduce<8>(pp, out, off, n, dtype, scale, blocks, stream); break;
  }
  return (int)hipGetLastError();
}

// dst[j*shard_bytes ...) = peers[j][j*shard_bytes ...) for every j != rank (byte offsets relative to the
// buffer start; shard_bytes must be a multiple of 16).
DTF_API int dtf_p2p_gather(const void* const* peers, int world, int rank, void* dst, long long shard_bytes,
                           long long total_bytes, int max_blocks, hipStream_t stream) {
  if (!peers || !dst || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) return -1;
  if (shard_bytes <= 0 || shard_bytes % 16 || (uintptr_t)dst % 16) return -2;
  if (world == 1) return 0;
  PeerPtrs pp;
  for (int j = 0; j < kMaxPeers; ++j) pp.p[j] = j < world ? peers[j] : nullptr;
  for (int j = 0; j < world; ++j)
    if (j != rank && (!pp.p[j] || (uintptr_t)pp.p[j] % 16)) return -2;
  int blocks = grid_for(shard_bytes / 16 + 1, kBlock);
  const int cap = max_blocks > 0 ? max_blocks : kMaxBlocks;
  if (blocks * world > cap) blocks = cap / world > 0 ? cap / world : 1;
  hipLaunchKernelGGL(p2p_gather_kernel, dim3(blocks, world), dim3(kBlock), 0, stream, pp, (char*)dst, shard_bytes,
                     total_bytes, rank);
  return (int)hipGetLastError();
}

// flags[j] is rank j's flag array (world unsigned words, fine-grained memory), mapped into this
// process. budget_ms bounds the wait; *status (device memory, local) collects timeout bits.
DTF_API int dtf_p2p_barrier(unsigned* const* flags, int world, int rank, unsigned epoch, int budget_ms,
                            unsigned* status, hipStream_t stream) {
  if (!flags || !status || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) return -1;
  FlagPtrs fp;
  for (int j = 0; j < kMaxPeers; ++j) fp.p[j] = j < world ? flags[j] : nullptr;
  for (int j = 0; j < world; ++j)
    if (!fp.p[j]) return -2;
  if (budget_ms < 1) budget_ms = 1;
  const long long ticks = (long long)budget_ms * 100000LL;   // wall_clock64 counts at 100 MHz
  hipLaunchKernelGGL(p2p_barrier_kernel, dim3(1), dim3(kWave), 0, stream, fp, rank, world, epoch, ticks, status);
  return (int)hipGetLastError();
}

// Every XCD's L2 (blocks are dealt round-robin over the XCDs: 64 blocks reach each of the eight several times over)
// and every CU's L1 they land on drop their copies of memory this device does not own. Launched in front of a kernel
// that reads a peer's buffer with ordinary loads (the optimizer kernel reading a worker's gradient, a copy out of an
// owner's reply slot): what that kernel then loads comes over the link, not out of a cache line left by last step's
// read of the same addresses.
namespace {
__global__ __launch_bounds__(kWave) void p2p_acquire_kernel() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope
}
}  // namespace

DTF_API int dtf_p2p_acquire(hipStream_t stream) {
  hipLaunchKernelGGL(p2p_acquire_kernel, dim3(64), dim3(kWave), 0, stream);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The one-hop all-reduce (reduce-scatter + all-gather) as ONE launch per rank instead of five (barrier, reduce,
// barrier, gather, barrier): a resident grid (sized from the bucket, capped so that the backward pass keeps its CUs)
// walks through
//   entry barrier -> rank r sums shard r of every peer into its own buffer -> barrier -> pull the other ranks'
//   finished shards -> exit barrier (nobody may overwrite a shard a peer is still reading).
// No barrier writes a cache back. What a peer reads is either the work of EARLIER kernels (the packed gradients: a
// kernel's end has already written its stores back, that is what makes them visible to the next kernel on another
// XCD) or the reduced shard, which this kernel stores through the caches (`store16_through`, `sc0 sc1`) and drains
// with `s_waitcnt vmcnt(0)` before it signals; a release fence would instead write back the whole L2 of the XCD,
// dirty with the backward pass's activations, once per block and barrier. Flags and counters are relaxed
// device- / system-scope accesses behind that drain; after a barrier every block invalidates its stale copies of
// peer memory (system-scope acquire) before it loads again.
// A barrier is ONE hop: when all waves of a block have drained their stores, lane j of the block adds 1 to the
// counter "arrivals from rank r" in peer j's flag array (fine-grained memory, system-scope atomic over the link) and
// polls its own array's counter "arrivals from rank j" until every block of rank j's grid has arrived. So every
// block learns by itself that every block of every rank is there: no grid-level gather through a block 0, no
// second counter to release the others (that chain was four dependent round trips per barrier). All ranks launch
// the same grid for a call (it is a function of the bucket size and the CU budget). Every wait is bounded by a
// wall-clock budget: on expiry bit j (the late rank) is set in *status and the kernel carries on, so the grid always
// drains and the host learns that the result is not to be trusted (comm/ipc.py checks the status word every step).
// Counters only grow; the host passes the arrivals counted before each call.
namespace {

// Block width is a launch parameter (256, 512 or 1024 threads, compiled for 1024): with at most one block per CU (the
// residency rule) a four-wave block leaves one wave per SIMD, whose loads, sums and stores then run one after the
// other (2-3 TB/s on one device); the host widens the blocks once the grid has reached its CU budget.
constexpr int kFusedBlock = 1024;

struct FusedArgs {
  PeerPtrs bufs;          // element 0 of every rank's registered buffer
  FlagPtrs flags;         // every rank's counter array (one word per source rank)
  unsigned* status;
  long long off, n, shard;   // region [off, off + n) in elements, shard = elements per rank (16-byte multiple)
  long long budget;
  float scale;
  int rank, world;
  unsigned arrive_base;   // arrivals per source rank counted before this call (sum over earlier calls of 3 x grid)
  int store_mode;         // A/B switch (DTF_FUSED_STORE): 0 plain stores (WRONG across XCDs, timing only), 1 sc1, 2 sc0 sc1
};

inline int fused_store_mode() {
  const char* e = getenv("DTF_FUSED_STORE");
  return e ? atoi(e) : 2;
}

__device__ __forceinline__ void fused_barrier(const FusedArgs& a, unsigned k, unsigned grid) {
  // k = 1, 2, 3: the call's k-th barrier
  drain_stores();                                             // every wave: its write-through stores are in memory
  __syncthreads();
  const int j = threadIdx.x;
  if (j < a.world) {
    __hip_atomic_fetch_add(a.flags.p[j] + a.rank, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned want = a.arrive_base + grid * k;
    const long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(a.flags.p[a.rank] + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
      if (wall_clock64() - t0 > a.budget) { atomicOr(a.status, 1u << j); break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  // ONE wave per block drops the stale copies of peer memory (system-scope acquire: this CU's L1, the non-coherent
  // lines of this XCD's L2) before anybody loads again; the other waves only wait for it. Every wave fencing cost
  // ~50 us per call at 16 MiB: each invalidate walks the cache.
  if (threadIdx.x < kWave) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  __syncthreads();
}

template <int W, typename T>
__device__ __forceinline__ void fused_body(const FusedArgs& a) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int PER = 16 / (int)sizeof(T);                    // elements per 16-byte access
  const long long stride = (long long)gridDim.x * blockDim.x;
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned grid = gridDim.x;
  fused_barrier(a, 1, grid);
  // ---- reduce-scatter: this rank's shard, summed over the ranks in rank order (same bits on every rank)
  {
    const long long lo = min((long long)a.rank * a.shard, a.n);
    const long long cnt = max(0ll, min(a.shard, a.n - lo));
    const long long nv = cnt / PER;
    T* out = reinterpret_cast<T*>(const_cast<void*>(a.bufs.p[a.rank])) + a.off + lo;
    // at least eight 16-byte loads in flight per lane whatever the number of ranks
    constexpr int U = W >= 8 ? 1 : (W >= 4 ? 2 : (W >= 2 ? 4 : 8));
    for (long long i0 = tid; i0 < nv; i0 += stride * U) {
      u32x4 v[U][W];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = min(i0 + u * stride, nv - 1);
#pragma unroll
        for (int j = 0; j < W; ++j)
          v[u][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.bufs.p[j]) + a.off + lo) + i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * stride;
        if (i >= nv) break;
        u32x4 o;
        if (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float s = __uint_as_float(v[u][0][e]);
#pragma unroll
            for (int j = 1; j < W; ++j) s += __uint_as_float(v[u][j][e]);
            o[e] = __float_as_uint(s * a.scale);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float s0 = __uint_as_float(v[u][0][e] << 16), s1 = __uint_as_float(v[u][0][e] & 0xffff0000u);
#pragma unroll
            for (int j = 1; j < W; ++j) {
              s0 += __uint_as_float(v[u][j][e] << 16);
              s1 += __uint_as_float(v[u][j][e] & 0xffff0000u);
            }
            o[e] = pack_bf16x2(s0 * a.scale, s1 * a.scale);
          }
        }
        // the peers pull this shard in the next phase
        if (a.store_mode == 2) store16_through(reinterpret_cast<u32x4*>(out) + i, o);
        else if (a.store_mode == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<u32x4*>(out) + i), "v"(o) : "memory");
        else reinterpret_cast<u32x4*>(out)[i] = o;
      }
    }
    for (long long i = nv * PER + tid; i < cnt; i += stride) {   // tail of the last shard
      float s = 0.f;
      for (int j = 0; j < W; ++j) {
        const T* p = reinterpret_cast<const T*>(a.bufs.p[j]) + a.off + lo + i;
        s += sizeof(T) == 4 ? *reinterpret_cast<const float*>(p) : bf16_to_f32(*reinterpret_cast<const unsigned short*>(p));
      }
      s *= a.scale;
      if (sizeof(T) == 4)
        __hip_atomic_store(reinterpret_cast<float*>(out + i), s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else
        __hip_atomic_store(reinterpret_cast<unsigned short*>(out + i), f32_to_bf16(s), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  fused_barrier(a, 2, grid);
  // ---- all-gather: the other ranks' finished shards, all peers' copies interleaved (all links at once), four
  // 16-byte loads in flight per lane
  {
    char* dst = reinterpret_cast<char*>(const_cast<void*>(a.bufs.p[a.rank])) + a.off * (long long)sizeof(T);
    const long long shard_b = a.shard * (long long)sizeof(T), total_b = a.n * (long long)sizeof(T);
    const long long nv = shard_b >> 4;
    constexpr int U = 8;
    for (long long i0 = tid; i0 < nv * W; i0 += stride * U) {
      u32x4 v[U];
      long long bo[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * stride;
        const int j = (int)(i % W);
        const long long b = (long long)j * shard_b + ((i / W) << 4);
        bo[u] = (i < nv * W && j != a.rank && b < total_b) ? b : -1;
        if (bo[u] >= 0 && b + 16 <= total_b)
          v[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(
              reinterpret_cast<const char*>(a.bufs.p[j]) + a.off * (long long)sizeof(T) + b));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (bo[u] < 0) continue;
        if (bo[u] + 16 <= total_b) {
          *reinterpret_cast<u32x4*>(dst + bo[u]) = v[u];
        } else {
          const int j = (int)((i0 + u * stride) % W);
          const char* src = reinterpret_cast<const char*>(a.bufs.p[j]) + a.off * (long long)sizeof(T);
          for (long long c = bo[u]; c < total_b && c < bo[u] + 16; ++c) dst[c] = src[c];
        }
      }
    }
  }
  fused_barrier(a, 3, grid);
}

template <int W, typename T>
__global__ __launch_bounds__(kFusedBlock) void p2p_allreduce_fused_kernel(FusedArgs a) {
  fused_body<W, T>(a);
}

// Test vehicle: W virtual ranks in ONE launch (blockIdx.y = rank), so that their grids are resident together by
// construction; each runs the very code of a real rank against its own counters, flags and status word.
struct VirtualRanks {
  unsigned* status[kMaxPeers];
};

template <int W, typename T>
__global__ __launch_bounds__(kFusedBlock) void p2p_allreduce_fused_virtual_kernel(FusedArgs a, VirtualRanks v) {
  a.rank = blockIdx.y;
  a.status = v.status[blockIdx.y];
  fused_body<W, T>(a);
}

template <int W>
void launch_fused_virtual(const FusedArgs& a, const VirtualRanks& v, int dtype, int blocks, int threads, hipStream_t stream) {
  if (dtype == 0)
    hipLaunchKernelGGL((p2p_allreduce_fused_virtual_kernel<W, float>), dim3(blocks, W), dim3(threads), 0, stream, a, v);
  else
    hipLaunchKernelGGL((p2p_allreduce_fused_virtual_kernel<W, unsigned short>), dim3(blocks, W), dim3(threads), 0, stream, a, v);
}

template <int W>
void launch_fused(const FusedArgs& a, int dtype, int blocks, int threads, hipStream_t stream) {
  if (dtype == 0) hipLaunchKernelGGL((p2p_allreduce_fused_kernel<W, float>), dim3(blocks), dim3(threads), 0, stream, a);
  else hipLaunchKernelGGL((p2p_allreduce_fused_kernel<W, unsigned short>), dim3(blocks), dim3(threads), 0, stream, a);
}

}  // namespace

// The single-launch all-reduce for `world` virtual ranks of one process in one grid (tests): statuses[r] is rank r's
// status word, flags[r] its counter array.
DTF_API int dtf_p2p_allreduce_fused_virtual(const void* const* peers, unsigned* const* flags, int world,
                                            unsigned* const* statuses, long long off, long long n, int dtype,
                                            float scale, unsigned arrive_base, int blocks, int threads,
                                            int budget_ms, hipStream_t stream) {
  if (!peers || !flags || !statuses || world < 1 || world > kMaxPeers) return -1;
  if (threads != 256 && threads != 512 && threads != 1024) return -1;
  if ((dtype != 0 && dtype != 1) || off < 0 || blocks < 1 || blocks * world > 256) return -1;
  if (n <= 0) return 0;
  const int esz = dtype == 0 ? 4 : 2;
  FusedArgs a;
  VirtualRanks v;
  for (int j = 0; j < kMaxPeers; ++j) {
    a.bufs.p[j] = j < world ? peers[j] : nullptr;
    a.flags.p[j] = j < world ? flags[j] : nullptr;
    v.status[j] = j < world ? statuses[j] : nullptr;
  }
  for (int j = 0; j < world; ++j)
    if (!a.bufs.p[j] || !a.flags.p[j] || !v.status[j] || ((uintptr_t)a.bufs.p[j] + (uintptr_t)off * esz) % 16)
      return -2;
  const long long q = 16 / esz;
  const long long per = (n + world - 1) / world;
  a.shard = (per + q - 1) / q * q;
  a.status = nullptr; a.off = off; a.n = n; a.scale = scale; a.rank = 0; a.world = world;
  a.arrive_base = arrive_base; a.store_mode = fused_store_mode();
  a.budget = (long long)(budget_ms < 1 ? 1 : budget_ms) * 100000LL;
  switch (world) {
    case 1: launch_fused_virtual<1>(a, v, dtype, blocks, threads, stream); break;
    case 2: launch_fused_virtual<2>(a, v, dtype, blocks, threads, stream); break;
    case 3: launch_fused_virtual<3>(a, v, dtype, blocks, threads, stream); break;
    case 4: launch_fused_virtual<4>(a, v, dtype, blocks, threads, stream); break;
    case 5: launch_fused_virtual<5>(a, v, dtype, blocks, threads, stream); break;
    case 6: launch_fused_virtual<6>(a, v, dtype, blocks, threads, stream); break;
    case 7: launch_fused_virtual<7>(a, v, dtype, blocks, threads, stream); break;
    default: launch_fused_virtual<8>(a, v, dtype, blocks, threads, stream); break;
  }
  return (int)hipGetLastError();
}

// One-launch direct all-reduce of the region [off, off + n) (elements) of every rank's registered buffer.
// peers[j]: element 0 of rank j's buffer; flags[j]: rank j's counter array (zero-initialised fine-grained words that
// persist across calls); arrive_base: sum over the earlier calls of 3 * blocks.
// blocks x threads: grid and block size of THIS call, the same on every rank. Every block must be able to become resident while the
// others spin in a barrier: at most one block per CU (256) is accepted, which holds next to any kernel that itself
// terminates (a block that has to wait for a free slot only delays the barrier).
DTF_API int dtf_p2p_allreduce_fused(const void* const* peers, unsigned* const* flags, int world, int rank,
                                    unsigned* status, long long off, long long n, int dtype, float scale,
                                    unsigned arrive_base, int blocks, int threads, int budget_ms,
                                    hipStream_t stream) {
  if (!peers || !flags || !status || world < 1 || world > kMaxPeers || rank < 0 || rank >= world) return -1;
  if (threads != 256 && threads != 512 && threads != 1024) return -1;
  if ((dtype != 0 && dtype != 1) || off < 0 || blocks < 1 || blocks > 256) return -1;
  if (n <= 0) return 0;
  const int esz = dtype == 0 ? 4 : 2;
  FusedArgs a;
  for (int j = 0; j < kMaxPeers; ++j) {
    a.bufs.p[j] = j < world ? peers[j] : nullptr;
    a.flags.p[j] = j < world ? flags[j] : nullptr;
  }
  for (int j = 0; j < world; ++j)
    if (!a.bufs.p[j] || !a.flags.p[j] || ((uintptr_t)a.bufs.p[j] + (uintptr_t)off * esz) % 16) return -2;
  const long long q = 16 / esz;
  const long long per = (n + world - 1) / world;
  a.shard = (per + q - 1) / q * q;
  a.status = status; a.off = off; a.n = n; a.scale = scale; a.rank = rank; a.world = world;
  a.arrive_base = arrive_base; a.store_mode = fused_store_mode();
  a.budget = (long long)(budget_ms < 1 ? 1 : budget_ms) * 100000LL;
  switch (world) {
    case 1: launch_fused<1>(a, dtype, blocks, threads, stream); break;
    case 2: launch_fused<2>(a, dtype, blocks, threads, stream); break;
    case 3: launch_fused<3>(a, dtype, blocks, threads, stream); break;
    case 4: launch_fused<4>(a, dtype, blocks, threads, stream); break;
    case 5: launch_fused<5>(a, dtype, blocks, threads, stream); break;
    case 6: launch_fused<6>(a, dtype, blocks, threads, stream); break;
    case 7: launch_fused<7>(a, dtype, blocks, threads, stream); break;
    default: launch_fused<8>(a, dtype, blocks, threads, stream); break;
  }
  return (int)hipGetLastError();
}
