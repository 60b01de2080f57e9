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
