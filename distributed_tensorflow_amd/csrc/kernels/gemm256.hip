// Deep-pipelined bf16 GEMM for large compute-bound shapes (gfx950): C[M,N] = act(op(A) * op(B) + bias) in
// the three layouts of a dense layer: NT (forward, A[M,K] B[N,K]), NN (data gradient, A[M,K] B[K,N]) and
// TN (weight gradient, A[K,M] B[K,N], with split-K because its contraction runs over the tokens).
//
// The 128x128 kernel in gemm.hip reads 0.5 KiB of LDS per MFMA, which makes the LDS port as busy as
// the matrix cores and caps it near a third of the dense peak. This kernel gives every wave a
// 128 x 64 output (0.375 KiB of LDS per MFMA), runs one 256 x 256 x 64 tile per CU with 8 waves, and
// never lets the memory pipeline drain:
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4); the XOR swizzle that keeps the
//     ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address, the LDS
//     image of a DMA being lane-linear;
//   * a K-tile is four half-tiles of 128 rows (A rows 0-127 / 128-255, B likewise) in one of two
//     LDS buffers, and a wave's output is four 64 x 32 quadrants, one per half-tile pair, so the
//     half-tiles are needed one phase after the other rather than all at once;
//   * each of the 4 phases of a K-tile reads one quadrant's fragments, issues the DMA of ONE
//     half-tile (three half-tiles stay in flight across the barriers: `s_waitcnt vmcnt(6)`, raw
//     s_barrier, never a drain), then runs the quadrant's 16 MFMAs;
//   * the waves of the two row groups run one barrier apart, so on every SIMD one wave issues
//     MFMAs while the other reads LDS and issues DMA.
// Hazards (phases counted globally, P = 4 * tile + phase): a half-tile last read in phase L is
// overwritten by a DMA issued in phase >= L + 2; a DMA waited for in phase W is read in phase
// >= W + 1. The issue order below (phase 0: B1 of tile t+1, 1: A1 of t+1, 2: A0 of t+2, 3: B0 of
// t+2) and the read order (0: A0 B0, 1: B1, 2: A1, 3: none) satisfy both with a wait that leaves the
// three newest half-tiles in flight.
// A k-major operand (rows = k) is staged as 64 k-rows of 256 bytes and its fragments are read with the
// hardware-transposing ds_read_b64_tr_b16; the DMA source permutation is the XOR of that layout.
// The reference trains its dense layers through TensorFlow's CPU MatMul (reference initializer.py:16-18).
#include "dtf_common.h"
#include <stdlib.h>
#include <type_traits>

using namespace dtf;

namespace {

constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kT256 = 512;
constexpr int kHalfBytes = 128 * TK * 2;   // 16 KiB: 128 rows x 64 bf16
constexpr int kBufBytes = 4 * kHalfBytes;  // A0 A1 B0 B1
constexpr int kLds256 = 2 * kBufBytes;     // 128 KiB of operand buffers
constexpr int kCusStagger = 256;
constexpr int kLdsAll = kLds256;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct G256Args {
  const unsigned short* A;
  const unsigned short* B;
  unsigned short* C;
  const void* bias;
  long long lda, ldb, ldc;
  int M, N, K;
  int bias_dtype, act;
  int tiles_m, tiles_n;
  const unsigned short* Cin;   // optional bf16 tensor (indexed like C) combined with the result ...
  int cin_mode;                // ... 1: added, 2: multiplied
  unsigned short* C2;          // act 3: second output (indexed like C), the derivative of the activation
  float* W;              // fp32 slabs [splits][M][N] when splits > 1
  int k_per_split;       // multiple of TK
  int splits;
  long long* stamps;     // diagnostic: per output tile 4 wall-clock stamps (loop top, operands there, k-loop done, stores issued); null = off
  float* col_sums;       // optional fp32 [4 * tiles_m][N]: column sums of the stored result over each wave's 64 rows
  int stagger;           // measurement hooks. > 0: 10 ns ticks between the start groups of the first round of workgroups; -1: no output stores
};

__device__ __forceinline__ int swz256(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// k-major half-tile: 64 k-rows x 16 chunks of 16 bytes; chunk XOR for conflict-free transposed reads
__device__ __forceinline__ int swz_t256(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

typedef short s16x4_t256 __attribute__((ext_vector_type(4)));
typedef short s16x8_t256 __attribute__((ext_vector_type(8)));

// Fragment (16 rows x 32 k) of a half-tile: T = 0 rows are k-contiguous (ds_read_b128), T = 1 k-major
// (two transposing 64-bit reads). row0 is a multiple of 16.
template <int T>
__device__ __forceinline__ bf16x8_t frag256(const char* half, int row0, int ks, int lane) {
  if (T == 0) {
    return *reinterpret_cast<const bf16x8_t*>(half + swz256(row0 + (lane & 15), ks * 4 + (lane >> 4)));
  } else {
    const int g4 = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int k = ks * 32 + 8 * g4 + q;
    const int chunk = (row0 >> 3) + (p >> 1);
    const int off = k * 256 + ((chunk ^ swz_t256(k)) << 4) + ((p & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4_t256* lp_t;
    const s16x4_t256 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(half + off));
    const s16x4_t256 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(half + off + 4 * 256));
    const s16x8_t256 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

// GELU (erf form) and its derivative for bf16 outputs, sized for an epilogue that handles 128 elements per lane:
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) through the rational approximation erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3) e^{-z^2},
// t = 1 / (1 + p z) (Abramowitz & Stegun 7.1.25, |error| <= 2.5e-5, far below bf16 resolution). With z = |x| / sqrt 2
// the exponential is e^{-x^2 / 2}, the same one the density phi(x) of the derivative needs: one v_exp_f32 and one
// v_rcp_f32 per element instead of the branchy library erff.
//   gelu(x) = x Phi(x),   gelu'(x) = Phi(x) + x phi(x)
// Not inlined: spread over the 128 unrolled elements of the epilogue its temporaries push the kernel (already at
// 240-255 registers) into scratch spills that slow every variant down by 15-30 %.
// Two elements per call so that the polynomial runs on packed fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32).
typedef float f32x2_256 __attribute__((ext_vector_type(2)));
__device__ __noinline__ f32x4 gelu_pair256(f32x2_256 x) {        // -> {gelu(x0), gelu(x1), gelu'(x0), gelu'(x1)}
  const f32x2_256 xx = x * x * -0.72134752044448170f;
  const f32x2_256 e = {__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};          // e^{-x^2/2}
  const f32x2_256 ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2_256 den = ax * (0.47047f * 0.70710678118654752f) + 1.f;
  const f32x2_256 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  // half of the polynomial: hq = 0.5 (1 - erf(|x| / sqrt 2))
  const f32x2_256 hq = t * ((t * (0.5f * 0.7478556f) + 0.5f * -0.0958798f) * t + 0.5f * 0.3480242f) * e;
  const f32x2_256 one_m = 1.f - hq;
  const f32x2_256 cdf = {x[0] >= 0.f ? one_m[0] : hq[0], x[1] >= 0.f ? one_m[1] : hq[1]};
  const f32x2_256 h = x * cdf;
  const f32x2_256 d = x * 0.3989422804014327f * e + cdf;
  return f32x4{h[0], h[1], d[0], d[1]};
}

#define WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#define WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

template <int A_T, int B_T>
__global__ __launch_bounds__(kT256, 1) void gemm256_kernel(G256Args g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const int fr = lane & 15, fq = lane >> 4;

  // Persistent over output tiles: block b runs tiles b, b + gridDim.x, ... The DMA of the next tile's first two
  // K-tiles is issued BEFORE the epilogue of the current one, so that the output stores (younger in the wave's
  // in-order memory queue) can stay in flight under the next tile's first K-tiles.
  const int nwg = g.tiles_m * g.tiles_n;
  // Split-K launches use a flat grid over (split, tile) pairs in split-major order, cut into eight contiguous ranges,
  // one per XCD (workgroup L is dispatched to XCD L % 8): the workgroups of one split read the same K-slab of both
  // operands at the same time, and with a whole split on one XCD that slab crosses the fabric once instead of once
  // per XCD that happens to hold some of the split's tiles.
  int split = 0, flat_tile = -1;
  if (g.splits > 1) {
    const int total = nwg * g.splits, L = blockIdx.x;
    const int xcd = L & 7, q8 = total >> 3, rem = total & 7;
    const int w = (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (L >> 3);
    split = w / nwg;
    flat_tile = w - split * nwg;
  }
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nt = (kend - kbeg) / TK;
  const long long adv_a = A_T == 0 ? TK : (long long)TK * g.lda;
  const long long adv_b = B_T == 0 ? TK : (long long)TK * g.ldb;
  const int dma_dst = wid * 1024;          // wave-uniform; the DMA adds lane * 16
  const int arow = wr * 64;                    // + i * 16, inside an A half-tile
  const int brow = wc * 32;                    // + j * 16, inside a B half-tile

  int m0 = 0, n0 = 0;
  unsigned src_off[4][2];                      // element offsets; the host checks that they fit 32 bits
  // tile index -> (m0, n0) and the per-thread source offsets of its two DMA pieces in each of the four
  // half-tiles (first K-tile of this split)
  auto setup_tile = [&](int it) {
    // XCD-aware bijective map (an XCD owns a contiguous range of tiles; with gridDim.x a multiple of 8
    // a block stays on one XCD's range for all its tiles)
    const int xcd = it & 7, q8 = nwg >> 3, rem = nwg & 7;
    const int vt = flat_tile >= 0 ? it : (xcd < rem ? xcd * (q8 + 1) : rem * (q8 + 1) + (xcd - rem) * q8) + (it >> 3);
    // inside an XCD's range walk groups of kGroupM tile rows column by column: the group's A panels
    // and the current B panel stay in that XCD's L2
    constexpr int kGroupM = 8;
    const int group_sz = kGroupM * g.tiles_n;
    const int gid = vt / group_sz, in_g = vt - gid * group_sz;
    const int first_m = gid * kGroupM;
    const int gm = min(g.tiles_m - first_m, kGroupM);
    m0 = (first_m + in_g % gm) * TM;
    n0 = (in_g / gm) * TN;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const bool isA = w < 2;
      const int h = w & 1;
      const int T = isA ? A_T : B_T;
      const long long ld = isA ? g.lda : g.ldb;
      const int r0 = (isA ? m0 : n0) + h * 128;
      const int rows = isA ? g.M : g.N;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int p = u * 512 + tid;
        if (T == 0) {
          const int row = p >> 3, cl = p & 7;
          const int cg = cl ^ ((row >> 1) & 7);
          int gr = r0 + row;
          gr = gr > rows - 1 ? rows - 1 : gr;      // rows past the edge re-read the last row; never stored
          src_off[w][u] = (unsigned)((long long)gr * ld + kbeg + cg * 8);
        } else {
          const int k = p >> 4, cl = p & 15;
          const int cg = cl ^ swz_t256(k);
          int col = r0 + cg * 8;
          col = col > rows - 8 ? rows - 8 : col;   // columns past the edge re-read the last chunk
          src_off[w][u] = (unsigned)((long long)(kbeg + k) * ld + col);
        }
      }
    }
  };

  auto issue = [&](int which, int tile) {  // which: 0 A0, 1 A1, 2 B0, 3 B1
    const unsigned short* base = which < 2 ? g.A : g.B;
    char* dst = smem + (tile & 1) * kBufBytes + which * kHalfBytes + dma_dst;
    const unsigned k0 = (unsigned)(tile * (which < 2 ? adv_a : adv_b));
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (size_t)(src_off[which][0] + k0)), (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + (size_t)(src_off[which][1] + k0)), (lds_ptr_t)(dst + 8192), 16, 0,
                                     0);
  };
  // K-tiles 0 and 1, all eight half-tiles (both LDS buffers): issued before the previous tile's output stores, so
  // that everything the first K-tile and a half needs is OLDER than those stores in the wave's memory queue
  auto issue_prologue = [&]() {
    issue(0, 0); issue(2, 0); issue(3, 0); issue(1, 0);
    if (nt > 1) { issue(0, 1); issue(2, 1); issue(3, 1); issue(1, 1); }
  };

  f32x4 acc[8][4];
  bf16x8_t a[4][2], b0[2][2], b1[2][2];

#define LOAD_A(H)                                                                              \
  do {                                                                                         \
    const char* hp = buf + (H) * kHalfBytes;                                                   \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)   \
        a[i][ks] = frag256<A_T>(hp, arow + i * 16, ks, lane);                                  \
  } while (0)
#define LOAD_B(DST, H)                                                                         \
  do {                                                                                         \
    const char* hp = buf + (2 + (H)) * kHalfBytes;                                             \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)   \
        DST[j][ks] = frag256<B_T>(hp, brow + j * 16, ks, lane);                                \
  } while (0)
#define MFMA_Q(QI, QJ, BR)                                                                     \
  do {                                                                                         \
    __builtin_amdgcn_s_setprio(1);                                                             \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) _Pragma("unroll") for (int i = 0; i < 4; ++i)   \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                          \
            acc[(QI) * 4 + i][(QJ) * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(         \
                BR[j][ks], a[i][ks], acc[(QI) * 4 + i][(QJ) * 2 + j], 0, 0, 0);                \
    __builtin_amdgcn_s_setprio(0);                                                             \
  } while (0)
  // issue the phase's half-tile when its tile exists; otherwise drain, because the counted wait only
  // bounds what is in flight when every phase issues
#define ISSUE_OR_DRAIN(WHICH, TILE)                                                            \
  do {                                                                                         \
    if ((TILE) < nt) { issue(WHICH, TILE); WAIT_VM(6); } else { WAIT_VM(0); }                  \
  } while (0)
  // The memory counter retires in issue order (loads, LDS-DMA and stores alike), so `vmcnt(N)` means "all but the N
  // youngest are done". The previous tile's s_out output stores sit between this tile's prologue (issued before that
  // tile's epilogue) and the DMA issued inside this tile's first K-tiles: a wait that only needs something OLDER than the stores adds
  // them to its count and leaves them in flight (they drain under the first K-tile and a half instead of in front of
  // the main loop).
#define WAIT_6S()                                                                              \
  do {                                                                                         \
    if (s_out == 0) WAIT_VM(6); else if (s_out == 16) WAIT_VM(22); else if (s_out == 24) WAIT_VM(30); else WAIT_VM(38); \
  } while (0)
#define WAIT_8S()                                                                              \
  do {                                                                                         \
    if (s_out == 0) WAIT_VM(8); else if (s_out == 16) WAIT_VM(24); else if (s_out == 24) WAIT_VM(32); else WAIT_VM(40); \
  } while (0)
#define ISSUE_OR_DRAIN_S(WHICH, TILE)                                                          \
  do {                                                                                         \
    if ((TILE) < nt) { issue(WHICH, TILE); WAIT_6S(); } else { WAIT_VM(0); }                   \
  } while (0)

  int it = flat_tile >= 0 ? flat_tile : (int)blockIdx.x;
  const bool stamp = g.stamps != nullptr && tid == 0;
  // Workgroups of equal work that start together also store together (measurement hook: the first round of
  // workgroups, one per CU, starts in four groups g.stagger ticks apart; no effect measured)
  if (g.stagger > 0) {
    const int lin = blockIdx.x + blockIdx.y * gridDim.x;
    const int grp = (lin >> 3) & 3;
    if (lin < kCusStagger && grp) {
      const long long until = wall_clock64() + (long long)grp * g.stagger;
      while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
  }
  int s_out = 0;     // output stores of the previous tile that may still be in flight (per wave: 0, 16, 24 or 32)
  if (it < nwg) { setup_tile(it); issue_prologue(); }
  const int it_step = flat_tile >= 0 ? nwg : (int)gridDim.x;     // a split-K workgroup owns one tile
  for (; it < nwg; it += it_step) {
    const int cm0 = m0, cn0 = n0;
    const int next = it + it_step;
    if (stamp) g.stamps[(split * nwg + it) * 4 + 0] = wall_clock64();
    // K-tile 0 has landed; K-tile 1 and the previous tile's stores may still be in flight
    if (nt > 1) WAIT_8S(); else WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // the two row groups run one barrier apart
    if (stamp) g.stamps[(split * nwg + it) * 4 + 1] = wall_clock64();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int t = 0; t < nt; ++t) {
      const char* buf = smem + (t & 1) * kBufBytes;
      // phase 0: quadrant (A0, B0)
      LOAD_B(b0, 0);
      __builtin_amdgcn_sched_barrier(0);
      LOAD_A(0);
      if (t == 0) {
        // B1 of K-tile 1 came with the prologue, and the wait at the top covered this K-tile
      } else if (t == 1) {
        ISSUE_OR_DRAIN_S(3, t + 1);
        s_out = 0;                      // the next wait (A0 of K-tile 2, younger than the stores) retires them
      } else {
        ISSUE_OR_DRAIN(3, t + 1);
      }
      __builtin_amdgcn_s_barrier();
      WAIT_LGKM0();
      __builtin_amdgcn_sched_barrier(0);
      MFMA_Q(0, 0, b0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // phase 1: quadrant (A0, B1)
      LOAD_B(b1, 1);
      if (t == 0) {
        if (nt > 1) WAIT_6S(); else WAIT_VM(0);     // A1 of K-tile 1 came with the prologue
      } else {
        ISSUE_OR_DRAIN(1, t + 1);
      }
      __builtin_amdgcn_s_barrier();
      WAIT_LGKM0();
      __builtin_amdgcn_sched_barrier(0);
      MFMA_Q(0, 1, b1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // phase 2: quadrant (A1, B1)
      LOAD_A(1);
      if (t == 0) ISSUE_OR_DRAIN_S(0, t + 2); else ISSUE_OR_DRAIN(0, t + 2);
      __builtin_amdgcn_s_barrier();
      WAIT_LGKM0();
      __builtin_amdgcn_sched_barrier(0);
      MFMA_Q(1, 1, b1);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      // phase 3: quadrant (A1, B0), fragments already in registers
      if (t == 0) ISSUE_OR_DRAIN_S(2, t + 2); else ISSUE_OR_DRAIN(2, t + 2);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      MFMA_Q(1, 0, b0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    }
    s_out = 0;
    if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with the other group's last barrier
    WAIT_VM(0);
    __syncthreads();                             // every wave is done reading the operand buffers
    if (stamp) g.stamps[(split * nwg + it) * 4 + 2] = wall_clock64();

    // bias is added here, with ordinary loads that are waited for before the next tile's DMA is issued
    if (g.splits == 1 && g.bias) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int n = cn0 + (c >> 1) * 128 + wc * 32 + (c & 1) * 16 + fq * 4;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (n < g.N) {
          if (g.bias_dtype == 0) {
            bv = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.bias) + n);
          } else {
            const u16x4 tb = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(g.bias) + n);
            bv[0] = bf16_to_f32(tb[0]); bv[1] = bf16_to_f32(tb[1]); bv[2] = bf16_to_f32(tb[2]); bv[3] = bf16_to_f32(tb[3]);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][c] += bv;
      }
    }
    WAIT_VM(0);
    __builtin_amdgcn_sched_barrier(0);

    if (next < nwg) { setup_tile(next); issue_prologue(); }
    __builtin_amdgcn_sched_barrier(0);

    if (g.splits > 1) {
      // fp32 partial slab of this split: 16 bytes per lane (4 consecutive n); reduced by dtf_splitk_reduce
      float* W = g.W + (long long)split * g.M * g.N;
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = cm0 + qi * 128 + wr * 64 + i * 16 + fr;
          if (m >= g.M) continue;
#pragma unroll
          for (int qj = 0; qj < 2; ++qj)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int n = cn0 + qj * 128 + wc * 32 + j * 16 + fq * 4;
              if (n < g.N) *reinterpret_cast<f32x4*>(W + (long long)m * g.N + n) = acc[qi * 4 + i][qj * 2 + j];
            }
        }
      continue;
    }

    // ---- epilogue: activation in fp32, then each lane turns its two 8-byte pieces of a 16-row slice (4 consecutive
    // columns of the two 16-column groups) into ONE 16-byte piece by a row swap with its neighbour (v_permlane16_swap:
    // even 16-lane rows keep the first group and receive the neighbour's first group, odd rows the second), so that a
    // store instruction writes 16 rows of 64 contiguous bytes straight from registers: no staging through LDS, no LDS
    // round trip between the conversion and the store (the staged form spent 3.6 us per tile issuing its 16 stores,
    // each behind its own ds_read and wait). The activation and the accumulate input are compile-time choices of the
    // body: tested per element they cost a branch (and the instruction fetch of the skipped code) 128 times per lane.
    // The epilogue's indices are re-derived from values the compiler cannot trace back past this point: otherwise
    // it hoists the epilogue's address arithmetic above the main loop, where the extra live registers spill.
    int cm0e = cm0, cn0e = cn0, lane_e = lane;
    asm volatile("" : "+s"(cm0e), "+s"(cn0e), "+v"(lane_e));
    const int fr_e = lane_e & 15, fq_e = lane_e >> 4;
    const int col_e = (fq_e & 1) * 16 + (fq_e >> 1) * 8;     // first of the lane's 8 columns inside the wave's 32
    auto epilogue = [&](auto act_c, auto cin_c, auto cs_c) {
      constexpr int ACT = decltype(act_c)::value;     // 0 none, 1 relu, 2 gelu, 3 gelu + its derivative into C2
      constexpr int CIN = decltype(cin_c)::value;     // 0 none, 1 add Cin, 2 multiply by Cin
      // column sums of the stored (rounded) result: the bias gradient of the layer whose output gradient this product
      // is, summed per wave over its 64 rows of a quadrant and written as one partial row per (tile row, half, wave
      // row); a finishing kernel adds the partial rows. Saves a pass over the result.
      constexpr bool CS = decltype(cs_c)::value;
      // The combine input comes in by LDS-DMA, all 16 KiB of the wave's four quadrants at once, into the operand
      // buffers the main loop no longer needs: one exposed fetch latency instead of one per quadrant, and no
      // registers (holding even one quadrant ahead in registers spills). Only when no next tile's prologue
      // owns those buffers, i.e. one workgroup per tile; the persistent grid fetches a quadrant's four pieces with
      // ordinary loads before that quadrant's first store.
      // Full tiles only (no clamping of edge addresses): a wave-uniform base per 16-row slice plus ONE per-lane
      // 32-bit offset, so the 16 fetches cost no address registers either.
      const bool cin_dma = CIN != 0 && next >= nwg && it_step >= nwg && cm0e + TM <= g.M && cn0e + TN <= g.N;
      const char* cbuf = smem + wid * 16384 + fr_e * 64 + (col_e >> 3) * 16;
      if (CIN != 0 && cin_dma) {
        __syncthreads();          // every wave has read its last operand fragments
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        const unsigned short* wbase = g.Cin + (long long)(cm0e + (wu >> 2) * 64) * g.ldc + cn0e + (wu & 3) * 32;
        const unsigned lane_off = (unsigned)(lane_e >> 2) * (unsigned)g.ldc + (unsigned)(lane_e & 3) * 8u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const unsigned short* sbase = wbase + (long long)((q >> 1) * 128 + v * 16) * g.ldc + (q & 1) * 128;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(sbase + lane_off),
                                             (lds_ptr_t)(smem + wu * 16384 + q * 4096 + v * 1024), 16, 0, 0);
          }
        WAIT_VM(0);
      }
#pragma unroll
      for (int qi = 0; qi < 2; ++qi)
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
          const int n = cn0e + qj * 128 + wc * 32 + col_e;
          const int mq = cm0e + qi * 128 + wr * 64 + fr_e;
          const long long off0 = (long long)mq * g.ldc + n;
          u32x4 ci[4];
          float cs[CS ? 8 : 1];
          if (CS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[CS ? e : 0] = 0.f;
          }
          if (CIN != 0 && !cin_dma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ci[i] = u32x4{0u, 0u, 0u, 0u};
              if (mq + i * 16 < g.M && n < g.N) ci[i] = *reinterpret_cast<const u32x4*>(g.Cin + off0 + (long long)i * 16 * g.ldc);
            }
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            u32x4 val, der;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x4 v = acc[qi * 4 + i][qj * 2 + j];
              if (ACT == 3) {
                const f32x4 lo = gelu_pair256(f32x2_256{v[0], v[1]}), hi = gelu_pair256(f32x2_256{v[2], v[3]});
                v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                der[j * 2 + 0] = pack_bf16x2(lo[2], lo[3]);
                der[j * 2 + 1] = pack_bf16x2(hi[2], hi[3]);
              } else if (ACT == 2) {
                const f32x4 lo = gelu_pair256(f32x2_256{v[0], v[1]}), hi = gelu_pair256(f32x2_256{v[2], v[3]});
                v = f32x4{lo[0], lo[1], hi[0], hi[1]};
              } else if (ACT == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
              }
              val[j * 2 + 0] = pack_bf16x2(v[0], v[1]);
              val[j * 2 + 1] = pack_bf16x2(v[2], v[3]);
            }
            // {group 0 | group 1} per lane -> 8 consecutive columns per lane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const u32x2 sw = __builtin_bit_cast(u32x2, __builtin_amdgcn_permlane16_swap(val[h], val[2 + h], false, false));
              const unsigned s0 = sw[0], s1 = sw[1];
              val[h] = s0; val[2 + h] = s1;
              if (ACT == 3) {
                const u32x2 sd = __builtin_bit_cast(u32x2, __builtin_amdgcn_permlane16_swap(der[h], der[2 + h], false, false));
                const unsigned d0 = sd[0], d1 = sd[1];
                der[h] = d0; der[2 + h] = d1;
              }
            }
            const bool inside = mq + i * 16 < g.M && n < g.N && g.stagger != -1;
            if (CIN != 0) {
              // out = bf16(bf16(acc) (+|*) C_in): the rounding of an unfused GEMM followed by an elementwise kernel
              const u32x4 cv = cin_dma ? *reinterpret_cast<const u32x4*>(cbuf + (qi * 2 + qj) * 4096 + i * 1024) : ci[i];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float a0 = __uint_as_float(val[e] << 16), a1 = __uint_as_float(val[e] & 0xffff0000u);
                const float c0 = __uint_as_float(cv[e] << 16), c1 = __uint_as_float(cv[e] & 0xffff0000u);
                const float lo = CIN == 1 ? a0 + c0 : a0 * c0;
                const float hi = CIN == 1 ? a1 + c1 : a1 * c1;
                val[e] = pack_bf16x2(lo, hi);
              }
            }
            if (inside) {
              *reinterpret_cast<u32x4*>(g.C + off0 + (long long)i * 16 * g.ldc) = val;
              if (ACT == 3) *reinterpret_cast<u32x4*>(g.C2 + off0 + (long long)i * 16 * g.ldc) = der;
              if (CS) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  cs[CS ? 2 * e : 0] += __uint_as_float(val[e] << 16);
                  cs[CS ? 2 * e + 1 : 0] += __uint_as_float(val[e] & 0xffff0000u);
                }
              }
            }
          }
          if (CS) {
            // the 16 lanes of a DPP row hold the 16 rows of a slice for the same 8 columns: rotate-and-add
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float v = cs[CS ? e : 0];
              v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
              v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
              v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
              v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
              cs[CS ? e : 0] = v;
            }
            if (fr_e == 0 && n < g.N) {
              float* dst = g.col_sums + (long long)(((cm0e >> 8) * 2 + qi) * 2 + wr) * g.N + n;
              *reinterpret_cast<f32x4*>(dst) = f32x4{cs[0], cs[CS ? 1 : 0], cs[CS ? 2 : 0], cs[CS ? 3 : 0]};
              *reinterpret_cast<f32x4*>(dst + 4) = f32x4{cs[CS ? 4 : 0], cs[CS ? 5 : 0], cs[CS ? 6 : 0], cs[CS ? 7 : 0]};
            }
          }
        }
    };
    using std::integral_constant;
    using no_cs = std::false_type;
    if (g.Cin) {
      if (g.cin_mode == 2) {
        if (g.col_sums) epilogue(integral_constant<int, 0>{}, integral_constant<int, 2>{}, std::true_type{});
        else epilogue(integral_constant<int, 0>{}, integral_constant<int, 2>{}, no_cs{});
      } else epilogue(integral_constant<int, 0>{}, integral_constant<int, 1>{}, no_cs{});
    } else if (g.act == 0) epilogue(integral_constant<int, 0>{}, integral_constant<int, 0>{}, no_cs{});
    else if (g.act == 1) epilogue(integral_constant<int, 1>{}, integral_constant<int, 0>{}, no_cs{});
    else if (g.act == 2) epilogue(integral_constant<int, 2>{}, integral_constant<int, 0>{}, no_cs{});
    else epilogue(integral_constant<int, 3>{}, integral_constant<int, 0>{}, no_cs{});
    // a full tile issued exactly 16 (32 with the derivative output) stores per wave after the next tile's prologue:
    // the counted waits of that tile's first K-tiles let them fly. An edge tile skips stores lane by lane and wave
    // by wave, so its count is unknown: drain.
    if (next < nwg) {
      if (cm0 + TM <= g.M && cn0 + TN <= g.N && g.stagger != -1) {
        s_out = (!g.Cin && g.act == 3) ? 32 : (g.Cin && g.cin_mode == 2 && g.col_sums) ? 24 : 16;
      } else {
        WAIT_VM(0);
        s_out = 0;
      }
    }
    if (stamp) g.stamps[(split * nwg + it) * 4 + 3] = wall_clock64();
  }
}

}  // namespace

static long long* g_stamps256 = nullptr;
// diagnostic: device buffer of 4 * tiles * splits 64-bit stamps filled by the next launches (null = off)
DTF_API void dtf_gemm256_set_stamps(void* p) { g_stamps256 = (long long*)p; }
static int g_stagger256 = 0;
// measurement hooks: ticks of 10 ns between the four start groups of the first round of workgroups; -1 = the kernel
// without its output stores (what the stores cost: 17 % of a 65536 x 3072 x 768 product)
DTF_API void dtf_gemm256_set_stagger(int ticks) { g_stagger256 = ticks; }
// persistent grid for launches without split-K (slabs: one workgroup per tile and split); DTF_GEMM256_PERSISTENT=0
// restores one workgroup per tile for A/B runs
static int g_persist256 = [] {
  const char* e = getenv("DTF_GEMM256_PERSISTENT");
  return e && e[0] == '0' ? 0 : 1;
}();
constexpr int kCus256 = 256;

// 1: persistent grid (one workgroup per CU walks several tiles), 0: one workgroup per tile (measurement hook)
DTF_API void dtf_gemm256_set_persistent(int on) { g_persist256 = on; }

// defined in gemm.hip: sums fp32 slabs [splits][M][N] into C with bias / activation
extern "C" int dtf_splitk_reduce(const float* W, void* C, const void* bias, int bias_dtype, int act, int out_fp32,
                                 int splits, int M, int N, long long ldc, hipStream_t stream);

// C (bf16, row-major, ldc) = act(op(A) * op(B) + bias). layout 0 NT: A[M,K] B[N,K]; 1 NN: A[M,K] B[K,N];
// 2 TN: A[K,M] B[K,N]. Requirements: K % 64 == 0, N % 8 == 0 (M % 8 == 0 for TN), leading dimensions % 8 == 0,
// 16-byte aligned bases, at least 8 rows / columns per operand. act: 0 none, 1 relu, 2 gelu(erf), 3 gelu(erf)
// with gelu'(pre-activation) written to c2 (laid out like C; what the backward pass multiplies by).
// splits > 1: workspace holds splits * M * N floats. cin: optional bf16 tensor laid out like C, added to
// (cin_mode 1) or multiplied into (cin_mode 2) the result. Returns -9 when the shape does not qualify (the
// caller then uses dtf_gemm_bf16).
// col_sums (only with cin_mode 2, a full-width result: N % 8 == 0): fp32 [4 * ceil(M / 256)][N] partial rows of the
// column sums of the stored result (rows past M contribute nothing); dtf_col_sum_finish adds them.
DTF_API int dtf_gemm256_cs(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                           long long lda, long long ldb, long long ldc, int bias_dtype, int act, int splits,
                           void* workspace, const void* cin, int cin_mode, void* c2, float* col_sums,
                           hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (K % TK || N % 8 || lda % 8 || ldb % 8 || ldc % 8 || M < 8 || N < 8) return -9;
  if (layout == 2 && M % 8) return -9;
  // operand element offsets are kept in 32 bits inside the kernel
  const long long a_elems = layout == 2 ? (long long)K * lda : (long long)M * lda;
  const long long b_elems = layout == 0 ? (long long)N * ldb : (long long)K * ldb;
  if (a_elems >= (1LL << 32) || b_elems >= (1LL << 32)) return -9;
  if (layout < 0 || layout > 2) return -1;
  static bool attr_set = false;
  if (!attr_set) {
    const void* fns[3] = {reinterpret_cast<const void*>(gemm256_kernel<0, 0>),
                          reinterpret_cast<const void*>(gemm256_kernel<0, 1>),
                          reinterpret_cast<const void*>(gemm256_kernel<1, 1>)};
    for (const void* f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsAll);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  if (splits < 1 || cin || act == 3) splits = 1;   // slabs do not carry the accumulate input / the second output
  if (cin && act != 0) return -9;                  // the accumulate epilogue has no activation
  if (cin && cin_mode != 1 && cin_mode != 2) return -1;
  if (act == 3 && (!c2 || splits != 1)) return -9; // the derivative output is written by the in-kernel epilogue only
  if (act < 0 || act > 3) return -1;
  int kps = ((K + splits - 1) / splits + TK - 1) / TK * TK;
  splits = (K + kps - 1) / kps;
  if (splits > 1 && !workspace) return -3;
  G256Args g;
  g.A = (const unsigned short*)A; g.B = (const unsigned short*)B; g.C = (unsigned short*)C; g.bias = bias;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.bias_dtype = bias_dtype; g.act = act;
  g.tiles_m = (M + TM - 1) / TM; g.tiles_n = (N + TN - 1) / TN;
  g.W = (float*)workspace; g.k_per_split = kps; g.splits = splits; g.Cin = (const unsigned short*)cin;
  g.cin_mode = cin ? cin_mode : 0; g.C2 = (unsigned short*)c2;
  g.stamps = g_stamps256;
  g.col_sums = (cin && cin_mode == 2) ? col_sums : nullptr;
  if (col_sums && !g.col_sums) return -9;
  g.stagger = g_stagger256;
  // persistent: one workgroup per CU (the kernel takes the whole LDS), each walking its share of tiles
  const int nwg = g.tiles_m * g.tiles_n;
  int gx = g_persist256 && splits == 1 ? kCus256 : nwg;
  if (gx > nwg) gx = nwg;
  const dim3 grid(splits > 1 ? nwg * splits : gx, 1);
  switch (layout) {
    case 0: hipLaunchKernelGGL((gemm256_kernel<0, 0>), grid, dim3(kT256), kLdsAll, stream, g); break;
    case 1: hipLaunchKernelGGL((gemm256_kernel<0, 1>), grid, dim3(kT256), kLdsAll, stream, g); break;
    default: hipLaunchKernelGGL((gemm256_kernel<1, 1>), grid, dim3(kT256), kLdsAll, stream, g); break;
  }
  if (splits > 1) return dtf_splitk_reduce((const float*)workspace, C, bias, bias_dtype, act, 0, splits, M, N, ldc, stream);
  return (int)hipGetLastError();
}

DTF_API int dtf_gemm256_ex(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                           long long lda, long long ldb, long long ldc, int bias_dtype, int act, int splits,
                           void* workspace, const void* cin, int cin_mode, void* c2, hipStream_t stream) {
  return dtf_gemm256_cs(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, splits, workspace, cin, cin_mode,
                        c2, nullptr, stream);
}

DTF_API int dtf_gemm256(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                        long long lda, long long ldb, long long ldc, int bias_dtype, int act, int splits,
                        void* workspace, const void* cin, hipStream_t stream) {
  return dtf_gemm256_ex(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, splits, workspace, cin, 1,
                        nullptr, stream);
}

DTF_API int dtf_gemm256_nt(const void* A, const void* B, void* C, const void* bias, int M, int N, int K, long long lda,
                           long long ldb, long long ldc, int bias_dtype, int act, hipStream_t stream) {
  return dtf_gemm256(0, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, 1, nullptr, nullptr, stream);
}
