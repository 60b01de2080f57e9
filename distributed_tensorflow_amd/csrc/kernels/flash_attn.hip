// Fused (flash-style) multi-head self-attention for gfx950, head dimension 64, bf16 in / out.
//
// The unfused path (ops/attention.py) materialises the [B, nh, S, S] scores: at BERT-base shapes that is
// 403 MB per layer written and re-read four times forward and backward. Here a workgroup owns 128
// query rows of one (batch, head) -- a wave owns 32 -- and walks the keys in tiles of 64 with an
// online softmax, so scores and probabilities never leave registers.
//
// Register-level layout (all products use v_mfma_f32_16x16x32_bf16 with the operand that carries the
// QUERY as the second operand): S^T = K Q^T leaves in every lane the scores of ONE query (lane & 15)
// for four consecutive keys per 16-key block. That makes
//   * the row maximum / sum a few in-lane operations plus two cross-lane steps (xor 16, 32),
//   * the probabilities, converted to bf16, directly the second operand of O^T = V^T P^T: the lane's
//     8 values for a 32-key step are its 4 keys of block 2s and its 4 keys of block 2s+1; the V
//     fragment is fetched (transposing LDS read) in the same permuted key order, so no data moves
//     between lanes,
//   * the output accumulator O[q = lane & 15][4 consecutive d], which the per-lane rescale factor of
//     the online softmax multiplies directly.
// The forward pass stores the log-sum-exp of every row (base 2, scaled scores); the backward pass
// recomputes the probabilities from it in two kernels without atomics: one owns query rows and
// produces dQ, one owns keys and produces dK and dV.
// Inputs live inside the fused projection buffer qkv[B, S, 3, nh, 64]; an optional additive fp32
// key bias [B, S] masks padding. The reference has no attention model (reference initializer.py:12-21);
// this is the attention of the BERT-base configuration.
#include "dtf_common.h"

using namespace dtf;

namespace {

constexpr int HD = 64;          // head dimension
constexpr int KT = 64;          // keys (or queries) per tile
constexpr int QB = 128;         // rows owned by a workgroup (4 waves x 32)
constexpr float kLog2e = 1.4426950408889634f;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_f __attribute__((ext_vector_type(4)));
typedef short s16x8_f __attribute__((ext_vector_type(8)));

struct AttnArgs {
  const unsigned short* qkv;   // [B, S, 3, nh, 64]
  const float* bias;           // [B, S] additive key bias or null
  unsigned short* out;         // fwd: [B, S, nh, 64]
  float* lse;                  // [B, nh, S] log2-sum-exp of the scaled scores
  const unsigned short* dout;  // bwd: [B, S, nh, 64]
  float* delta;                // bwd: [B, nh, S] rowsum(dO * O): written by the dQ kernel, read by the dK / dV kernel
  unsigned short* dqkv;        // bwd: [B, S, 3, nh, 64]
  int B, S, nh;
  float scale;                 // 1 / sqrt(hd)
  float* col_parts;            // bwd, optional: fp32 [B * S / 32][3 * nh * 64] column sums of dqkv per wave (32 rows)
};

// row image: [64 rows][128 B], 16-byte chunks XOR-swizzled for conflict-free ds_read_b128 fragments
__device__ __forceinline__ int row_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// transposed-read image: [64 k][128 B]; the XOR keeps 32-byte chunk pairs together
__device__ __forceinline__ int tr_sw(int k) { return (((k >> 1) & 1) | (((k >> 2) & 1) << 1)) << 1; }
__device__ __forceinline__ int tr_off(int k, int chunk) { return k * 128 + ((chunk ^ tr_sw(k)) << 4); }

// fragment of a row image: 16 rows (row0..) x 32 contraction elements (ks)
__device__ __forceinline__ bf16x8_t frag_row(const char* img, int row0, int ks, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(img + row_off(row0 + (lane & 15), ks * 4 + (lane >> 4)));
}

// fragment of a k-major image for output columns c0..c0+15 and the 32-wide contraction step ks, in the
// PERMUTED order lane group g: {ks*32 + 4g .. +3} then {ks*32 + 16 + 4g .. +3} (matches pack_p below)
__device__ __forceinline__ bf16x8_t frag_tr(const char* img, int c0, int ks, int lane) {
  const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
  const int chunk = (c0 >> 3) + (p >> 1);
  const int k_lo = ks * 32 + 4 * g + q, k_hi = k_lo + 16;
  typedef __attribute__((address_space(3))) s16x4_f* lp_t;
  const s16x4_f lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + tr_off(k_lo, chunk) + ((p & 1) << 3)));
  const s16x4_f hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(img + tr_off(k_hi, chunk) + ((p & 1) << 3)));
  const s16x8_f v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8_t, v);
}

// two floats -> packed bf16 pair in one v_cvt_pk_bf16_f32 (round to nearest even)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// the bare v_exp_f32: arguments here are <= 0 (or far below), where the denormal fix-up of exp2f() buys nothing
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

// two accumulator blocks (16-key blocks 2s and 2s+1 of one query column) -> the bf16 operand of the next product
__device__ __forceinline__ bf16x8_t pack_p(const f32x4& a, const f32x4& b) {
  u32x4 r;
  r[0] = pack2(a[0], a[1]);
  r[1] = pack2(a[2], a[3]);
  r[2] = pack2(b[0], b[1]);
  r[3] = pack2(b[2], b[3]);
  return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ u32x2 pack_row4(const f32x4& v) {
  u32x2 w;
  w[0] = pack2(v[0], v[1]);
  w[1] = pack2(v[2], v[3]);
  return w;
}

// Reductions over lane ^ 16 and lane ^ 32 through the gfx950 row / half swaps (v_permlane16_swap,
// v_permlane32_swap): plain VALU instructions, no trip through the LDS crossbar. Swapping a value with
// itself returns {rows 0,0,2,2 | rows 1,1,3,3} (16) and {lower half twice | upper half twice} (32).
struct SwapPair { float x, y; };
__device__ __forceinline__ SwapPair swap16(float v) {
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_permlane16_swap(u, u, false, false));
  const unsigned r0 = r[0], r1 = r[1];     // scalars first: a bit_cast of a vector element reads element 0
  return {__uint_as_float(r0), __uint_as_float(r1)};
}
__device__ __forceinline__ SwapPair swap32(float v) {
  const unsigned u = __float_as_uint(v);
  const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_permlane32_swap(u, u, false, false));
  const unsigned r0 = r[0], r1 = r[1];
  return {__uint_as_float(r0), __uint_as_float(r1)};
}
__device__ __forceinline__ float group_max(float v) {   // over the 4 lanes that share lane & 15
  const SwapPair a = swap16(v);
  const SwapPair b = swap32(fmaxf(a.x, a.y));
  return fmaxf(b.x, b.y);
}
__device__ __forceinline__ float group_sum(float v) {
  const SwapPair a = swap16(v);
  const SwapPair b = swap32(a.x + a.y);
  return b.x + b.y;
}

// 64 rows x 64 bf16 of one (batch, head) starting at token t0, component `which` (0 q, 1 k, 2 v) of qkv,
// or of a [B, S, nh, 64] tensor when `src` is one (ld = nh * 64): each thread loads two 16-byte pieces
struct TileRegs { u32x4 v[2]; };
__device__ __forceinline__ TileRegs load_tile(const unsigned short* base, long long ld, int t0, int tid) {
  TileRegs r;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = u * 256 + tid, row = p >> 3, chunk = p & 7;
    r.v[u] = *reinterpret_cast<const u32x4*>(base + (long long)(t0 + row) * ld + chunk * 8);
  }
  return r;
}
__device__ __forceinline__ void store_row_img(char* img, const TileRegs& r, int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = u * 256 + tid, row = p >> 3, chunk = p & 7;
    *reinterpret_cast<u32x4*>(img + row_off(row, chunk)) = r.v[u];
  }
}
__device__ __forceinline__ void store_tr_img(char* img, const TileRegs& r, int tid) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int p = u * 256 + tid, row = p >> 3, chunk = p & 7;
    *reinterpret_cast<u32x4*>(img + tr_off(row, chunk)) = r.v[u];
  }
}


// Column sums of what a wave stores (its 32 rows x 64 columns of dQ, dK or dV, rounded to bf16 like the stores): the
// bias gradient of the fused QKV projection, which then needs no pass over dqkv. v[db][rb] holds, per lane, 4 consecutive
// columns (db * 16 + g * 4 ...) of row rb * 16 + fr: the two row blocks add in the lane, the 16 rows of a block across
// the DPP row (rotate-and-add), and the lane with fr == 0 writes the wave's partial row.
__device__ __forceinline__ void wave_col_sums(const u32x2 (&w)[4][2], float* dst, int fr, int g) {
#pragma unroll
  for (int db = 0; db < 4; ++db) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      t[2 * e] = __uint_as_float(w[db][0][e] << 16) + __uint_as_float(w[db][1][e] << 16);
      t[2 * e + 1] = __uint_as_float(w[db][0][e] & 0xffff0000u) + __uint_as_float(w[db][1][e] & 0xffff0000u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = t[e];
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
      v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
      t[e] = v;
    }
    if (fr == 0) *reinterpret_cast<f32x4*>(dst + db * 16 + g * 4) = t;
  }
}

// ------------------------------------------------------------------------------ forward
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void flash_fwd_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 8192];   // [buf][K row image | V tr image]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wid * 32;
  const long long ld = 3LL * a.nh * HD;
  const unsigned short* qp = a.qkv + ((long long)b * a.S * 3 + 0) * a.nh * HD + h * HD;
  const unsigned short* kp = qp + (long long)a.nh * HD;
  const unsigned short* vp = kp + (long long)a.nh * HD;
  const float c = a.scale * kLog2e;

  // Q fragments: second operand, row = query, straight from global memory
  bf16x8_t qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (long long)(q0 + qb * 16 + fr) * ld + ks * 32 + g * 8);

  f32x4 o[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) o[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-1e30f, -1e30f}, l[2] = {0.f, 0.f};

  const int nt = a.S / KT;
  TileRegs kr = load_tile(kp, ld, 0, tid), vr = load_tile(vp, ld, 0, tid);
  store_row_img(smem, kr, tid);
  store_tr_img(smem + 8192, vr, tid);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* kimg = smem + (t & 1) * 16384;
    const char* vimg = kimg + 8192;
    if (t + 1 < nt) {
      kr = load_tile(kp, ld, (t + 1) * KT, tid);
      vr = load_tile(vp, ld, (t + 1) * KT, tid);
    }
    // S^T = K Q^T : s[jb][qb], lane = (query fr, keys jb*16 + 4g + r)
    f32x4 s[4][2];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) s[jb][qb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const bf16x8_t kf = frag_row(kimg, jb * 16, ks, lane);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          s[jb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s[jb][qb], 0, 0, 0);
      }
    // scale (+ key bias), online softmax per query column
    f32x4 kb[4];
    if (BIAS) {
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)b * a.S + t * KT + jb * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) kb[jb][r] = fmaxf(bv[r], -1e30f) * kLog2e;
      }
    }
    bf16x8_t pf[2][2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = -1e30f;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (BIAS) s[jb][qb][r] = fmaf(s[jb][qb][r], c, kb[jb][r]);
          mx = fmaxf(mx, s[jb][qb][r]);
        }
      mx = group_max(mx);
      if (!BIAS) mx *= c;                  // c > 0: the maximum commutes with the scaling
      const float mn = fmaxf(m[qb], mx);
      const float alpha = fexp2(m[qb] - mn);
      m[qb] = mn;
      float sum = 0.f;
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = BIAS ? fexp2(s[jb][qb][r] - mn) : fexp2(fmaf(s[jb][qb][r], c, -mn));
          s[jb][qb][r] = p;
          sum += p;
        }
      l[qb] = l[qb] * alpha + sum;       // per-lane partial; combined over the lane group at the end
#pragma unroll
      for (int db = 0; db < 4; ++db) o[db][qb] *= alpha;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) pf[qb][ks] = pack_p(s[2 * ks][qb], s[2 * ks + 1][qb]);
    }
    // O^T += V^T P^T
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8_t vf = frag_tr(vimg, db * 16, ks, lane);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          o[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][ks], o[db][qb], 0, 0, 0);
      }
    if (t + 1 < nt) {
      char* nk = smem + ((t + 1) & 1) * 16384;
      store_row_img(nk, kr, tid);
      store_tr_img(nk + 8192, vr, tid);
    }
    __syncthreads();
  }
  // normalise, store O[b, q, h, d] and the log-sum-exp
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float lt = group_sum(l[qb]);
    const float inv = 1.f / lt;
    const int q = q0 + qb * 16 + fr;
    unsigned short* orow = a.out + ((long long)(b * a.S + q) * a.nh + h) * HD;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      *reinterpret_cast<u32x2*>(orow + db * 16 + g * 4) = pack_row4(o[db][qb] * inv);
    }
    if (g == 0) a.lse[((long long)b * a.nh + h) * a.S + q] = m[qb] + log2f(lt);
  }
}

// ------------------------------------------------------------------------------ backward: dQ
// Workgroup = 128 query rows of one (batch, head); loops over key tiles. Per tile: S^T = K Q^T and
// dP^T = V dO^T (key on the accumulator row, query on the lane), P = exp2(S' - lse),
// dS = P * (dP - delta) * scale, dQ^T += K^T dS^T with K fetched by transposing reads.
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void flash_bwd_dq_kernel(AttnArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 3 * 8192];   // [buf][K row | K tr | V row]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * QB + wid * 32;
  const long long ld = 3LL * a.nh * HD, ldo = (long long)a.nh * HD;
  const unsigned short* qp = a.qkv + ((long long)b * a.S * 3) * a.nh * HD + h * HD;
  const unsigned short* kp = qp + (long long)a.nh * HD;
  const unsigned short* vp = kp + (long long)a.nh * HD;
  const unsigned short* dop = a.dout + (long long)b * a.S * ldo + h * HD;
  const float c = a.scale * kLog2e;

  bf16x8_t qf[2][2], dof[2][2];
  float lse[2], dl[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = q0 + qb * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[qb][ks] = *reinterpret_cast<const bf16x8_t*>(qp + (long long)q * ld + ks * 32 + g * 8);
      dof[qb][ks] = *reinterpret_cast<const bf16x8_t*>(dop + (long long)q * ldo + ks * 32 + g * 8);
    }
    lse[qb] = a.lse[((long long)b * a.nh + h) * a.S + q];
    // delta[q] = sum_d dO[q, d] O[q, d]: the lane's 16 columns here, the other three column groups of the query two
    // swaps away; written out for the dK / dV kernel, which runs after this one (no kernel of its own, and dO is
    // read once less)
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(a.out + (long long)b * a.S * ldo + h * HD + (long long)q * ldo +
                                                             ks * 32 + g * 8);
#pragma unroll
      for (int k = 0; k < 8; ++k) part = fmaf((float)of[k], (float)dof[qb][ks][k], part);
    }
    const SwapPair p16 = swap16(part);
    const float half = p16.x + p16.y;
    const SwapPair p32 = swap32(half);
    dl[qb] = p32.x + p32.y;
    if (g == 0) a.delta[((long long)b * a.nh + h) * a.S + q] = dl[qb];
  }
  f32x4 dq[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) dq[db][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nt = a.S / KT;
  TileRegs kr = load_tile(kp, ld, 0, tid), vr = load_tile(vp, ld, 0, tid);
  store_row_img(smem, kr, tid);
  store_tr_img(smem + 8192, kr, tid);
  store_row_img(smem + 16384, vr, tid);
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* kimg = smem + (t & 1) * 24576;
    const char* ktr = kimg + 8192;
    const char* vimg = kimg + 16384;
    if (t + 1 < nt) {
      kr = load_tile(kp, ld, (t + 1) * KT, tid);
      vr = load_tile(vp, ld, (t + 1) * KT, tid);
    }
    // two halves of 32 keys: the S / dP products of the second half are in flight on the matrix pipe while the
    // vector unit turns the first half into dS, and the dQ products of the first half overlap the second's
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      f32x4 s[2][2], dp[2][2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) { s[jj][qb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[jj][qb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const bf16x8_t kf = frag_row(kimg, (hf * 2 + jj) * 16, ks, lane);
          const bf16x8_t vf = frag_row(vimg, (hf * 2 + jj) * 16, ks, lane);
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
            s[jj][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qb][ks], s[jj][qb], 0, 0, 0);
            dp[jj][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dof[qb][ks], dp[jj][qb], 0, 0, 0);
          }
        }
      bf16x8_t dsf[2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          f32x4 kb = {0.f, 0.f, 0.f, 0.f};
          if (BIAS) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + (long long)b * a.S + t * KT + (hf * 2 + jj) * 16 + g * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) kb[r] = fmaxf(bv[r], -1e30f) * kLog2e;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fexp2(fmaf(s[jj][qb][r], c, BIAS ? kb[r] - lse[qb] : -lse[qb]));
            s[jj][qb][r] = p * (dp[jj][qb][r] - dl[qb]);     // dS / scale: the factor is applied to dQ once, at the end
          }
        }
        dsf[qb] = pack_p(s[0][qb], s[1][qb]);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8_t kt = frag_tr(ktr, db * 16, hf, lane);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
          dq[db][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[qb], dq[db][qb], 0, 0, 0);
      }
    }
    if (t + 1 < nt) {
      char* nk = smem + ((t + 1) & 1) * 24576;
      store_row_img(nk, kr, tid);
      store_tr_img(nk + 8192, kr, tid);
      store_row_img(nk + 16384, vr, tid);
    }
    __syncthreads();
  }
  unsigned short* dqp = a.dqkv + ((long long)b * a.S * 3) * a.nh * HD + h * HD;
  u32x2 wq[4][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    unsigned short* row = dqp + (long long)(q0 + qb * 16 + fr) * ld;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      wq[db][qb] = pack_row4(dq[db][qb] * a.scale);
      *reinterpret_cast<u32x2*>(row + db * 16 + g * 4) = wq[db][qb];
    }
  }
  if (a.col_parts)
    wave_col_sums(wq, a.col_parts + ((long long)b * (a.S >> 5) + (q0 >> 5)) * (3 * a.nh * HD) + h * HD, fr, g);
}

// ------------------------------------------------------------------------------ backward: dK, dV
// Workgroup = 128 keys of one (batch, head), a wave 32 of them; loops over query tiles of 64. Here the
// KEY rides on the lane (second operand): S = Q K^T and dP = dO V^T give, per lane, one key and four
// consecutive queries per 16-query block; P and dS are then the second operands of
// dV^T += dO^T P and dK^T += Q^T dS, whose first operands are transposing reads of the dO and Q tiles.
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void flash_bwd_dkdv_kernel(AttnArgs a) {
  // [buf][Q row | Q tr | dO row | dO tr], then [buf][lse of the 64 queries | delta of the 64 queries]
  __shared__ __attribute__((aligned(16))) char smem[2 * 4 * 8192 + 2 * 512];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, k0 = blockIdx.x * QB + wid * 32;
  const long long ld = 3LL * a.nh * HD, ldo = (long long)a.nh * HD;
  const unsigned short* qp = a.qkv + ((long long)b * a.S * 3) * a.nh * HD + h * HD;
  const unsigned short* kp = qp + (long long)a.nh * HD;
  const unsigned short* vp = kp + (long long)a.nh * HD;
  const unsigned short* dop = a.dout + (long long)b * a.S * ldo + h * HD;
  const float* lsep = a.lse + ((long long)b * a.nh + h) * a.S;
  const float* dlp = a.delta + ((long long)b * a.nh + h) * a.S;
  const float c = a.scale * kLog2e;

  bf16x8_t kf[2][2], vf[2][2];
  float kbias[2] = {0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const int k = k0 + kb * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      kf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(kp + (long long)k * ld + ks * 32 + g * 8);
      vf[kb][ks] = *reinterpret_cast<const bf16x8_t*>(vp + (long long)k * ld + ks * 32 + g * 8);
    }
    if (BIAS) kbias[kb] = fmaxf(a.bias[(long long)b * a.S + k], -1e30f) * kLog2e;
  }
  f32x4 dk[4][2], dv[4][2];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) { dk[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[db][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int nt = a.S / KT;
  // the per-query statistics ride along with the tiles: threads 0..15 fetch lse, 16..31 delta, four queries each,
  // one tile ahead, so that their (global-memory) latency is hidden like the tiles' own
  char* stat = smem + 2 * 4 * 8192;
  const float* statp = (tid < 16 ? lsep : dlp) + (tid & 15) * 4;
  TileRegs qr = load_tile(qp, ld, 0, tid), dr = load_tile(dop, ldo, 0, tid);
  f32x4 st = {0.f, 0.f, 0.f, 0.f};
  if (tid < 32) st = *reinterpret_cast<const f32x4*>(statp);
  store_row_img(smem, qr, tid);
  store_tr_img(smem + 8192, qr, tid);
  store_row_img(smem + 16384, dr, tid);
  store_tr_img(smem + 24576, dr, tid);
  if (tid < 32) *reinterpret_cast<f32x4*>(stat + tid * 16) = st;
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const char* qimg = smem + (t & 1) * 32768;
    const char* qtr = qimg + 8192;
    const char* dimg = qimg + 16384;
    const char* dtr = qimg + 24576;
    if (t + 1 < nt) {
      qr = load_tile(qp, ld, (t + 1) * KT, tid);
      dr = load_tile(dop, ldo, (t + 1) * KT, tid);
      if (tid < 32) st = *reinterpret_cast<const f32x4*>(statp + (t + 1) * KT);
    }
    const char* stc = stat + (t & 1) * 512;
    // two halves of 32 queries (see the dQ kernel); s[ii][kb]: lane = (key fr of block kb, queries
    // t*64 + (2 hf + ii)*16 + 4g + r)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      f32x4 s[2][2], dp[2][2];
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) { s[ii][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[ii][kb] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const bf16x8_t qfr = frag_row(qimg, (hf * 2 + ii) * 16, ks, lane);
          const bf16x8_t dfr = frag_row(dimg, (hf * 2 + ii) * 16, ks, lane);
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            s[ii][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr, kf[kb][ks], s[ii][kb], 0, 0, 0);
            dp[ii][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr, vf[kb][ks], dp[ii][kb], 0, 0, 0);
          }
        }
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(stc + ((hf * 2 + ii) * 16 + g * 4) * 4);
        const f32x4 de = *reinterpret_cast<const f32x4*>(stc + 256 + ((hf * 2 + ii) * 16 + g * 4) * 4);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = fexp2(fmaf(s[ii][kb][r], c, BIAS ? kbias[kb] - ls[r] : -ls[r]));
            s[ii][kb][r] = p;
            dp[ii][kb][r] = p * (dp[ii][kb][r] - de[r]);      // dS / scale, applied to dK at the end
          }
      }
      bf16x8_t pf[2], dsf[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        pf[kb] = pack_p(s[0][kb], s[1][kb]);
        dsf[kb] = pack_p(dp[0][kb], dp[1][kb]);
      }
#pragma unroll
      for (int db = 0; db < 4; ++db) {
        const bf16x8_t dot = frag_tr(dtr, db * 16, hf, lane);
        const bf16x8_t qt = frag_tr(qtr, db * 16, hf, lane);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          dv[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pf[kb], dv[db][kb], 0, 0, 0);
          dk[db][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dk[db][kb], 0, 0, 0);
        }
      }
    }
    if (t + 1 < nt) {
      char* nq = smem + ((t + 1) & 1) * 32768;
      store_row_img(nq, qr, tid);
      store_tr_img(nq + 8192, qr, tid);
      store_row_img(nq + 16384, dr, tid);
      store_tr_img(nq + 24576, dr, tid);
      if (tid < 32) *reinterpret_cast<f32x4*>(stat + ((t + 1) & 1) * 512 + tid * 16) = st;
    }
    __syncthreads();
  }
  unsigned short* dkp = a.dqkv + ((long long)b * a.S * 3 + 1) * a.nh * HD + h * HD;
  unsigned short* dvp = dkp + (long long)a.nh * HD;
  u32x2 wk[4][2], wv[4][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    const long long off = (long long)(k0 + kb * 16 + fr) * ld;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      wk[db][kb] = pack_row4(dk[db][kb] * a.scale);
      wv[db][kb] = pack_row4(dv[db][kb]);
      *reinterpret_cast<u32x2*>(dkp + off + db * 16 + g * 4) = wk[db][kb];
      *reinterpret_cast<u32x2*>(dvp + off + db * 16 + g * 4) = wv[db][kb];
    }
  }
  if (a.col_parts) {
    float* dst = a.col_parts + ((long long)b * (a.S >> 5) + (k0 >> 5)) * (3 * a.nh * HD) + h * HD;
    wave_col_sums(wk, dst + a.nh * HD, fr, g);
    wave_col_sums(wv, dst + 2 * a.nh * HD, fr, g);
  }
}

bool attn_shape_ok(int B, int S, int nh, int hd) { return B > 0 && nh > 0 && hd == HD && S >= QB && S % QB == 0; }

}  // namespace

// qkv [B, S, 3, nh, 64] bf16 -> out [B, S, nh, 64] bf16, lse [B, nh, S] fp32. bias: fp32 [B, S] or null.
// Returns -9 when the shape is not covered (head dim != 64 or S % 128 != 0).
DTF_API int dtf_flash_attn_fwd(const void* qkv, const float* bias, void* out, float* lse, int B, int S, int nh, int hd,
                               float scale, hipStream_t stream) {
  if (!attn_shape_ok(B, S, nh, hd)) return -9;
  AttnArgs a = {};
  a.qkv = (const unsigned short*)qkv; a.bias = bias; a.out = (unsigned short*)out; a.lse = lse;
  a.B = B; a.S = S; a.nh = nh; a.scale = scale;
  if (bias) hipLaunchKernelGGL(flash_fwd_kernel<true>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(flash_fwd_kernel<false>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

// dqkv [B, S, 3, nh, 64] from dout [B, S, nh, 64], the forward's out and lse; delta: fp32 [B, nh, S] scratch.
extern "C" int dtf_col_sum_finish(float* part, float* out, int n_part, int N, hipStream_t stream);

// col_parts (optional): fp32 [B * S / 32][3 * nh * 64] scratch; col_sums: fp32 [3 * nh * 64] receives the column sums of
// dqkv viewed as [B * S, 3 * nh * 64] (the bias gradient of the fused QKV projection).
DTF_API int dtf_flash_attn_bwd_cs(const void* qkv, const float* bias, const void* out, const float* lse, const void* dout,
                                  float* delta, void* dqkv, int B, int S, int nh, int hd, float scale, float* col_parts,
                                  float* col_sums, hipStream_t stream) {
  if (!attn_shape_ok(B, S, nh, hd)) return -9;
  if ((col_parts != nullptr) != (col_sums != nullptr)) return -1;
  // (delta is computed and written by the dQ kernel, read by the dK / dV kernel launched after it)
  AttnArgs a = {};
  a.qkv = (const unsigned short*)qkv; a.bias = bias; a.lse = const_cast<float*>(lse); a.out = (unsigned short*)out;
  a.dout = (const unsigned short*)dout; a.delta = delta; a.dqkv = (unsigned short*)dqkv;
  a.B = B; a.S = S; a.nh = nh; a.scale = scale; a.col_parts = col_parts;
  if (bias) {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<true>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(flash_bwd_dkdv_kernel<true>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL(flash_bwd_dq_kernel<false>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(flash_bwd_dkdv_kernel<false>, dim3(S / QB, nh, B), dim3(256), 0, stream, a);
  }
  if (col_parts) return dtf_col_sum_finish(col_parts, col_sums, B * (S >> 5), 3 * nh * HD, stream);
  return (int)hipGetLastError();
}

DTF_API int dtf_flash_attn_bwd(const void* qkv, const float* bias, const void* out, const float* lse, const void* dout,
                               float* delta, void* dqkv, int B, int S, int nh, int hd, float scale,
                               hipStream_t stream) {
  return dtf_flash_attn_bwd_cs(qkv, bias, out, lse, dout, delta, dqkv, B, S, nh, hd, scale, nullptr, nullptr, stream);
}
