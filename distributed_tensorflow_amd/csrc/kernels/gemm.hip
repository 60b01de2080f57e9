// bf16 MFMA GEMM and implicit-GEMM convolution for gfx950: fused epilogues, three operand
// layouts, split-K, two-level batching, NHWC convolution geometry.
//
//   C[M,N] = act(A x B + bias) (+ C_in)      fp32 accumulate, bf16 or fp32 output
//
// Layout template flags describe how each operand lies in memory relative to the contraction
// index k:
//   A_T = 0 : A[m][k] (k contiguous)        A_T = 1 : A[k][m] (m contiguous)
//   B_T = 0 : B[n][k] (k contiguous)        B_T = 1 : B[k][n] (n contiguous)
// which covers the three products of a dense layer:
//   forward  y  = x  . W^T   : A_T=0, B_T=0   ("NT")
//   dgrad    dx = dy . W     : A_T=0, B_T=1   ("NN")
//   wgrad    dW = dy^T . x   : A_T=1, B_T=1   ("TN")
// These are the MatMul(+BiasAdd+ReLU) ops the reference model executes inside TensorFlow
// (reference initializer.py:16-18, centralized/client.py:12,19).
//
// The CONV template flag turns the same kernel into an implicit-GEMM convolution on NHWC
// activations and [O][KH][KW][I] weights, with no im2col buffer: the operand that is an image
// is gathered pixel by pixel while it is staged (a shifted / strided row address and a bounds
// test per 16-byte load, zero fill outside the image):
//   CONV=1 forward : M = output pixels, K = (kh, kw, ci), A = x gathered,  B = w            (NT)
//   CONV=2 dgrad   : M = input pixels,  K = (kh, kw, co), A = dy gathered, B = w per tap    (NN)
//   CONV=3 wgrad   : per tap (blockIdx.z): M = co, N = ci, K = output pixels,
//                    A = dy, B = x gathered                                                  (TN)
//   CONV=4 wgrad^T : one plane: M = (kh, kw, ci), N = co, K = output pixels,
//                    A = x gathered, B = dy; the result is dW transposed, [KH*KW*CI][CO]      (TN)
//                    -- for narrow layers (CO <= 64): dy is re-read once per 128 taps x channels
//                    instead of once per tap
// so 1x1, 3x3, 7x7, stride 1 or 2 all run on the MFMA main loop below.
//
// Structure (tiled for 64-lane wavefronts, not a warp-shaped port):
//   * 128x128 output tile per 256-thread block, BK = 64; 4 waves as 2x2, each wave owns
//     64x64 = 4x4 MFMA tiles of v_mfma_f32_16x16x32_bf16;
//   * operand tiles are staged global -> registers -> LDS with 16-byte accesses only. A
//     k-contiguous operand becomes a [row][k] image (128-byte rows) read with ds_read_b128; an
//     operand whose k is the slow index keeps its memory order, a [k][row] image (256-byte rows),
//     and its MFMA fragments are read with ds_read_b64_tr_b16, the hardware transposed read, so
//     no register transpose and no extra LDS instructions are spent on layout;
//   * the 16-byte chunk index is XORed with (row >> 1) & 7 in the [row][k] image and with a
//     function of k in the [k][row] image, which makes both kinds of read conflict-free on the
//     64-bank LDS (see swz_off / swz_t);
//   * global loads for tile t+1 are issued before the MFMAs of tile t and written to the other
//     LDS buffer after them: one barrier per k-tile;
//   * MFMAs are issued with swapped operands (B fragment as the A operand), so each lane ends up
//     with 4 consecutive n of one output row; the tile is then staged through LDS and written as
//     whole 128-byte row segments (optionally adding an existing tensor C_in on the way out:
//     the residual-branch gradient accumulation of a ResNet block);
//   * blockIdx is remapped so that the blocks that share an XCD (and its L2) work on
//     neighbouring tiles.
// Requirements checked on the host: 16-byte aligned base pointers, leading dimensions and
// channel counts that are multiples of 8 elements; M and K are otherwise free (edges are
// zero-filled / masked).
#include "dtf_common.h"
#include <type_traits>

using namespace dtf;

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kThreads = 256;
constexpr int kTileBytes = BM * BK * 2;  // 16 KiB per operand tile

// Convolution geometry (NHWC). x: [N][H][W][CI], dy / y: [N][OH][OW][CO], w: [CO][KH][KW][CI].
struct ConvGeom {
  int H, W, OH, OW, S, P, KH, KW, CI, CO;   // S, P: stride and padding along H (and along W unless SW / PW differ)
  int NB;            // images in the batch (row order of the strided data gradient)
  int SW, PW;        // stride and (left) padding along W; forward and weight gradient only
  int fold;          // weight gradient: 1 = one launch plane per kernel ROW, its KW taps side by side in N (N = KW * CI);
                     // 2 = every tap side by side (CONV 4: the gathered operand's columns are (kh, kw, ci))
  float r_ohw, r_ow, r_hw, r_w, r_ci, r_co, r_kw, r_nb;   // reciprocals for the exact float divmod below
};

struct GemmArgs {
  const unsigned short* A;
  const unsigned short* B;
  void* C;               // bf16 or fp32 [M][ldc]; fp32 [splits][batch][M][N] workspace when splits > 1
  const void* bias;      // [N] or null
  const unsigned short* Cin;  // optional bf16 tensor added to the result (same indexing as C)
  const unsigned char* cin_mask;  // optional: one bit per element of Cin (byte (m * ldc + n) / 8); 0 = add nothing
  long long lda, ldb, ldc;
  int M, N, K;
  int k_per_split;       // multiple of BK
  int splits;
  int bias_dtype;        // 0 fp32, 1 bf16
  int act;               // 0 none, 1 relu, 2 gelu(erf)
  int out_fp32;
  int tiles_m, tiles_n;
  // batched GEMM: blockIdx.z = b1 * nb2 + b2; operand offsets b1 * s?1 + b2 * s?2 (elements)
  int nb2;
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  // dropout in the epilogue (after the activation): drop where philox(seed, *off + idx/4) < thresh
  unsigned drop_thresh;  // 0 = no dropout
  float drop_scale;      // 1 / (1 - p)
  unsigned long long drop_seed;
  const unsigned long long* drop_off;
  // optional per-column statistics of the bf16 output for a following batch norm: fp32 [tiles_m][2][N],
  // row tile_m holds sum(c) and sum(c^2) over that tile's 128 output rows
  float* col_stats;
  // stats_mode 1: the output is the gradient dy entering a batch norm whose input was bn_x (laid out like C). The
  // rows then hold sum dy' and sum dy' * (bn_x - mean), dy' = dy masked by the norm's ReLU: bn_relu 0 none,
  // 2 sign of fma(bn_x, bn_scale, bn_shift), 3 one bit per element in bn_mask (byte (m * ldc + n) / 8).
  int stats_mode, bn_relu;
  const unsigned short* bn_x;
  const float* bn_mean;
  const float* bn_scale;
  const float* bn_shift;
  const unsigned char* bn_mask;
  // up2 (plain products only): row m = (n, a, b) of an [NB][H/2][W/2] grid is stored at pixel (n, 2a, 2b) of an
  // [NB][H][W] image and zeros at its three neighbours (2a, 2b+1), (2a+1, 2b), (2a+1, 2b+1): the data gradient of a
  // 1x1 convolution of stride 2 as one GEMM over the pixels that were read. H, W in cv.
  int up2;
  // Operand transform (template flag TF: 1 = A operand, 2 = B operand): the operand is the INPUT of a batch norm whose
  // normalised (and rectified) output this product consumes; every element becomes bf16(relu(fma(x, tf_scale[c],
  // tf_shift[c]))) between its load and the LDS store, c = its channel. Elements outside the operand (tile edges,
  // convolution padding) stay zero: the padding of a convolution pads the ACTIVATION. The norm's apply pass over
  // that tensor (one read, one write) does not run.
  const float* tf_scale;
  const float* tf_shift;
  int tf_relu;
  int tn_shift;      // log2(tiles_n) when tiles_n is a power of two, else -1
  int xcd_splits;    // 1: flat grid, the tiles of a split share an XCD (see the kernel's tile map)
  ConvGeom cv;
};

// q = x / d, r = x % d for 0 <= x < 2^24 with rcp = 1/d: float estimate plus one correction.
__device__ __forceinline__ void divmod(int x, int d, float rcp, int& q, int& r) {
  q = (int)((float)x * rcp);
  r = x - q * d;
  if (r < 0) { --q; r += d; }
  else if (r >= d) { ++q; r -= d; }
}

// Sixteen per-lane values (c0[0..7], c1[0..7]) summed over the 8 lanes {lane ^ 8, lane ^ 16, lane ^ 32 and their
// combinations}, without a trip through LDS and without computing every total in every lane: a DPP rotation inside
// the 16-lane row, then the gfx950 row swap (v_permlane16_swap exchanges the odd rows of its first operand with the
// even rows of its second: swapping c0[e] with c1[e] and adding leaves the total of c0[e] in the even rows and that
// of c1[e] in the odd rows, one swap and one add for two values), then the half swap the same way. A lane ends up
// with out[0..3] = totals of statistic (lane >> 4) & 1, elements 4 h .. 4 h + 3, h = lane >> 5. Each total is
// ((v + v[lane ^ 8]) + the same of lane ^ 16) + the same of lane ^ 32: a fixed order.
__device__ __forceinline__ void sum16_lanes_8_16_32(float (&c0)[8], float (&c1)[8], float (&out)[4]) {
  typedef unsigned int u32x2_s __attribute__((ext_vector_type(2)));
  float h[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c0[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c0[e]), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    c1[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(c1[e]), 0x128, 0xf, 0xf, false));
    const u32x2_s a = __builtin_bit_cast(
        u32x2_s, __builtin_amdgcn_permlane16_swap(__float_as_uint(c0[e]), __float_as_uint(c1[e]), false, false));
    const unsigned a0 = a[0], a1 = a[1];
    h[e] = __uint_as_float(a0) + __uint_as_float(a1);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32x2_s b = __builtin_bit_cast(
        u32x2_s, __builtin_amdgcn_permlane32_swap(__float_as_uint(h[e]), __float_as_uint(h[e + 4]), false, false));
    const unsigned b0 = b[0], b1 = b[1];
    out[e] = __uint_as_float(b0) + __uint_as_float(b1);
  }
}

__device__ __forceinline__ int swz_off(int row, int chunk) {
  return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4);
}
// LDS image of a row-contiguous operand: [k][row] with 256-byte k-rows, stored as loaded. The
// 16-byte chunk index is XORed with swz_t(k) so that the eight k-rows a 32-lane half touches in
// one ds_read_b64_tr_b16 (k&3 = 0..3 for two values of (k>>3)&1) fall on eight different 32-byte
// slots of the 256-byte bank row: conflict-free transposed reads.
__device__ __forceinline__ int swz_t(int k) { return ((k & 3) | (((k >> 3) & 1) << 2)) << 1; }

// Row order of the data-gradient GEMM. Stride 1: m = (n, ih, iw), the memory order. Stride 2: the
// rows are grouped by the parity class (ih & 1, iw & 1) of the input pixel,
//   m = ((class * N + n) * H/2 + ih/2) * W/2 + iw/2,
// because a filter tap (kh, kw) reaches an input pixel only when ih + P - kh and iw + P - kw are
// even: inside one class the set of contributing taps is the same for every row, so a tile that
// lies in one class skips the k-tiles of the other taps (3/4 of them for a 3x3 filter, all of them
// for three of the four classes of a strided 1x1). H and W must be even for stride 2.
__device__ __forceinline__ void dgrad_pixel(const ConvGeom& c, int m, int& n, int& ih, int& iw) {
  int rem;
  if (c.S == 1) {
    divmod(m, c.H * c.W, c.r_hw, n, rem);
    divmod(rem, c.W, c.r_w, ih, iw);
  } else {
    const int h2 = c.H >> 1, w2 = c.W >> 1;
    int q, a, b;
    divmod(m, h2 * w2, 4.f * c.r_hw, q, rem);         // q = class * N + n
    divmod(rem, w2, 2.f * c.r_w, a, b);
    int cls;
    divmod(q, c.NB, c.r_nb, cls, n);                  // (an integer division per lane costs four times this)
    ih = 2 * a + (cls >> 1);
    iw = 2 * b + (cls & 1);
  }
}

// Operand loads go through a buffer descriptor: a 32-bit byte offset per lane instead of a 64-bit address, and a lane
// whose element lies outside the operand (tile edge, padding of a convolution) passes an offset beyond the
// descriptor's range and receives zeros from the hardware's range check -- no zero-initialised registers and no
// divergent branch around the load. Operands are below 4 GiB (checked on the host).
typedef unsigned buf_u32x4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t rs, bool ok, unsigned elem_off) {
  const unsigned off = ok ? elem_off * 2u : 0xffffffffu;
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

// The same load with the LDS as its destination (LDS-DMA, `buffer_load_dwordx4 ... lds`): the 64 lanes of the
// instruction fill 1 KiB at `lds` (a wave-uniform address) in lane order, lanes out of range write zeros. The data
// never passes through registers, and not through the VGPR -> LDS store path either, which moves a 16-byte store at a
// third of the rate the LDS reads run at (and bounds the register-staged main loop on MFMA-heavy shapes).
typedef __attribute__((address_space(3))) void* lds_dst_t;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, bool ok, unsigned elem_off, char* lds) {
  const unsigned off = ok ? elem_off * 2u : 0xffffffffu;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_dst_t)lds, 16, (int)off, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// Operand stagers. Each thread owns 4 x 16-byte chunks of a tile. Per-row geometry that does not
// depend on the k-tile is computed once in init() and kept in registers.
//
// k-contiguous image ([row][k]): chunk ids tid + 256*i -> row = id >> 3, k-chunk = id & 7 (the
// same k-chunk for all 4 chunks of a thread, 32 rows apart).
// row-contiguous image ([k][row]): row-chunk rc = tid & 15 (rows rc*8..+7), k = (tid >> 4)*4 + i.

//
// ROWS = 128 or 64 rows per tile. A 64-row tile has half the chunks (two per thread); its row-contiguous image keeps
// the 256-byte k-rows of the 128-row one (only chunks 0..7 are filled), so both read with the same swizzle.

template <int T, int GATHER, int ROWS, bool TF = false>   // GATHER: 0 plain, 1 conv forward x, 2 conv dgrad dy, 3 conv wgrad x, 4 conv dgrad w
struct Stager {
  static constexpr int NCH = ROWS / 32;    // 16-byte chunks per thread
  static constexpr int RCB = ROWS == 64 ? 3 : 4;   // T = 1: log2(row chunks per k-row)
  u32x4 r[NCH];
  unsigned okmask;   // TF: bit i = chunk i of the tile in r[] holds operand data (not an edge / padding zero)
  int tfc;           // TF: channel of the first element of this thread's chunks in that tile
  int base[NCH];       // element offset of the row's first tap (GATHER 1, 2; see init)
  int ca[NCH], cb[NCH];  // ih0 / iw0 (GATHER 1) or ih+P / iw+P (GATHER 2)
  bool rowok[NCH];
  int tkh, tkw;      // filter tap of this launch plane (GATHER 3)
  int coff;          // GATHER 3: element offset of this thread's 8 columns inside the source pixel

  __device__ __forceinline__ void init(const GemmArgs& g, int rows, int row0, int tap) {
    if (GATHER == 3) {
      const int col = row0 + ((threadIdx.x & ((1 << RCB) - 1)) << 3);
      if (g.cv.fold) {                       // plane = kernel row; column = (kw, ci), or (kh, kw, ci) when fold = 2
        int q, rr;
        divmod(col < rows ? col : 0, g.cv.CI, g.cv.r_ci, q, rr);
        tkh = tap; tkw = q; coff = rr;
        if (g.cv.fold == 2) divmod(q, g.cv.KW, g.cv.r_kw, tkh, tkw);
      } else {
        divmod(tap, g.cv.KW, g.cv.r_kw, tkh, tkw);
        coff = col;
      }
    }
    if (T == 0 && (GATHER == 1 || GATHER == 2)) {
      const ConvGeom& c = g.cv;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int row = row0 + ((threadIdx.x + kThreads * i) >> 3);
        rowok[i] = row < rows;
        const int m = rowok[i] ? row : 0;
        int n, rem, a, b;
        if (GATHER == 1) {                  // m = (n, oh, ow)
          divmod(m, c.OH * c.OW, c.r_ohw, n, rem);
          divmod(rem, c.OW, c.r_ow, a, b);
          ca[i] = a * c.S - c.P; cb[i] = b * c.SW - c.PW;
          base[i] = ((n * c.H + ca[i]) * c.W + cb[i]) * c.CI;    // element offset of tap (0, 0); may be negative
        } else {                            // m = input pixel, see dgrad_pixel()
          dgrad_pixel(c, m, n, a, b);
          ca[i] = a + c.P; cb[i] = b + c.P;
          // stride 1: the source pixel is linear in the tap, so the row part of its offset is fixed here;
          // stride 2: only the image's offset is
          base[i] = c.S == 1 ? ((n * c.OH + ca[i]) * c.OW + cb[i]) * c.CO : n * c.OH * c.OW * c.CO;
        }
      }
    }
  }

  __device__ __forceinline__ void load(const GemmArgs& g, const unsigned short* __restrict__ X, long long ld, int rows,
                                       int row0, int k0, int kend, int tap) {
    const ConvGeom& c = g.cv;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)0xffffffffu, 0x00020000);
    const unsigned uld = (unsigned)ld;
    if (T == 0) {
      const int k = k0 + ((threadIdx.x & 7) << 3);
      const bool kok = k < kend;
      if (TF) { okmask = 0u; tfc = kok ? k : 0; }
      if (GATHER == 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int row = row0 + ((threadIdx.x + kThreads * i) >> 3);
          r[i] = buf_load16(rs, row < rows && kok, (unsigned)row * uld + (unsigned)k);
          if (TF) okmask |= (row < rows && kok ? 1u : 0u) << i;
        }
      } else if (GATHER == 1) {             // k = (kh, kw, ci)
        int t, ci, kh, kw;
        divmod(kok ? k : 0, c.CI, c.r_ci, t, ci);
        divmod(t, c.KW, c.r_kw, kh, kw);
        const int tapo = (kh * c.W + kw) * c.CI + ci;
        if (TF) tfc = ci;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int ih = ca[i] + kh, iw = cb[i] + kw;
          const bool ok = rowok[i] && kok && (unsigned)ih < (unsigned)c.H && (unsigned)iw < (unsigned)c.W;
          r[i] = buf_load16(rs, ok, (unsigned)(base[i] + tapo));
          if (TF) okmask |= (ok ? 1u : 0u) << i;
        }
      } else {                              // GATHER 2: k = (kh, kw, co), source pixel ((ih+P-kh)/S, (iw+P-kw)/S)
        int t, co, kh, kw;
        divmod(kok ? k : 0, c.CO, c.r_co, t, co);
        divmod(t, c.KW, c.r_kw, kh, kw);
        if (c.S == 1) {
          const int tapo = co - (kh * c.OW + kw) * c.CO;
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int th = ca[i] - kh, tw = cb[i] - kw;
            const bool ok = rowok[i] && kok && (unsigned)th < (unsigned)c.OH && (unsigned)tw < (unsigned)c.OW;
            r[i] = buf_load16(rs, ok, (unsigned)(base[i] + tapo));
          }
        } else {                                 // S == 2
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int th = ca[i] - kh, tw = cb[i] - kw;
            const int oh = th >> 1, ow = tw >> 1;
            const bool ok = rowok[i] && kok && th >= 0 && tw >= 0 && !((th | tw) & 1) && oh < c.OH && ow < c.OW;
            r[i] = buf_load16(rs, ok, (unsigned)(base[i] + (oh * c.OW + ow) * c.CO + co));
          }
        }
      }
    } else {
      const int rc = threadIdx.x & ((1 << RCB) - 1), kg = threadIdx.x >> RCB;
      const int row = row0 + (rc << 3);
      const bool rok = row < rows;
      const int kf = k0 + kg * NCH;         // this thread's first k of the tile; k = kf + i
      // one divmod per tile; the following k are reached by increment-with-carry
      int t = 0, co = 0, n = 0, oh = 0, ow = 0;
      if (TF) { okmask = 0u; tfc = GATHER == 3 ? coff : (rok ? row : 0); }
      if (GATHER == 4) divmod(kf < kend ? kf : 0, c.CO, c.r_co, t, co);
      if (GATHER == 3) {
        int rem;
        divmod(kf < kend ? kf : 0, c.OH * c.OW, c.r_ohw, n, rem);
        divmod(rem, c.OW, c.r_ow, oh, ow);
      }
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int k = kf + i;
        const bool ok = rok && k < kend;
        if (GATHER == 0) {
          r[i] = buf_load16(rs, ok, (unsigned)k * uld + (unsigned)row);
          if (TF) okmask |= (ok ? 1u : 0u) << i;
        } else if (GATHER == 4) {           // weights for dgrad: k = (tap, co) -> w[co][tap][ci = row]
          r[i] = buf_load16(rs, ok, (unsigned)co * uld + (unsigned)(t * c.CI + row));
        } else {                            // GATHER 3: k = output pixel, tap fixed: x[(n, oh*S+kh-P, ow*S+kw-P)][row]
          const int ih = oh * c.S + tkh - c.P, iw = ow * c.SW + tkw - c.PW;
          const bool in = ok && (unsigned)ih < (unsigned)c.H && (unsigned)iw < (unsigned)c.W;
          r[i] = buf_load16(rs, in, (unsigned)(((n * c.H + ih) * c.W + iw) * c.CI + coff));
          if (TF) okmask |= (in ? 1u : 0u) << i;
        }
        if (GATHER == 4) { if (++co == c.CO) { co = 0; ++t; } }
        if (GATHER == 3) { if (++ow == c.OW) { ow = 0; if (++oh == c.OH) { oh = 0; ++n; } } }
      }
    }
  }

  // The tile by LDS-DMA (see dma16): no staging registers. A wave's instruction i fills piece w + 4 i ([row][k] image:
  // 8 rows of 128 bytes) or 4 w + i ([k][row] image: 4 k-rows of 256 bytes); the lane-linear image of a piece is the
  // swizzled layout when every lane fetches the chunk that belongs at its position, i.e. the XOR of swz_off / swz_t
  // is applied to the SOURCE index. GATHER 3 (the x operand of a per-tap weight gradient) has no DMA form.
  __device__ __forceinline__ void dma(const GemmArgs& g, const unsigned short* __restrict__ X, long long ld, int rows,
                                      int row0, int k0, int kend, int tap, char* lds) {
    static_assert(GATHER != 3 && !TF, "no LDS-DMA form");
    const ConvGeom& c = g.cv;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(X), 0, (int)0xffffffffu, 0x00020000);
    const unsigned uld = (unsigned)ld;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (T == 0) {
      // rows as in init(): row0 + ((tid + 256 i) >> 3); LDS slot tid & 7 of that row holds k-chunk slot ^ ((row >> 1) & 7)
      const int k = k0 + ((((int)threadIdx.x & 7) ^ (((int)threadIdx.x >> 4) & 7)) << 3);
      const bool kok = k < kend;
      if (GATHER == 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int row = row0 + ((threadIdx.x + kThreads * i) >> 3);
          dma16(rs, row < rows && kok, (unsigned)row * uld + (unsigned)k, lds + (wv + 4 * i) * 1024);
        }
      } else if (GATHER == 1) {
        int t, ci, kh, kw;
        divmod(kok ? k : 0, c.CI, c.r_ci, t, ci);
        divmod(t, c.KW, c.r_kw, kh, kw);
        const int tapo = (kh * c.W + kw) * c.CI + ci;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
          const int ih = ca[i] + kh, iw = cb[i] + kw;
          const bool ok = rowok[i] && kok && (unsigned)ih < (unsigned)c.H && (unsigned)iw < (unsigned)c.W;
          dma16(rs, ok, (unsigned)(base[i] + tapo), lds + (wv + 4 * i) * 1024);
        }
      } else {                              // GATHER 2
        int t, co, kh, kw;
        divmod(kok ? k : 0, c.CO, c.r_co, t, co);
        divmod(t, c.KW, c.r_kw, kh, kw);
        if (c.S == 1) {
          const int tapo = co - (kh * c.OW + kw) * c.CO;
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int th = ca[i] - kh, tw = cb[i] - kw;
            const bool ok = rowok[i] && kok && (unsigned)th < (unsigned)c.OH && (unsigned)tw < (unsigned)c.OW;
            dma16(rs, ok, (unsigned)(base[i] + tapo), lds + (wv + 4 * i) * 1024);
          }
        } else {
#pragma unroll
          for (int i = 0; i < NCH; ++i) {
            const int th = ca[i] - kh, tw = cb[i] - kw;
            const int oh = th >> 1, ow = tw >> 1;
            const bool ok = rowok[i] && kok && th >= 0 && tw >= 0 && !((th | tw) & 1) && oh < c.OH && ow < c.OW;
            dma16(rs, ok, (unsigned)(base[i] + (oh * c.OW + ow) * c.CO + co), lds + (wv + 4 * i) * 1024);
          }
        }
      }
    } else {
      // piece 4 w + i = k-rows 16 w + 4 i .. + 3; lane l: k-row + (l >> 4), slot l & 15 holds row chunk slot ^ swz_t(k)
      const int lane = threadIdx.x & 63;
      const int kk0 = 16 * wv + (lane >> 4);
      int t = 0, co = 0;
      if (GATHER == 4) divmod(k0 + kk0 < kend ? k0 + kk0 : 0, c.CO, c.r_co, t, co);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = kk0 + 4 * i;
        const int rc = (lane & 15) ^ swz_t(kk);
        const int row = row0 + (rc << 3);
        const bool ok = rc < ROWS / 8 && row < rows && k0 + kk < kend;
        if (GATHER == 0) dma16(rs, ok, (unsigned)(k0 + kk) * uld + (unsigned)row, lds + (4 * wv + i) * 1024);
        else dma16(rs, ok, (unsigned)co * uld + (unsigned)(t * c.CI + row), lds + (4 * wv + i) * 1024);   // GATHER 4
        if (GATHER == 4) { co += 4; if (co >= c.CO) { co -= c.CO; ++t; } }
      }
    }
  }

  // bf16(relu(fma(x, scale[c], shift[c]))) on the chunks of the tile in r[]; chunks without operand data stay zero
  __device__ __forceinline__ void transform(const GemmArgs& g) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(g.tf_scale + tfc), s1 = *reinterpret_cast<const f32x4*>(g.tf_scale + tfc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(g.tf_shift + tfc), h1 = *reinterpret_cast<const f32x4*>(g.tf_shift + tfc + 4);
    const float sc[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    const float sh[8] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      u32x4 v = r[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = fmaf(__uint_as_float(v[e] << 16), sc[2 * e], sh[2 * e]);
        float hi = fmaf(__uint_as_float(v[e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]);
        if (g.tf_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
        v[e] = pack_bf16x2(lo, hi);
      }
      const bool ok = (okmask >> i) & 1u;
      r[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
  }

  __device__ __forceinline__ void store(char* lds) const {
    if (T == 0) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int id = threadIdx.x + kThreads * i;
        *reinterpret_cast<u32x4*>(lds + swz_off(id >> 3, id & 7)) = r[i];
      }
    } else {
      const int rc = threadIdx.x & ((1 << RCB) - 1), kg = threadIdx.x >> RCB;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int k = kg * NCH + i;
        *reinterpret_cast<u32x4*>(lds + k * 256 + ((rc ^ swz_t(k)) << 4)) = r[i];
      }
    }
  }
};

// MFMA operand fragment (row = row0 + (lane & 15), k = ks*32 + 8*(lane >> 4) + 0..7) of a tile.
// T = 0: [row][k] image, one ds_read_b128. T = 1: [k][row] image, two hardware-transposed reads
// (each returns 4 consecutive k of this lane's row).
template <int T>
__device__ __forceinline__ bf16x8_t load_frag(const char* lds, int row0, int ks, int lane) {
  if (T == 0) {
    const int row = row0 + (lane & 15);
    return *reinterpret_cast<const bf16x8_t*>(lds + swz_off(row, ks * 4 + (lane >> 4)));
  } else {
    const int g = lane >> 4, t = lane & 15, q = t >> 2, p = t & 3;
    const int k = ks * 32 + 8 * g + q;
    const int chunk = (row0 >> 3) + (p >> 1);
    const int off = k * 256 + ((chunk ^ swz_t(k)) << 4) + ((p & 1) << 3);
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr_t;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(lds + off));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr_t)(lds + off + 4 * 256));  // k + 4: same swizzle
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8_t, v);
  }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// V selects the output tile: 0 = 128 x 128 (waves 2 x 2, 64 x 64 each), 1 = 128 x 64 (waves 4 x 1, 32 x 64 each) for
// products at most 64 columns wide (the 64-channel layers of a ResNet, a convolution stem): half the accumulators,
// half the B tile, four resident blocks per CU instead of three.
// EPI != 0: a lean store phase. Bit 0 marks it: no split-K slab, bias, activation, dropout, fp32 output or up2
// placement is compiled in. The other bits say what IS compiled in, unconditionally: 2 = column statistics of the
// output (stats_mode 0), 4 = the accumulate input C_in (its mask stays a run-time test), 8 = the sums of a batch
// norm's backward pass (stats_mode 1). The host launches the variant that matches the request exactly. The general
// store phase (EPI 0) tests every feature per accumulator group and per pass at run time; for the short-contraction
// products of the 1x1 convolutions, whose blocks live a few microseconds, those hundreds of untaken branches were
// 11-13 % of the kernel.
template <int A_T, int B_T, int CONV, int STAGES, int V, int TF = 0, int EPI = 0>
__global__ __launch_bounds__(kThreads, V ? 4 : 3) void gemm_bf16_kernel(GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // STAGES x (A tile, B tile)
  constexpr int BN_ = V ? 64 : 128;          // tile columns
  constexpr int WTM = V ? 32 : 64;           // rows of a wave's sub-tile
  constexpr int MI = WTM / 16;               // MFMA tiles per wave along m
  constexpr int kBTile = (B_T == 1 ? 128 : BN_) * BK * 2;   // bytes of the B image (see Stager)
  constexpr int kStage = kTileBytes + kBTile;

  // ---- block -> tile map: XCD-aware, bijective for any grid size --------------------
  // CONV 3 (weight gradient): the filter taps of one (tile, split) are neighbouring blocks of one XCD,
  // so the dy tile they all read and the overlapping x pixels are served by that XCD's L2.
  const int ntaps = CONV == 3 ? (g.cv.fold ? g.cv.KH : g.cv.KH * g.cv.KW) : 1;
  const int nwg = g.tiles_m * g.tiles_n * ntaps;
  const int orig = blockIdx.x;
  int vt, split = blockIdx.y;
  if (A_T == 1 && g.xcd_splits) {
    // Weight gradient with the contraction split over the pixels: the blocks of ONE split (all tiles, all filter
    // taps) read the same slab of both operands, so they are placed on one XCD, next to each other in dispatch
    // order, and the slab crosses the fabric once instead of once per block. The grid is flat: block L runs on
    // XCD L % 8; XCD x works through splits x, x + 8, ...
    const int i = orig >> 3, sl = i / nwg;
    vt = i - sl * nwg;
    split = sl * 8 + (orig & 7);
    if (split >= g.splits) return;           // the grid is padded to a multiple of 8 splits
  } else {
    const int xcd = orig & 7, q = nwg >> 3, rem = nwg & 7;
    vt = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (orig >> 3);
  }
  const int tile = vt / ntaps;
  const int tap = vt - tile * ntaps;
  int tm = g.tn_shift >= 0 ? tile >> g.tn_shift : tile / g.tiles_n;   // tiles_n is a power of two in most launches
  const int tn = tile - tm * g.tiles_n;
  // Strided data gradient: rows are ordered by parity class and the classes do unequal work (a class
  // is reached by 0..4 of the filter taps), while an XCD owns a contiguous range of tiles. Deal the
  // row tiles of the four classes out round-robin so that every XCD gets the same mix.
  if (CONV == 2 && g.cv.S == 2 && (g.tiles_m & 3) == 0) tm = (tm & 3) * (g.tiles_m >> 2) + (tm >> 2);
  const int m0 = tm * BM, n0 = tn * BN_;
  const int plane = CONV == 3 ? tap : (int)blockIdx.z;
  const int planes = CONV == 3 ? ntaps : (int)gridDim.z;
  // lean variants are launched for a single plane only
  const int b1 = EPI ? 0 : plane / g.nb2, b2 = EPI ? 0 : plane - b1 * g.nb2;
  const unsigned short* Ap = g.A + b1 * g.sA1 + b2 * g.sA2;
  const unsigned short* Bp = g.B + b1 * g.sB1 + b2 * g.sB2;
  const long long c_off = b1 * g.sC1 + b2 * g.sC2;
  const int kbeg = split * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wm = V ? wid : wid >> 1, wn = V ? 0 : wid & 1;
  const int fr = lane & 15, fq = lane >> 4;

  // Lean variants (always at least one k-tile) start the accumulators with the first tile's first MFMA, whose
  // addend is then the constant zero: 16 * MI register clears fewer per wave.
  constexpr bool kPeelZero = EPI != 0 && STAGES != 2;
  f32x4 acc[MI][4];
  if (!kPeelZero) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  constexpr int GA = CONV == 1 ? 1 : CONV == 2 ? 2 : CONV == 4 ? 3 : 0;
  constexpr int GB = CONV == 2 ? 4 : CONV == 3 ? 3 : 0;
  Stager<A_T, GA, BM, TF == 1> sa;
  Stager<B_T, GB, BN_, TF == 2> sb;
  sa.init(g, g.M, m0, tap);
  sb.init(g, g.N, n0, tap);

  // Strided data gradient: when the whole tile lies in one parity class, k-tiles whose filter tap
  // cannot reach that class are skipped (they would multiply zeros). skip_par = class parities + P.
  bool can_skip = false;
  int par_h = 0, par_w = 0;
  if (CONV == 2 && g.cv.S == 2 && (g.cv.CO % BK) == 0) {
    const int per_class = g.cv.NB * (g.cv.H >> 1) * (g.cv.W >> 1);
    const int c_lo = m0 / per_class, c_hi = min(m0 + BM - 1, g.M - 1) / per_class;
    can_skip = c_lo == c_hi;
    par_h = ((c_lo >> 1) + g.cv.P) & 1;
    par_w = ((c_lo & 1) + g.cv.P) & 1;
  }
  // The taps that reach the class, one bit each, and a (tap, k-tile inside the tap) cursor: stepping to the next
  // contributing k-tile is a few scalar adds and shifts. (A division by CO and one by KW per CANDIDATE tile, each
  // emulated through the vector pipe's reciprocal, sat on the critical path of tiles that run two to eight k-tiles:
  // the stride-2 3x3 data gradients of 128 / 256 / 512 channels went from 723 / 516 / 417 to 696 / 457 / 336 us.)
  unsigned long long tapmask = 0ull;
  int kpt = 1, cur_tap = 0, cur_j = 0;
  const int ntaps_all = g.cv.KH * g.cv.KW;
  if (CONV == 2 && can_skip) {
    if (ntaps_all > 64 || kbeg != 0) {
      can_skip = false;
    } else {
      kpt = g.cv.CO / BK;
      for (int kh = 0; kh < g.cv.KH; ++kh)
        for (int kw = 0; kw < g.cv.KW; ++kw)
          if (((kh & 1) == par_h) && ((kw & 1) == par_w)) tapmask |= 1ull << (kh * g.cv.KW + kw);
    }
  }
  auto next_tile = [&](int t) -> int {     // first k-tile index >= t that contributes; t = 0, then the last result + 1
    if (CONV == 2 && can_skip) {
      if (t == 0) { cur_tap = 0; cur_j = 0; }
      else if (++cur_j == kpt) { cur_j = 0; ++cur_tap; }
      if (cur_j == 0)
        while (cur_tap < ntaps_all && !((tapmask >> cur_tap) & 1ull)) ++cur_tap;
      return cur_tap * kpt + cur_j;        // = nt once the taps are used up (K = taps * CO, CO a multiple of BK)
    }
    return t;
  };

  auto compute = [&](const char* la, auto first_tag) {
    constexpr bool kFirst = decltype(first_tag)::value;
    const char* lb = la + kTileBytes;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t a[MI], b[4];
#pragma unroll
      for (int i = 0; i < MI; ++i) a[i] = load_frag<A_T>(la, wm * WTM + i * 16, ks, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = load_frag<B_T>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // swapped operands: result register r of lane l is C[m = i*16 + (l&15)][n = j*16 + (l>>4)*4 + r]
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              b[j], a[i], kFirst && ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
    }
  };
  using Yes = std::true_type;
  using No = std::false_type;

  int t = next_tile(0);
  if constexpr (STAGES == 3) {
    // LDS-DMA, two buffers: the next tile lands in the other buffer while this one is read. One barrier per k-tile:
    // behind it every wave's pieces of tile t are in LDS (each wave waited for its own) and nobody reads the buffer
    // tile t + 1 is about to overwrite (it held tile t - 1).
    if (t < nt) {
      sa.dma(g, Ap, g.lda, g.M, m0, kbeg + t * BK, kend, tap, smem);
      sb.dma(g, Bp, g.ldb, g.N, n0, kbeg + t * BK, kend, tap, smem + kTileBytes);
    }
    int cur = 0;
    bool first = kPeelZero;
    while (t < nt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int tnext = next_tile(t + 1);
      if (tnext < nt) {
        char* nb = smem + (cur ^ 1) * kStage;
        const int k0 = kbeg + tnext * BK;
        sa.dma(g, Ap, g.lda, g.M, m0, k0, kend, tap, nb);
        sb.dma(g, Bp, g.ldb, g.N, n0, k0, kend, tap, nb + kTileBytes);
      }
      if (kPeelZero && first) {
        compute(smem + cur * kStage, Yes{});
        first = false;
      } else {
        compute(smem + cur * kStage, No{});
      }
      cur ^= 1;
      t = tnext;
    }
    __syncthreads();                       // the store phase stages through the operand buffers
  } else if (STAGES == 2) {
    // two LDS buffers (64 KiB, 2 blocks per CU): the next tile is written while this one is read
    if (t < nt) {
      sa.load(g, Ap, g.lda, g.M, m0, kbeg + t * BK, kend, tap);
      sb.load(g, Bp, g.ldb, g.N, n0, kbeg + t * BK, kend, tap);
      sa.store(smem);
      sb.store(smem + kTileBytes);
    }
    __syncthreads();
    int cur = 0;
    while (t < nt) {
      const int tnext = next_tile(t + 1);
      const bool more = tnext < nt;
      if (more) {
        const int k0 = kbeg + tnext * BK;
        sa.load(g, Ap, g.lda, g.M, m0, k0, kend, tap);
        sb.load(g, Bp, g.ldb, g.N, n0, k0, kend, tap);
      }
      compute(smem + cur * kStage, No{});
      if (more) {
        char* na = smem + (cur ^ 1) * kStage;
        sa.store(na);
        sb.store(na + kTileBytes);
      }
      __syncthreads();
      cur ^= 1;
      t = tnext;
    }
  } else {
    // one LDS buffer (32 KiB: residency is then set by the ~156 registers per lane, 3 blocks per CU instead of the 2 that 64 KiB allow): the next tile waits in registers while this one
    // is read. For short-K, bandwidth-bound shapes the extra resident blocks hide more latency than
    // the second buffer does.
    if (t < nt) {
      sa.load(g, Ap, g.lda, g.M, m0, kbeg + t * BK, kend, tap);
      sb.load(g, Bp, g.ldb, g.N, n0, kbeg + t * BK, kend, tap);
    }
    bool first = kPeelZero;
    while (t < nt) {
      if (TF == 1) sa.transform(g);
      if (TF == 2) sb.transform(g);
      sa.store(smem);
      sb.store(smem + kTileBytes);
      __syncthreads();
      const int tnext = next_tile(t + 1);
      if (tnext < nt) {
        const int k0 = kbeg + tnext * BK;
        sa.load(g, Ap, g.lda, g.M, m0, k0, kend, tap);
        sb.load(g, Bp, g.ldb, g.N, n0, k0, kend, tap);
      }
      if (kPeelZero && first) {
        compute(smem, Yes{});
        first = false;
      } else {
        compute(smem, No{});
      }
      __syncthreads();
      t = tnext;
    }
  }

  // ---- epilogue ---------------------------------------------------------------------
  const int wm0 = m0 + wm * WTM, wn0 = n0 + wn * 64;
  if (!EPI && g.splits > 1) {
    // fp32 partial slab [split][batch][M][N]; 16 bytes per lane, 4 consecutive n
    float* W = reinterpret_cast<float*>(g.C) + ((long long)split * planes + plane) * g.M * g.N;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = wm0 + i * 16 + fr;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn0 + j * 16 + fq * 4;
        if (n < g.N) *reinterpret_cast<f32x4*>(W + (long long)m * g.N + n) = acc[i][j];
      }
    }
    return;
  }

  // bias + activation in registers
#pragma unroll
  for (int j = 0; j < (EPI ? 0 : 4); ++j) {
    const int n = wn0 + j * 16 + fq * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias && n < g.N && n + 4 > g.N) {
      // the last columns of a product whose width is not a multiple of 4: element by element, nothing read past N
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < g.N)
          bv[r] = g.bias_dtype == 0 ? reinterpret_cast<const float*>(g.bias)[n + r]
                                    : bf16_to_f32(reinterpret_cast<const unsigned short*>(g.bias)[n + r]);
    } else if (g.bias && n < g.N) {
      if (g.bias_dtype == 0) {
        f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.bias) + n);
        bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
      } else {
        u16x4 t = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(g.bias) + n);
        bv[0] = bf16_to_f32(t[0]); bv[1] = bf16_to_f32(t[1]); bv[2] = bf16_to_f32(t[2]); bv[3] = bf16_to_f32(t[3]);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        if (g.act == 1) v = fmaxf(v, 0.f);
        else if (g.act == 2) v = gelu_erf(v);
        acc[i][j][r] = v;
      }
  }
  if (!EPI && g.drop_thresh) {
    const unsigned long long off = *g.drop_off;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const long long m = wm0 + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn0 + j * 16 + fq * 4;
        unsigned rnd[4];
        philox4x32_10(g.drop_seed, off + (unsigned long long)((m * g.N + n) >> 2), rnd);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = rnd[r] >= g.drop_thresh ? acc[i][j][r] * g.drop_scale : 0.f;
      }
    }
  }

  if (!EPI && g.out_fp32) {
    float* C = reinterpret_cast<float*>(g.C) + c_off;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = wm0 + i * 16 + fr;
      if (m >= g.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = wn0 + j * 16 + fq * 4;
        if (n < g.N) *reinterpret_cast<f32x4*>(C + (long long)m * g.ldc + n) = acc[i][j];
      }
    }
    return;
  }

  // bf16 output: each wave stages its WTM x 64 tile in its own LDS ([row][64 n], 128-byte
  // rows, same XOR swizzle), then writes whole 128-byte row segments.
  char* stage = smem + wid * (WTM * 128);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 o;
      o[0] = pack_bf16x2(acc[i][j][0], acc[i][j][1]);
      o[1] = pack_bf16x2(acc[i][j][2], acc[i][j][3]);
      const int cb = j * 32 + fq * 8;  // byte column inside the 128-byte row
      *reinterpret_cast<u32x2*>(stage + swz_off(row, cb >> 4) + (cb & 8)) = o;
    }
  }
  __syncthreads();
  unsigned short* C = reinterpret_cast<unsigned short*>(g.C) + c_off;
  constexpr bool kLean = EPI != 0;
  const bool has_cin = kLean ? (EPI & 4) != 0 : g.Cin != nullptr;
  const bool stats0 = kLean ? (EPI & 2) != 0 : (g.col_stats && g.stats_mode == 0);
  const bool bn_grad_stats = kLean ? (EPI & 8) != 0 : (g.col_stats && g.stats_mode == 1);
  const unsigned short* Cin = has_cin ? g.Cin + c_off : nullptr;
  // Column statistics for the batch norm that follows (the accumulators are dead by now, so these 16
  // registers do not raise the kernel's peak): a lane sees 8 of its wave's 64 rows for one 8-column chunk.
  float cs0[8], cs1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { cs0[e] = 0.f; cs1[e] = 0.f; }
  // per-channel constants of the lane's fixed 8-column chunk (batch-norm backward statistics only)
  float bmu[8], bsa[8], bsb[8];
  if (bn_grad_stats) {
    const int n = wn0 + (lane & 7) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const bool in = n + e < g.N;
      bmu[e] = in ? g.bn_mean[n + e] : 0.f;
      bsa[e] = in && g.bn_relu == 2 ? g.bn_scale[n + e] : 0.f;
      bsb[e] = in && g.bn_relu == 2 ? g.bn_shift[n + e] : 0.f;
    }
  }
  // chunk of the lane in the tile's column statistics (both store-phase forms)
  const int chunk = lane & 7;
  const int n = wn0 + chunk * 8;
  const bool n_ok = n < g.N;
  if constexpr (kLean) {
    // The lane's chunk of pass p: row wm0 + row0 + 8 p of the output, columns n .. n + 7.
    constexpr int kPasses = WTM / 8;           // 64 lanes x 16 bytes = 8 rows of the sub-tile per pass
    const int row0 = lane >> 3;
    const bool full_m = wm0 + WTM <= g.M;      // uniform: no row of the wave's sub-tile is past the edge
    const bool remap = (CONV == 2 && g.cv.S == 2) || (CONV == 0 && !EPI && g.up2);
    auto row_ok = [&](int p) { return full_m || wm0 + row0 + 8 * p < g.M; };
    // lean variants step a precomputed 64-bit offset from pass to pass; the general kernel, short of registers,
    // multiplies afresh
    const long long off0 = kLean ? (long long)(wm0 + row0) * g.ldc + n : 0, step = kLean ? 8LL * g.ldc : 0;
    auto linear_off = [&](int p) { return kLean ? off0 + p * step : (long long)(wm0 + row0 + 8 * p) * g.ldc + n; };
    // Strided data gradient, lean form (no accumulate input: every pass asks for its offset once, in order): the
    // lane's row of pass 0 as (class, image, a, b) by division, the rows of the later passes -- 8 further each -- by
    // carrying. The store phase of a tile that lives for two to eight k-tiles was half of its vector instructions
    // when every pass divided anew.
    int cur_cls = 0, cur_n = 0, cur_a = 0, cur_b = 0, cur_p = 0;
    if (kLean && CONV == 2 && remap) {
      int ih, iw;
      dgrad_pixel(g.cv, min(wm0 + row0, g.M - 1), cur_n, ih, iw);
      cur_cls = ((ih & 1) << 1) | (iw & 1);
      cur_a = ih >> 1; cur_b = iw >> 1;
    }
    auto elem_off = [&](int p) -> long long {  // elements from C (and C_in) to that chunk
      if (!remap) return linear_off(p);
      const int m = wm0 + row0 + 8 * p;
      long long mr;
      if (kLean && CONV == 2) {
        const int h2 = g.cv.H >> 1, w2 = g.cv.W >> 1;
        while (cur_p < p) {                    // 8 rows on
          ++cur_p;
          cur_b += 8;
          while (cur_b >= w2) {
            cur_b -= w2;
            if (++cur_a == h2) {
              cur_a = 0;
              if (++cur_n == g.cv.NB) { cur_n = 0; ++cur_cls; }
            }
          }
        }
        mr = ((long long)cur_n * g.cv.H + 2 * cur_a + (cur_cls >> 1)) * g.cv.W + 2 * cur_b + (cur_cls & 1);
      } else if (CONV == 2) {                  // class-ordered GEMM row -> pixel row in memory
        int pn, ph, pw;
        dgrad_pixel(g.cv, m, pn, ph, pw);
        mr = ((long long)pn * g.cv.H + ph) * g.cv.W + pw;
      } else {                                 // up2: the even pixel of the 2x2 patch the row stands for
        int pn, rem, a, b;
        divmod(m, (g.cv.H >> 1) * (g.cv.W >> 1), 4.f * g.cv.r_hw, pn, rem);
        divmod(rem, g.cv.W >> 1, 2.f * g.cv.r_w, a, b);
        mr = ((long long)pn * g.cv.H + 2 * a) * g.cv.W + 2 * b;
      }
      return mr * g.ldc + n;
    };
    // Lean variants request everything the passes read before the first of them runs, so that a tile pays each
    // latency once: the accumulate input (global), the norm's input and mask bytes of the first four passes (global;
    // all eight would cost the kernel its third resident block), then the staged rows (LDS). The accumulators are
    // dead by now and their registers take the data. The general kernel, whose run-time feature tests keep more
    // alive, fetches pass by pass (it would lose a resident block to the extra registers).
    constexpr int kAhead = kLean ? kPasses : 1;
    u32x4 cin[kAhead];
    unsigned cin_keep[kAhead];
    auto fetch_cin = [&](int p, u32x4& c, unsigned& keep) {
      c = u32x4{0u, 0u, 0u, 0u};
      keep = 0xffu;
      if (row_ok(p) && n_ok) {
        const long long at = elem_off(p);
        c = *reinterpret_cast<const u32x4*>(Cin + at);
        if (g.cin_mask) keep = g.cin_mask[(c_off + at) >> 3];
      }
    };
    if (kLean && has_cin) {
  #pragma unroll
      for (int p = 0; p < kAhead; ++p) fetch_cin(p, cin[p], cin_keep[p]);
    }
    u32x4 bxv[4];
    unsigned bkeep[4];
    auto fetch_bn = [&](int p0) {
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        bxv[q] = u32x4{0u, 0u, 0u, 0u};
        bkeep[q] = 0xffu;
        if (row_ok(p0 + q) && n_ok) {
          const long long at = c_off + linear_off(p0 + q);
          bxv[q] = *reinterpret_cast<const u32x4*>(g.bn_x + at);
          if (g.bn_relu == 3) bkeep[q] = g.bn_mask[at >> 3];
        }
      }
    };
    if (bn_grad_stats) fetch_bn(0);
    u32x4 staged[kAhead];
    if (kLean) {
  #pragma unroll
      for (int p = 0; p < kAhead; ++p)
        staged[p] = *reinterpret_cast<const u32x4*>(stage + swz_off(row0 + 8 * p, chunk));
    }
  #pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      u32x4 v = kLean ? staged[p % kAhead] : *reinterpret_cast<const u32x4*>(stage + swz_off(row0 + 8 * p, chunk));
      const bool m_ok = row_ok(p);
      if (stats0 && m_ok) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
          cs0[2 * e] += lo; cs1[2 * e] = fmaf(lo, lo, cs1[2 * e]);
          cs0[2 * e + 1] += hi; cs1[2 * e + 1] = fmaf(hi, hi, cs1[2 * e + 1]);
        }
      }
      if (m_ok && n_ok) {
        const long long at = elem_off(p);
        if (CONV == 0 && !EPI && g.up2) {       // zeros for the three other pixels of the patch
          const u32x4 z = {0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(C + at + g.ldc) = z;
          *reinterpret_cast<u32x4*>(C + at + (long long)g.cv.W * g.ldc) = z;
          *reinterpret_cast<u32x4*>(C + at + (long long)(g.cv.W + 1) * g.ldc) = z;
        }
        if (has_cin) {
          // out = bf16(bf16(acc) + C_in): the rounding of an unfused GEMM followed by an add kernel
          if (!kLean) fetch_cin(p, cin[0], cin_keep[0]);
          u32x4 c = cin[p % kAhead];
          if (g.cin_mask) {
            // C_in is a gradient that still has to pass a ReLU: the mask its producer would have applied
            const unsigned keep = cin_keep[p % kAhead];
  #pragma unroll
            for (int e = 0; e < 4; ++e)
              c[e] &= ((keep >> (2 * e)) & 1u ? 0x0000ffffu : 0u) | ((keep >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
          }
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(v[e] << 16) + __uint_as_float(c[e] << 16);
            const float hi = __uint_as_float(v[e] & 0xffff0000u) + __uint_as_float(c[e] & 0xffff0000u);
            v[e] = pack_bf16x2(lo, hi);
          }
        }
        *reinterpret_cast<u32x4*>(C + at) = v;
        if (bn_grad_stats) {
          // v is the finished gradient (accumulate input included): mask it like the norm's backward will
          const u32x4 xv = bxv[p & 3];
          if (g.bn_relu == 2) {        // the mask recomputed from the norm's input (one uniform test per pass)
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xl = __uint_as_float(xv[e] << 16), xh = __uint_as_float(xv[e] & 0xffff0000u);
              float gl = __uint_as_float(v[e] << 16), gh = __uint_as_float(v[e] & 0xffff0000u);
              gl = fmaf(xl, bsa[2 * e], bsb[2 * e]) > 0.f ? gl : 0.f;
              gh = fmaf(xh, bsa[2 * e + 1], bsb[2 * e + 1]) > 0.f ? gh : 0.f;
              cs0[2 * e] += gl; cs1[2 * e] = fmaf(gl, xl - bmu[2 * e], cs1[2 * e]);
              cs0[2 * e + 1] += gh; cs1[2 * e + 1] = fmaf(gh, xh - bmu[2 * e + 1], cs1[2 * e + 1]);
            }
          } else {                     // mask bits (all ones when the norm has no ReLU)
            const unsigned keep = bkeep[p & 3];
  #pragma unroll
            for (int e = 0; e < 4; ++e) {
              // the pair's bits spread to a dword mask, applied to the packed pair before it is unpacked
              const unsigned mlo = (unsigned)__builtin_amdgcn_sbfe((int)keep, 2 * e, 1) & 0x0000ffffu;
              const unsigned mhi = (unsigned)__builtin_amdgcn_sbfe((int)keep, 2 * e + 1, 1) & 0xffff0000u;
              const unsigned gv = v[e] & (mlo | mhi);
              const float xl = __uint_as_float(xv[e] << 16), xh = __uint_as_float(xv[e] & 0xffff0000u);
              const float gl = __uint_as_float(gv << 16), gh = __uint_as_float(gv & 0xffff0000u);
              cs0[2 * e] += gl; cs1[2 * e] = fmaf(gl, xl - bmu[2 * e], cs1[2 * e]);
              cs0[2 * e + 1] += gh; cs1[2 * e + 1] = fmaf(gh, xh - bmu[2 * e + 1], cs1[2 * e + 1]);
            }
          }
        }
      }
      if (p == 3 && kPasses > 4 && bn_grad_stats) fetch_bn(4);
    }
  } else {
    // The general store phase goes pass by pass: read the staged row, fetch what the pass needs, store. Its
    // run-time feature tests keep too much alive for the look-ahead of the lean form.
    // ... and the norm's input (and mask bytes), fetched four passes at a time so that their latency is paid twice
    // per tile rather than per pass (all eight at once would cost the kernel its third resident block)
    u32x4 bxv[4];
    unsigned bkeep[4];
    auto fetch_bn = [&](int p0) {
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int id = lane + 64 * (p0 + q);
        const int m = wm0 + (id >> 3), n = wn0 + (id & 7) * 8;
        bxv[q] = u32x4{0u, 0u, 0u, 0u};
        bkeep[q] = 0xffu;
        if (m < g.M && n < g.N) {
          const long long at = c_off + (long long)m * g.ldc + n;
          bxv[q] = *reinterpret_cast<const u32x4*>(g.bn_x + at);
          if (g.bn_relu == 3) bkeep[q] = g.bn_mask[at >> 3];
        }
      }
    };
    if (bn_grad_stats) fetch_bn(0);
    constexpr int kPasses = WTM / 8;           // 64 lanes x 16 bytes = 8 rows of the sub-tile per pass
  #pragma unroll
    for (int p = 0; p < kPasses; ++p) {
      const int id = lane + 64 * p;
      const int row = id >> 3, chunk = id & 7;
      const int m = wm0 + row, n = wn0 + chunk * 8;
      u32x4 v = *reinterpret_cast<const u32x4*>(stage + swz_off(row, chunk));
      if (stats0 && m < g.M) {
  #pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = __uint_as_float(v[e] << 16), hi = __uint_as_float(v[e] & 0xffff0000u);
          cs0[2 * e] += lo; cs1[2 * e] = fmaf(lo, lo, cs1[2 * e]);
          cs0[2 * e + 1] += hi; cs1[2 * e + 1] = fmaf(hi, hi, cs1[2 * e + 1]);
        }
      }
      if (m < g.M && n < g.N) {
        long long mr = m;
        if (CONV == 2 && g.cv.S == 2) {       // class-ordered GEMM row -> pixel row in memory
          int pn, ph, pw;
          dgrad_pixel(g.cv, m, pn, ph, pw);
          mr = ((long long)pn * g.cv.H + ph) * g.cv.W + pw;
        } else if (CONV == 0 && g.up2) {
          int pn, rem, a, b;
          divmod(m, (g.cv.H >> 1) * (g.cv.W >> 1), 4.f * g.cv.r_hw, pn, rem);
          divmod(rem, g.cv.W >> 1, 2.f * g.cv.r_w, a, b);
          mr = ((long long)pn * g.cv.H + 2 * a) * g.cv.W + 2 * b;
          const u32x4 z = {0u, 0u, 0u, 0u};
          *reinterpret_cast<u32x4*>(C + (mr + 1) * g.ldc + n) = z;
          *reinterpret_cast<u32x4*>(C + (mr + g.cv.W) * g.ldc + n) = z;
          *reinterpret_cast<u32x4*>(C + (mr + g.cv.W + 1) * g.ldc + n) = z;
        }
        if (has_cin) {
          // out = bf16(bf16(acc) + C_in): the rounding of an unfused GEMM followed by an add kernel
          u32x4 c = *reinterpret_cast<const u32x4*>(Cin + mr * g.ldc + n);
          if (g.cin_mask) {
            // C_in is a gradient that still has to pass a ReLU: the mask its producer would have applied
            const unsigned keep = g.cin_mask[(c_off + mr * g.ldc + n) >> 3];
  #pragma unroll
            for (int e = 0; e < 4; ++e)
              c[e] &= ((keep >> (2 * e)) & 1u ? 0x0000ffffu : 0u) | ((keep >> (2 * e + 1)) & 1u ? 0xffff0000u : 0u);
          }
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float lo = __uint_as_float(v[e] << 16) + __uint_as_float(c[e] << 16);
            const float hi = __uint_as_float(v[e] & 0xffff0000u) + __uint_as_float(c[e] & 0xffff0000u);
            v[e] = pack_bf16x2(lo, hi);
          }
        }
        *reinterpret_cast<u32x4*>(C + mr * g.ldc + n) = v;
        if (bn_grad_stats) {
          // v is the finished gradient (accumulate input included): mask it like the norm's backward will
          const u32x4 xv = bxv[p & 3];
          const unsigned keep = bkeep[p & 3];
  #pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xl = __uint_as_float(xv[e] << 16), xh = __uint_as_float(xv[e] & 0xffff0000u);
            float gl = __uint_as_float(v[e] << 16), gh = __uint_as_float(v[e] & 0xffff0000u);
            if (g.bn_relu == 2) {
              gl = fmaf(xl, bsa[2 * e], bsb[2 * e]) > 0.f ? gl : 0.f;
              gh = fmaf(xh, bsa[2 * e + 1], bsb[2 * e + 1]) > 0.f ? gh : 0.f;
            } else {
              gl = (keep >> (2 * e)) & 1u ? gl : 0.f;
              gh = (keep >> (2 * e + 1)) & 1u ? gh : 0.f;
            }
            cs0[2 * e] += gl; cs1[2 * e] = fmaf(gl, xl - bmu[2 * e], cs1[2 * e]);
            cs0[2 * e + 1] += gh; cs1[2 * e + 1] = fmaf(gh, xh - bmu[2 * e + 1], cs1[2 * e + 1]);
          }
        }
      }
      if (p == 3 && kPasses > 4 && bn_grad_stats) fetch_bn(4);
    }
  }
  if (stats0 || bn_grad_stats) {
    // combine the 8 lanes that hold the same chunk (lane ^ 8, 16, 32): fixed order, no atomics. Afterwards a lane
    // holds 4 of the chunk's 16 totals: statistic (lane >> 4) & 1, elements 4 h .. 4 h + 3 with h = lane >> 5
    float t[4];
    sum16_lanes_8_16_32(cs0, cs1, t);
    const int st = (lane >> 4) & 1, hh = lane >> 5;
    const bool owner = !(lane & 8);                          // lane ^ 8 holds the same four values
    // ... then the waves of the block that cover the same columns (those stacked along m), through LDS and in
    // wave order: ONE partial row per 128-row tile, so the buffer and the fold kernel that sums it stay small
    float* red = reinterpret_cast<float*>(smem);            // [4 waves][2][64 columns]
    const int at = st * 64 + chunk * 8 + hh * 4;
    __syncthreads();                                         // every wave is done with its staging area
    if (owner && wm != 0) *reinterpret_cast<f32x4*>(red + wid * 128 + at) = f32x4{t[0], t[1], t[2], t[3]};
    __syncthreads();
    if (owner && wm == 0 && n_ok) {
      constexpr int kStep = V ? 1 : 2, kCount = V ? 4 : 2;   // wave ids of the waves below this one
      f32x4 sum = {t[0], t[1], t[2], t[3]};
#pragma unroll
      for (int q = 1; q < kCount; ++q) sum += *reinterpret_cast<const f32x4*>(red + (wid + q * kStep) * 128 + at);
      *reinterpret_cast<f32x4*>(g.col_stats + ((long long)(m0 / BM) * 2 + st) * g.N + n + hh * 4) = sum;
    }
  }
}

// Sum split-K slabs, add bias, activate, convert. A block owns 16 consecutive float4 outputs;
// its 16 split-lanes each sum every 16th slab (4 loads in flight), then combine through LDS in a
// fixed order, so the result does not depend on scheduling. blockIdx.y = batch plane.
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(const float* __restrict__ W, void* __restrict__ C,
                                                               const void* __restrict__ bias, int bias_dtype, int act,
                                                               int out_fp32, int splits, int M, int N, long long ldc,
                                                               long long c_batch_stride) {
  __shared__ f32x4 red[16][17];
  const int ol = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const long long n4 = (long long)M * (N >> 2);
  const long long plane = (long long)M * N;
  const long long slab = plane * gridDim.y;
  W += (long long)blockIdx.y * plane;
  const long long c_off = (long long)blockIdx.y * c_batch_stride;
  for (long long base = (long long)blockIdx.x * 16; base < n4; base += (long long)gridDim.x * 16) {
    const long long i = base + ol;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int m = 0, n = 0;
    if (i < n4) {
      m = (int)(i / (N >> 2));
      n = (int)(i % (N >> 2)) << 2;
      const float* p = W + (long long)m * N + n;
      int k = sl;
      for (; k + 48 < splits; k += 64) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p + k * slab);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p + (k + 16) * slab);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p + (k + 32) * slab);
        const f32x4 d = *reinterpret_cast<const f32x4*>(p + (k + 48) * slab);
        s += (a + b) + (c + d);
      }
      for (; k < splits; k += 16) s += *reinterpret_cast<const f32x4*>(p + k * slab);
    }
    __syncthreads();
    red[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && i < n4) {
      s = red[0][ol];
#pragma unroll
      for (int r = 1; r < 16; ++r) s += red[r][ol];
      if (bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          s[r] += bias_dtype == 0 ? reinterpret_cast<const float*>(bias)[n + r]
                                  : bf16_to_f32(reinterpret_cast<const unsigned short*>(bias)[n + r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (act == 1) s[r] = fmaxf(s[r], 0.f);
        else if (act == 2) s[r] = gelu_erf(s[r]);
      }
      if (out_fp32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(C) + c_off + (long long)m * ldc + n) = s;
      } else {
        u16x4 o;
        o[0] = f32_to_bf16(s[0]); o[1] = f32_to_bf16(s[1]); o[2] = f32_to_bf16(s[2]); o[3] = f32_to_bf16(s[3]);
        *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(C) + c_off + (long long)m * ldc + n) = o;
      }
    }
  }
}

// Extra, optional parameters of a launch (zero-initialise for a plain GEMM).
struct GemmExtra {
  int nb1, nb2;                      // batch = nb1 * nb2 (0 or 1 = not batched)
  long long sA1, sA2, sB1, sB2, sC1, sC2;
  int conv;                          // 0 plain, 1 conv forward, 2 conv dgrad, 3 conv wgrad
  int H, W, OH, OW, S, P, KH, KW, CI, CO, NB;
  const void* Cin;                   // optional bf16 tensor added to the output
  float drop_p;                      // dropout probability applied after the activation (0 = none)
  int drop_pad;
  unsigned long long drop_seed;
  const void* drop_off;              // device pointer to the 64-bit Philox offset of this call
  void* col_stats;                   // optional fp32 [ceil(M / 128)][2][N]: per-tile column sums of the output
  int stats_mode, bn_relu;           // see GemmArgs: 1 = statistics of a batch norm's incoming gradient
  const void* bn_x;
  const void* bn_mean;
  const void* bn_scale;
  const void* bn_shift;
  const void* bn_mask;
  const void* cin_mask;              // optional ReLU mask bits applied to Cin before it is added
  int SW, PW;                        // stride / left padding along W when they differ from S / P (0, 0 = same; forward
  int use_sw;                        //   and weight gradient only); use_sw = 1 makes them count
  int fold;                          // weight gradient: planes are kernel rows, N = KW * CI (see ConvGeom)
  int tile;                          // 0 = 128 x 128 output tile, 1 = 128 x 64
  int up2;                           // plain NN product stored at every other pixel of an [NB][H][W] image (see GemmArgs)
  const void* tf_scale;              // operand transform (see GemmArgs): fp32 per-channel scale / shift of the batch norm
  const void* tf_shift;              //   whose output this product consumes, applied to the image operand while staging
  int tf_relu;
};

// Sums fp32 split-K slabs [splits][M][N] into C (bias, activation, conversion); shared with gemm256.hip.
DTF_API int dtf_splitk_reduce(const float* W, void* C, const void* bias, int bias_dtype, int act, int out_fp32,
                              int splits, int M, int N, long long ldc, hipStream_t stream) {
  if (!W || !C || splits < 1 || N % 4) return -1;
  const long long n4 = (long long)M * (N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n4, 16), 1), dim3(kBlock), 0, stream, W, C, bias, bias_dtype,
                     act, out_fp32, splits, M, N, ldc, 0LL);
  return (int)hipGetLastError();
}

static int g_xcd_splits = 1;             // split-K weight gradients: tiles of a split on one XCD (measurement hook)
DTF_API void dtf_gemm_set_xcd_splits(int on) { g_xcd_splits = on != 0; }

static int g_lean = 1;                   // lean store phase where it applies (measurement hook: 0 = never,
DTF_API void dtf_gemm_set_lean(int on) { g_lean = on; }   // 1 = everywhere, 2 = plain products only)

static int g_dma_min_tiles = 1 << 20;    // automatic choice of the LDS-DMA main loop: k-tiles per block from which it is
                                         // taken (off by default: see gemm_dma_pays)
DTF_API void dtf_gemm_set_dma_min_tiles(int n) { g_dma_min_tiles = n; }

// Launches that have an LDS-DMA main loop (STAGES 3): the 128x128 tile; plain products in every layout, the
// convolution forward and data gradient. The per-tap weight gradient's gathered x operand and the operand transform
// have no DMA form.
static bool gemm_dma_ok(int layout, int conv, int variant, const GemmArgs& g) {
  if (variant != 0 || g.tf_scale) return false;
  return conv == 0 || (conv == 1 && layout == 0) || (conv == 2 && layout == 1);
}
// ... and where it measured faster than the register-staged loop (tools/bench_gemm_dma.py): it holds two 32 KiB
// buffers, so two blocks per CU instead of three, and its prefetch is one k-tile deep; that pays on long
// contractions of MFMA-bound products (+14-18 % from 2048 deep: 838-948 TFLOP/s at 8192 x 4096 x 4096) and loses on
// the short, memory-bound ones of the 1x1 convolutions. The [k][row] B image of the NN layout gains less: forward-type
// products (NT, convolution forward) from g_dma_min_tiles k-tiles, the other layouts from twice as many. In the
// ResNet-50 step a threshold of 32 changed nothing measurable (74.7 vs 74.4 ms: the 3x3 forward of 256 / 512
// channels gains 5 %, and large dense products already run on the 256x256 kernel), so the automatic choice is off and
// the loop is reached through dtf_gemm_set_stages(3) / dtf_gemm_set_dma_min_tiles.
static bool gemm_dma_pays(int layout, int conv, int k_tiles) {
  const int min_tiles = layout == 0 ? g_dma_min_tiles : 2 * g_dma_min_tiles;
  (void)conv;
  return k_tiles >= min_tiles;
}

template <int AT, int BT, int CV, int E>
static void launch_dma(dim3 grid, size_t lds, hipStream_t stream, const GemmArgs& g) {
  if constexpr (CV <= 2)
    hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 3, 0, 0, E>), grid, dim3(kThreads), lds, stream, g);
}

static int g_stages = 0;                 // 0 = automatic
constexpr int kSingleStageTiles = 1 << 20;   // in effect always: one buffer also measured faster on the long weight-gradient loops

// 0 = choose per launch, 1 = one LDS buffer, 2 = two LDS buffers (measurement hook).
DTF_API void dtf_gemm_set_stages(int stages) { g_stages = stages; }

static int gemm_launch(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                       long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                       int splits, void* workspace, const GemmExtra* ex, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (splits < 1) splits = 1;
  GemmArgs g;
  g.A = (const unsigned short*)A; g.B = (const unsigned short*)B;
  g.bias = bias; g.Cin = nullptr; g.cin_mask = nullptr; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
  g.bias_dtype = bias_dtype; g.act = act; g.out_fp32 = out_fp32;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.drop_thresh = 0; g.drop_scale = 1.f; g.drop_seed = 0; g.drop_off = nullptr;
  g.col_stats = nullptr; g.stats_mode = 0; g.bn_relu = 0;
  g.bn_x = nullptr; g.bn_mean = g.bn_scale = g.bn_shift = nullptr; g.bn_mask = nullptr;
  g.up2 = 0;
  g.tf_scale = g.tf_shift = nullptr; g.tf_relu = 0;
  const int variant = ex && ex->tile == 1 ? 1 : 0;
  const int bn_cols = variant ? 64 : BN;
  g.tiles_n = (N + bn_cols - 1) / bn_cols;
  g.tn_shift = -1;
  for (int sft = 0; sft < 16; ++sft)
    if (g.tiles_n == (1 << sft)) g.tn_shift = sft;
  int batch = 1, conv = 0;
  g.nb2 = 1; g.sA1 = g.sA2 = g.sB1 = g.sB2 = g.sC1 = g.sC2 = 0;
  ConvGeom& c = g.cv;
  c.H = c.W = c.OH = c.OW = c.S = c.KH = c.KW = c.CI = c.CO = c.NB = 1; c.P = 0;
  c.SW = 1; c.PW = 0; c.fold = 0;
  if (ex) {
    const int nb1 = ex->nb1 > 0 ? ex->nb1 : 1;
    g.nb2 = ex->nb2 > 0 ? ex->nb2 : 1;
    batch = nb1 * g.nb2;
    g.sA1 = ex->sA1; g.sA2 = ex->sA2; g.sB1 = ex->sB1; g.sB2 = ex->sB2; g.sC1 = ex->sC1; g.sC2 = ex->sC2;
    g.Cin = (const unsigned short*)ex->Cin;
    g.cin_mask = g.Cin ? (const unsigned char*)ex->cin_mask : nullptr;
    if (ex->drop_p > 0.f) {
      if (!ex->drop_off || ex->drop_p >= 1.f || ex->conv || ex->nb1 > 1 || ex->nb2 > 1) return -8;
      g.drop_thresh = dropout_threshold(ex->drop_p);
      g.drop_scale = 1.f / (1.f - ex->drop_p);
      g.drop_seed = ex->drop_seed;
      g.drop_off = (const unsigned long long*)ex->drop_off;
      splits = 1;                                      // the mask is applied in this kernel's epilogue
    }
    if (ex->col_stats) {
      // whole-output statistics: one plane, plain row order, bf16 output written by this kernel's own epilogue
      if (out_fp32 || batch > 1 || ex->conv >= 3 || (ex->conv == 2 && ex->S != 1) || N % 8 || ex->up2) return -10;
      g.col_stats = (float*)ex->col_stats;
      splits = 1;
      if (ex->stats_mode == 1) {
        if (!ex->bn_x || !ex->bn_mean || (ex->bn_relu == 2 && (!ex->bn_scale || !ex->bn_shift)) ||
            (ex->bn_relu == 3 && !ex->bn_mask) || (ex->bn_relu != 0 && ex->bn_relu != 2 && ex->bn_relu != 3))
          return -10;
        g.stats_mode = 1; g.bn_relu = ex->bn_relu;
        g.bn_x = (const unsigned short*)ex->bn_x; g.bn_mean = (const float*)ex->bn_mean;
        g.bn_scale = (const float*)ex->bn_scale; g.bn_shift = (const float*)ex->bn_shift;
        g.bn_mask = (const unsigned char*)ex->bn_mask;
      }
    }
    conv = ex->conv;
    if (ex->tf_scale) {
      if (!ex->tf_shift || (batch > 1 && conv != 3) || !((layout == 0 && (conv == 0 || conv == 1)) ||
                                           (layout == 2 && (conv == 0 || conv == 3 || conv == 4))))
        return -13;
      // the channel of an operand element is its k index (1x1) or its column (weight gradients): whole 8-channel chunks
      if ((layout == 0 && conv == 0 && K % 8) || (layout == 2 && conv == 0 && N % 8)) return -13;
      g.tf_scale = (const float*)ex->tf_scale; g.tf_shift = (const float*)ex->tf_shift; g.tf_relu = ex->tf_relu;
    }
    if (ex->up2) {
      // rows are the pixels a stride-2 1x1 convolution read: bf16 output written by this kernel's own epilogue
      if (conv || layout != 1 || out_fp32 || batch > 1 || ex->col_stats || ex->Cin || ((ex->H | ex->W) & 1) ||
          (long long)M >= (1 << 24) || (long long)M * 4 != (long long)ex->NB * ex->H * ex->W)
        return -12;
      g.up2 = 1; c.H = ex->H; c.W = ex->W; c.NB = ex->NB;
      splits = 1;
    }
    if (conv) {
      c.H = ex->H; c.W = ex->W; c.OH = ex->OH; c.OW = ex->OW; c.S = ex->S; c.P = ex->P; c.KH = ex->KH; c.KW = ex->KW;
      c.CI = ex->CI; c.CO = ex->CO; c.NB = ex->NB > 0 ? ex->NB : 1;
      c.SW = ex->use_sw ? ex->SW : c.S; c.PW = ex->use_sw ? ex->PW : c.P;
      c.fold = conv == 3 && ex->fold ? 1 : conv == 4 ? 2 : 0;
      if (conv == 4 && (c.CI % 8 || M != c.KH * c.KW * c.CI || batch > 1)) return -4;
      if (c.S != 1 && c.S != 2) return -4;
      if (c.SW != 1 && c.SW != 2) return -4;
      if (conv == 2 && (c.SW != c.S || c.PW != c.P)) return -4;   // the data gradient keeps square strides
      if (c.fold == 1 && (c.CI % 8 || N != c.KW * c.CI)) return -4;
      if (conv == 2 && c.S == 2 && ((c.H | c.W) & 1)) return -7;   // class-ordered rows need even H, W
      if (conv == 2 && out_fp32) return -6;
      if ((long long)M >= (1 << 24) || (long long)K >= (1 << 24)) return -5;   // float divmod range
    }
  }
  c.r_ohw = 1.f / (float)(c.OH * c.OW); c.r_ow = 1.f / (float)c.OW; c.r_hw = 1.f / (float)(c.H * c.W);
  c.r_w = 1.f / (float)c.W; c.r_ci = 1.f / (float)c.CI; c.r_co = 1.f / (float)c.CO; c.r_kw = 1.f / (float)c.KW;
  c.r_nb = 1.f / (float)c.NB;
  if (g.Cin && out_fp32) return -6;
  {
    // operands are addressed with 32-bit byte offsets: each must span less than 4 GiB
    const long long lim = 1ll << 31;   // elements
    const long long a_el = conv == 1 || conv == 4 ? (long long)c.NB * c.H * c.W * c.CI
                         : conv == 2 ? (long long)c.NB * c.OH * c.OW * c.CO
                                     : (long long)(layout == 2 ? K : M) * lda;
    const long long b_el = conv == 3 ? (long long)c.NB * c.H * c.W * c.CI : (long long)(layout == 0 ? N : K) * ldb;
    if (a_el >= lim || b_el >= lim) return -11;
  }
  if (g.Cin || (batch > 1 && conv != 3)) splits = 1;   // slabs carry neither C_in nor strided batches
  if (conv == 2 && c.S == 2) splits = 1;               // nor the class-ordered row map of a strided dgrad
  int kps = ((K + splits - 1) / splits + BK - 1) / BK * BK;
  splits = (K + kps - 1) / kps;
  g.k_per_split = kps; g.splits = splits;
  g.C = splits > 1 ? workspace : C;
  if (splits > 1 && !workspace) return -3;
  // the taps of a weight-gradient launch are folded into grid.x (see the kernel's tile map)
  dim3 grid(g.tiles_m * g.tiles_n * (conv == 3 ? batch : 1), splits, conv == 3 ? 1 : batch);
  g.xcd_splits = 0;
  if (g_xcd_splits && layout == 2 && (conv == 0 || conv == 3 || conv == 4) && splits > 1 && grid.x > 1) {
    g.xcd_splits = 1;
    grid = dim3(grid.x * ((splits + 7) / 8 * 8), 1, grid.z);
  }
  // LDS stages: two buffers pay when a block runs a long k-loop at ~2 blocks per CU; short loops
  // (k-tiles per block <= kSingleStageTiles) are latency- and bandwidth-bound and run better with one
  // buffer and twice the resident blocks. g_stages overrides (1 or 2) for A/B measurements.
  const int k_tiles = kps / BK;
  // 3 = LDS-DMA with two buffers (dma_ok: the layouts that have a DMA form, see gemm_dma_ok)
  const bool dma_ok = gemm_dma_ok(layout, conv, variant, g);
  int stages = g.tf_scale ? 1 : g_stages == 1 || g_stages == 2 ? g_stages : (k_tiles <= kSingleStageTiles ? 1 : 2);
  if (dma_ok && (g_stages == 3 || (g_stages == 0 && gemm_dma_pays(layout, conv, k_tiles)))) stages = 3;
  // LDS per stage: A image + B image (a row-contiguous B keeps 256-byte k-rows in either variant)
  const size_t b_tile = (layout == 0 && variant ? 64 : 128) * BK * 2;
  const size_t stage_bytes = kTileBytes + b_tile;
  const size_t lds = (stages >= 2 ? 2 : 1) * stage_bytes;
  // the lean store-phase variant that matches the request exactly (0 = none does: the general kernel)
  int epi = 0;
  // (the 128x128 form of the stride-1 data gradient spills with the look-ahead of the lean store phase and measured
  // slower with it, 469 vs 330 us per call: it keeps the general one. The stride-2 data gradient, whose parity
  // classes run 1-4 filter taps deep and which carries no statistics, takes the plain lean form in both tiles.)
  const bool lean_conv = (conv == 0 && layout != 2) || (g_lean == 1 && conv == 1 && layout == 0) ||
                         (g_lean == 1 && conv == 2 && layout == 1);
  if (g_lean && splits == 1 && !bias && act == 0 && !g.drop_thresh && !out_fp32 && !g.up2 && stages != 2 && lean_conv &&
      !g.tf_scale && batch == 1) {
    epi = 1 | (g.col_stats && g.stats_mode == 0 ? 2 : 0) | (g.Cin ? 4 : 0) | (g.col_stats && g.stats_mode == 1 ? 8 : 0);
    const bool have = conv == 1 ? (epi == 1 || epi == 3)
                    : conv == 2 ? ((epi == 1 && (variant == 1 || c.S == 2)) || (epi == 9 && variant == 1 && c.S == 1))
                    : layout == 0 ? (epi == 1 || epi == 3) : (epi == 1 || epi == 5 || epi == 9 || epi == 13);
    if (!have) epi = 0;
  }
#define GEMM_LEAN(BT, CV, E)                                                                                   \
  do {                                                                                                         \
    if (variant) hipLaunchKernelGGL((gemm_bf16_kernel<0, BT, CV, 1, 1, 0, E>), grid, dim3(kThreads), lds, stream, g); \
    else if (stages == 3) launch_dma<0, BT, CV, E>(grid, lds, stream, g);                                        \
    else if (CV != 2) hipLaunchKernelGGL((gemm_bf16_kernel<0, BT, CV == 2 ? 0 : CV, 1, 0, 0, E>), grid, dim3(kThreads), lds, stream, g); \
  } while (0)
#define GEMM_LAUNCH(AT, BT, CV)                                                                              \
  do {                                                                                                       \
    if (variant) hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 1, 1>), grid, dim3(kThreads), lds, stream, g); \
    else if (stages == 3) launch_dma<AT, BT, CV, 0>(grid, lds, stream, g);                                     \
    else if (stages == 2) hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 2, 0>), grid, dim3(kThreads), lds, stream, g); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 1, 0>), grid, dim3(kThreads), lds, stream, g);     \
  } while (0)
#define GEMM_LAUNCH_TF(AT, BT, CV, TFS)                                                                         \
  do {                                                                                                         \
    if (variant) hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 1, 1, TFS>), grid, dim3(kThreads), lds, stream, g); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<AT, BT, CV, 1, 0, TFS>), grid, dim3(kThreads), lds, stream, g);  \
  } while (0)
  if (g.tf_scale) {
    if (layout == 0 && conv == 0) GEMM_LAUNCH_TF(0, 0, 0, 1);
    else if (layout == 0 && conv == 1) GEMM_LAUNCH_TF(0, 0, 1, 1);
    else if (layout == 2 && conv == 0) GEMM_LAUNCH_TF(1, 1, 0, 2);
    else if (layout == 2 && conv == 3) GEMM_LAUNCH_TF(1, 1, 3, 2);
    else GEMM_LAUNCH_TF(1, 1, 4, 1);
  } else if (epi) {
    switch (conv * 64 + layout * 16 + epi) {
      case 1: GEMM_LEAN(0, 0, 1); break;
      case 3: GEMM_LEAN(0, 0, 3); break;
      case 16 + 1: GEMM_LEAN(1, 0, 1); break;
      case 16 + 5: GEMM_LEAN(1, 0, 5); break;
      case 16 + 9: GEMM_LEAN(1, 0, 9); break;
      case 16 + 13: GEMM_LEAN(1, 0, 13); break;
      case 64 + 1: GEMM_LEAN(0, 1, 1); break;
      case 64 + 3: GEMM_LEAN(0, 1, 3); break;
      case 128 + 16 + 1:
        if (variant) GEMM_LEAN(1, 2, 1);
        else if (stages == 3) launch_dma<0, 1, 2, 1>(grid, lds, stream, g);
        else hipLaunchKernelGGL((gemm_bf16_kernel<0, 1, 2, 1, 0, 0, 1>), grid, dim3(kThreads), lds, stream, g);
        break;
      case 128 + 16 + 9: GEMM_LEAN(1, 2, 9); break;
      default: return -1;
    }
  } else if (conv == 0) {
    switch (layout) {
      case 0: GEMM_LAUNCH(0, 0, 0); break;
      case 1: GEMM_LAUNCH(0, 1, 0); break;
      case 2: GEMM_LAUNCH(1, 1, 0); break;
      default: return -1;
    }
  } else if (conv == 1 && layout == 0) GEMM_LAUNCH(0, 0, 1);
  else if (conv == 2 && layout == 1) GEMM_LAUNCH(0, 1, 2);
  else if (conv == 3 && layout == 2) GEMM_LAUNCH(1, 1, 3);
  else if (conv == 4 && layout == 2) GEMM_LAUNCH(1, 1, 4);
  else return -1;
#undef GEMM_LAUNCH
#undef GEMM_LEAN
#undef GEMM_LAUNCH_TF
  if (splits > 1) {
    const long long n4 = (long long)M * (N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n4, 16), batch), dim3(kBlock), 0, stream,
                       (const float*)workspace, C, bias, bias_dtype, act, out_fp32, splits, M, N, ldc, g.sC2);
  }
  return (int)hipGetLastError();
}

// layout: 0 = NT, 1 = NN, 2 = TN (see the header comment). workspace: fp32, >= splits*batch*M*N floats
// when splits > 1.
DTF_API int dtf_gemm_bf16(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                          long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                          int splits, void* workspace, hipStream_t stream) {
  return gemm_launch(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, out_fp32, splits, workspace,
                     nullptr, stream);
}

// Batched / convolution / accumulate variant; `ex` points to a GemmExtra.
DTF_API int dtf_gemm_bf16_ex(int layout, const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                             long long lda, long long ldb, long long ldc, int bias_dtype, int act, int out_fp32,
                             int splits, void* workspace, const void* ex, hipStream_t stream) {
  return gemm_launch(layout, A, B, C, bias, M, N, K, lda, ldb, ldc, bias_dtype, act, out_fp32, splits, workspace,
                     (const GemmExtra*)ex, stream);
}
