// Row-wise kernels of the transformer (BERT) configuration for gfx950:
//   * LayerNorm forward/backward with the residual add fused in,
//   * attention softmax forward/backward (scale and key-padding bias fused),
// The reference has no attention model (reference initializer.py:12-21 is an MLP); these are
// the LayerNorm / attention ops of the "BERT-base MLM" benchmark configuration. The attention
// products themselves (QK^T, PV and their gradients) run on the batched MFMA GEMM.
//
// Common structure: one 64-lane wave owns a row; a lane holds up to CH 16-byte chunks
// (8 bf16 each) of it in registers, so a row is read from memory exactly once per pass.
#include "dtf_common.h"

using namespace dtf;

__device__ __forceinline__ void t_ld8(const unsigned short* p, float (&o)[8]) {
  u16x8 a = *reinterpret_cast<const u16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(a[i]);
}
__device__ __forceinline__ void t_st8(unsigned short* p, const float (&o)[8]) {
  u16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f32_to_bf16(o[i]);
  *reinterpret_cast<u16x8*>(p) = a;
}
template <typename P>
__device__ __forceinline__ void t_ldp(const P* p, float (&o)[8]);
template <>
__device__ __forceinline__ void t_ldp<float>(const float* p, float (&o)[8]) {
  f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
  o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
template <>
__device__ __forceinline__ void t_ldp<unsigned short>(const unsigned short* p, float (&o)[8]) { t_ld8(p, o); }

// ------------------------------------------------------------------------------ LayerNorm
// y = LN(x (+ res)) * gamma + beta ; optionally writes s = x + res (the new residual stream).
// stats[row] = {mean, rstd}. H % 8 == 0, H <= 64 * 8 * CH.
// A wave takes RW consecutive rows and requests all of their chunks (non-temporal: nothing here is read twice) and the
// coefficients before it converts the first.
template <int CH, typename P, int RW>
__global__ __launch_bounds__(kBlock) void ln_fwd_kernel(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ res,
                                                        const P* __restrict__ gamma, const P* __restrict__ beta,
                                                        unsigned short* __restrict__ y,
                                                        unsigned short* __restrict__ sum_out,
                                                        float* __restrict__ stats, long long rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * RW;
  if (row0 >= rows) return;
  const int nch = H >> 3;
  u16x8 xr[RW][CH], rr[RW][CH];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch && row0 + r < rows) {
        xr[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(x + (row0 + r) * H + ch * 8));
        if (res) rr[r][c] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(res + (row0 + r) * H + ch * 8));
      }
    }
  // the coefficients are requested with the activations: after the statistics they would be one more exposed latency
  float gm[CH][8], bt[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) { t_ldp<P>(gamma + ch * 8, gm[c]); t_ldp<P>(beta + ch * 8, bt[c]); }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const long long row = row0 + r;
    if (row >= rows) break;
    float v[CH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[c][k] = bf16_to_f32(xr[r][c][k]);
        if (res) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[c][k] += bf16_to_f32(rr[r][c][k]);
        }
        if (sum_out) {
          t_st8(sum_out + row * H + ch * 8, v[c]);
          // normalise what later layers will read back: the bf16-rounded sum
#pragma unroll
          for (int k = 0; k < 8; ++k) v[c][k] = bf16_to_f32(f32_to_bf16(v[c][k]));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s += v[c][k];
      }
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (lane + 64 * c < nch) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = v[c][k] - mean; q = fmaf(d, d, q); }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)H + eps);
    if (lane == 0) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaf((v[c][k] - mean) * rstd, gm[c][k], bt[c][k]);
        t_st8(y + row * H + ch * 8, o);
      }
    }
  }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) (+ dres_in when given);
// per-block partial sums of dgamma = sum g*xhat and dbeta = sum g go to part[block][NS][H]. CS (NS = 3): a third row
// holds the column sums of dx as stored (rounded to bf16) -- the bias gradient of the linear layer whose output this
// norm normalised, which then needs no pass of its own over dx.
template <int CH, typename P, bool CS>
__global__ __launch_bounds__(kBlock) void ln_bwd_kernel(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ dy,
                                                        const unsigned short* __restrict__ dres_in,
                                                        const P* __restrict__ gamma, const float* __restrict__ stats,
                                                        unsigned short* __restrict__ dx, float* __restrict__ part,
                                                        long long rows, int H, int rows_per_block) {
  constexpr int NS = CS ? 3 : 2;
  __shared__ float red[NS][4][64 * CH * 8 / 4 + 1];  // reduced in four rounds to bound LDS (see below)
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nch = H >> 3;
  float dg[CH][8], db[CH][8], dxs[CS ? CH : 1][8];
#pragma unroll
  for (int c = 0; c < (CS ? CH : 1); ++c)
#pragma unroll
    for (int k = 0; k < 8; ++k) dxs[c][k] = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { dg[c][k] = 0.f; db[c][k] = 0.f; }
  }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = min(rows, r0 + rows_per_block);
  for (long long row = r0 + wid; row < r1; row += kBlock / 64) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float xh[CH][8], g[CH][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        t_ld8(x + row * H + ch * 8, xh[c]);
        t_ld8(dy + row * H + ch * 8, g[c]);
        // gamma is fetched per row (a cache hit) rather than held for the whole loop: 16 registers that decide
        // whether four waves or three fit a SIMD; the laundered index keeps the compiler from hoisting the load
        float gam[8];
        int go = ch * 8;
        asm volatile("" : "+v"(go));
        t_ldp<P>(gamma + go, gam);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          xh[c][k] = (xh[c][k] - mean) * rstd;
          dg[c][k] = fmaf(g[c][k], xh[c][k], dg[c][k]);
          db[c][k] += g[c][k];
          g[c][k] *= gam[k];
          s1 += g[c][k];
          s2 = fmaf(g[c][k], xh[c][k], s2);
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int ch = lane + 64 * c;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = rstd * (g[c][k] - s1 - xh[c][k] * s2);
        if (dres_in) {
          float r[8];
          t_ld8(dres_in + row * H + ch * 8, r);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] += r[k];
        }
        t_st8(dx + row * H + ch * 8, o);
        if (CS) {
#pragma unroll
          for (int k = 0; k < 8; ++k) dxs[c][k] += bf16_to_f32(f32_to_bf16(o[k]));
        }
      }
    }
  }
  // combine the 4 waves' partials: wave w adds into wave 0's registers through LDS, 2 values at a time
  float* pg = part + ((long long)blockIdx.x * NS) * H;
  float* pb = pg + H;
  float* px = pb + H;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      __syncthreads();
      red[0][wid][lane * 2] = dg[c][k]; red[0][wid][lane * 2 + 1] = dg[c][k + 1];
      red[1][wid][lane * 2] = db[c][k]; red[1][wid][lane * 2 + 1] = db[c][k + 1];
      if (CS) { red[NS - 1][wid][lane * 2] = dxs[CS ? c : 0][k]; red[NS - 1][wid][lane * 2 + 1] = dxs[CS ? c : 0][k + 1]; }
      __syncthreads();
      if (wid == 0 && lane + 64 * c < nch) {
        const int col = (lane + 64 * c) * 8 + k;
        float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, x0 = 0.f, x1 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a0 += red[0][w][lane * 2]; a1 += red[0][w][lane * 2 + 1];
          b0 += red[1][w][lane * 2]; b1 += red[1][w][lane * 2 + 1];
          if (CS) { x0 += red[NS - 1][w][lane * 2]; x1 += red[NS - 1][w][lane * 2 + 1]; }
        }
        pg[col] = a0; pg[col + 1] = a1; pb[col] = b0; pb[col + 1] = b1;
        if (CS) { px[col] = x0; px[col + 1] = x1; }
      }
    }
  }
}

// column sums of part[n_part][ns][H] (ns = 2, or 3 with the sums of dx); 32 columns per block (see partials_col_sum).
__global__ __launch_bounds__(kBlock) void ln_bwd_finalize_kernel(const float* __restrict__ part, int n_part, int H,
                                                                 int ns, float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta, float* __restrict__ dxsum) {
  float s;
  if (!partials_col_sum(part, n_part, (long long)ns * H, ns * H, s)) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  if (c < H) dgamma[c] = s;
  else if (c < 2 * H) dbeta[c - H] = s;
  else dxsum[c - 2 * H] = s;
}

// ------------------------------------------------------------------------------ softmax
// In-place capable. Row r of the [rows, L] matrix belongs to batch b = r / rows_per_batch.
// p = softmax(scale * s + bias[b, :]) ;  bias may be null (fp32, 0 / -inf style additive mask).
template <int CH>
__global__ __launch_bounds__(kBlock) void softmax_fwd_kernel(const unsigned short* __restrict__ s,
                                                             unsigned short* __restrict__ p,
                                                             const float* __restrict__ bias, long long rows, int L,
                                                             long long rows_per_batch, float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = L >> 3;
  const float* brow = bias ? bias + (row / rows_per_batch) * L : nullptr;
  float v[CH][8];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      t_ld8(s + row * L + ch * 8, v[c]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[c][k] *= scale;
        if (brow) v[c][k] += brow[ch * 8 + k];
        m = fmaxf(m, v[c][k]);
      }
    }
  }
  m = wave_max(m);
  if (m == -INFINITY) m = 0.f;  // fully masked row -> zeros, not NaN
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    if (lane + 64 * c < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[c][k] = __expf(v[c][k] - m); sum += v[c][k]; }
    }
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[c][k] *= inv;
      t_st8(p + row * L + ch * 8, v[c]);
    }
  }
}

// ds = scale * p * (dp - sum(p * dp)); in-place capable on dp.
template <int CH>
__global__ __launch_bounds__(kBlock) void softmax_bwd_kernel(const unsigned short* __restrict__ p,
                                                             const unsigned short* __restrict__ dp,
                                                             unsigned short* __restrict__ ds, long long rows, int L,
                                                             float scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = L >> 3;
  float pv[CH][8], g[CH][8];
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
      t_ld8(p + row * L + ch * 8, pv[c]);
      t_ld8(dp + row * L + ch * 8, g[c]);
#pragma unroll
      for (int k = 0; k < 8; ++k) dot = fmaf(pv[c][k], g[c][k], dot);
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int ch = lane + 64 * c;
    if (ch < nch) {
#pragma unroll
      for (int k = 0; k < 8; ++k) g[c][k] = scale * pv[c][k] * (g[c][k] - dot);
      t_st8(ds + row * L + ch * 8, g[c]);
    }
  }
}

// ------------------------------------------------------------------------------ C ABI
static inline int ch_for(int n) { return n <= 512 ? 1 : n <= 1024 ? 2 : n <= 2048 ? 4 : n <= 4096 ? 8 : 0; }

static int g_ln_fwd_rows = 0;   // rows per wave of the LayerNorm forward kernel (0 = by row width; 1, 2, 4: measurement hook)
DTF_API void dtf_ln_set_fwd_rows(int n) { if (n == 0 || n == 1 || n == 2 || n == 4) g_ln_fwd_rows = n; }

DTF_API int dtf_ln_fwd(const void* x, const void* res, const void* gamma, const void* beta, void* y, void* sum_out,
                       float* stats, long long rows, int H, float eps, int param_dtype, hipStream_t stream) {
  if (rows <= 0) return 0;
  const int ch = ch_for(H);
  if (H % 8 || ch == 0) return -1;
  // rows per wave: rows of at most 4 chunks per lane are taken in pairs (768 wide, 196,608 rows: 212 us against 219 with
  // one and 220 with four: the kernel stays near 4.2 TB/s whatever the shape of its requests); g_ln_fwd_rows overrides
  const int rw = g_ln_fwd_rows ? g_ln_fwd_rows : (ch <= 4 ? 2 : 1);
  const long long per_block = 4LL * rw;
  const dim3 grid((unsigned)((rows + per_block - 1) / per_block));
#define LN(CHN, P, RWN)                                                                                      \
  hipLaunchKernelGGL((ln_fwd_kernel<CHN, P, RWN>), grid, dim3(kBlock), 0, stream, (const unsigned short*)x, \
                     (const unsigned short*)res, (const P*)gamma, (const P*)beta, (unsigned short*)y,       \
                     (unsigned short*)sum_out, stats, rows, H, eps)
#define LN_R(CHN, P) do { if (rw == 4 && CHN <= 2) LN(CHN, P, 4); else if (rw >= 2 && CHN <= 4) LN(CHN, P, 2); else LN(CHN, P, 1); } while (0)
#define LN_D(CHN) do { if (param_dtype == 0) LN_R(CHN, float); else LN_R(CHN, unsigned short); } while (0)
  switch (ch) { case 1: LN_D(1); break; case 2: LN_D(2); break; case 4: LN_D(4); break; default: LN_D(8); }
#undef LN_R
#undef LN_D
#undef LN
  return (int)hipGetLastError();
}

static int g_ln_bwd_blocks = 1024;    // cap on the blocks (= partial rows) of the LayerNorm backward pass: four blocks per CU (BERT-base step 57.15 -> 56.90 ms against 512)
DTF_API void dtf_ln_set_bwd_blocks(int n) { if (n >= 1) g_ln_bwd_blocks = n; }

DTF_API int dtf_ln_bwd_num_partials(long long rows) {
  long long nb = (rows + 63) / 64;
  if (nb > g_ln_bwd_blocks) nb = g_ln_bwd_blocks;
  return (int)(nb < 1 ? 1 : nb);
}

// part: fp32 [n_partials * (dxsum ? 3 : 2) * H]. dxsum (optional, fp32 [H]): column sums of dx.
DTF_API int dtf_ln_bwd(const void* x, const void* dy, const void* dres_in, const void* gamma, const float* stats,
                       void* dx, float* dgamma, float* dbeta, float* part, long long rows, int H, int param_dtype,
                       float* dxsum, hipStream_t stream) {
  if (rows <= 0) return 0;
  const int ch = ch_for(H);
  if (H % 8 || ch == 0 || ch > 4) return -1;
  const int nb = dtf_ln_bwd_num_partials(rows);
  const int rpb = (int)((rows + nb - 1) / nb);
#define LNB(CHN, P, CS)                                                                                          \
  hipLaunchKernelGGL((ln_bwd_kernel<CHN, P, CS>), dim3(nb), dim3(kBlock), 0, stream, (const unsigned short*)x,  \
                     (const unsigned short*)dy, (const unsigned short*)dres_in, (const P*)gamma, stats,         \
                     (unsigned short*)dx, part, rows, H, rpb)
#define LNB_P(CHN, CS) do { if (param_dtype == 0) LNB(CHN, float, CS); else LNB(CHN, unsigned short, CS); } while (0)
#define LNB_D(CHN) do { if (dxsum) LNB_P(CHN, true); else LNB_P(CHN, false); } while (0)
  switch (ch) { case 1: LNB_D(1); break; case 2: LNB_D(2); break; default: LNB_D(4); }
#undef LNB_D
#undef LNB_P
#undef LNB
  const int ns = dxsum ? 3 : 2;
  hipLaunchKernelGGL(ln_bwd_finalize_kernel, dim3((ns * H + 31) / 32), dim3(kBlock), 0, stream, part, nb, H, ns, dgamma,
                     dbeta, dxsum);
  return (int)hipGetLastError();
}

DTF_API int dtf_softmax_fwd(const void* s, void* p, const float* bias, long long rows, int L, long long rows_per_batch,
                            float scale, hipStream_t stream) {
  if (rows <= 0) return 0;
  const int ch = ch_for(L);
  if (L % 8 || ch == 0) return -1;
  const dim3 grid((unsigned)((rows + 3) / 4));
#define SM(CHN)                                                                                               \
  hipLaunchKernelGGL((softmax_fwd_kernel<CHN>), grid, dim3(kBlock), 0, stream, (const unsigned short*)s,     \
                     (unsigned short*)p, bias, rows, L, rows_per_batch, scale)
  switch (ch) { case 1: SM(1); break; case 2: SM(2); break; case 4: SM(4); break; default: SM(8); }
#undef SM
  return (int)hipGetLastError();
}

DTF_API int dtf_softmax_bwd(const void* p, const void* dp, void* ds, long long rows, int L, float scale,
                            hipStream_t stream) {
  if (rows <= 0) return 0;
  const int ch = ch_for(L);
  if (L % 8 || ch == 0) return -1;
  const dim3 grid((unsigned)((rows + 3) / 4));
#define SMB(CHN)                                                                                              \
  hipLaunchKernelGGL((softmax_bwd_kernel<CHN>), grid, dim3(kBlock), 0, stream, (const unsigned short*)p,     \
                     (const unsigned short*)dp, (unsigned short*)ds, rows, L, scale)
  switch (ch) { case 1: SMB(1); break; case 2: SMB(2); break; case 4: SMB(4); break; default: SMB(8); }
#undef SMB
  return (int)hipGetLastError();
}
