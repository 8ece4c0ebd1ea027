// Split-precision linear layers:  C[M, N] = act(A[M, K] . W[N, K]^T + bias[N]) + residual[M, N]
//
// The f32 matrix pipe (v_mfma_f32_32x32x2_f32, 157 TFLOP/s) bounds gemm.hip; the f16 pipe is 16x faster.  Here both
// operands are held as PAIRS of f16 images with one power-of-two scale per row,
//     x = 2^-e * (hi + lo),   hi = f16(x 2^e),   lo = f16(x 2^e - hi),   max_k |x_k| 2^e in [2^14, 2^15)
// (22 significant bits per element; x 2^e - hi is exact in f32), and a product is THREE f16 MFMAs accumulated in f32:
//     a.w  ~=  a_lo w_hi + a_hi w_lo + a_hi w_hi          (the a_lo w_lo term, < 2^-22 relative, is dropped)
// Products of two f16 are exact in f32, so the only errors are the two operand representations (2^-22 each), the
// dropped term and the f32 accumulation the exact kernel has as well: ~3 * 2^-22 per product against 2^-24 for the
// f32 chain.  Used for the T5 / BERT / adaptor linear layers (weights are static: split once at load); the dense arm,
// the fine stage and the RQ kernels keep their exact f32 chains.  Same role in the reference as gemm.hip
// (MEVI/transformers/modeling_t5.py:181-186, 217-220, 350-358, 412; modeling_bert.py linear layers).
//
// Kernel = a persistent LDS-DMA tile stream (mfma_split_stream.h, re-cut from the dense pre-filter's mfma_pp_f16.h): 256 x 256
// output tiles, 8 waves, one v_mfma_f32_32x32x16_f16 per 16 k.  An image row is [hi (Kp halves) | lo (Kp halves)], Kp = K
// rounded up to 32 (at least 64); per 16 k the three products a_lo w_hi, a_hi w_hi, a_hi w_lo are accumulated in that
// order, each operand slab fetched into LDS once per 32 k.  Every output depends on its own A row, its own W row and this
// fixed order only, so a row has the same bits whatever batch it travels in; gemm_split_skinny_kernel (few outputs: the
// latency path) issues the identical MFMA sequence from global memory and therefore returns the same bits.
#include <cstdlib>

#include "mfma_split_stream16.h"

namespace mevi {
namespace {

__device__ __forceinline__ int pow2_exp(float m) {
  // e with m * 2^e in [2^14, 2^15); clamped so that 2^e and 2^-e stay finite normal floats
  if (!(m > 0.f) || isinf(m)) return 0;
  int e;
  (void)frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
  int s = 15 - e;
  return s > 100 ? 100 : (s < -100 ? -100 : s);
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const float4 v, int e, h4 &hi, h4 &lo) {
  const float x[4] = {ldexpf(v.x, e), ldexpf(v.y, e), ldexpf(v.z, e), ldexpf(v.w, e)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = (_Float16)x[i];
    lo[i] = (_Float16)(x[i] - (float)hi[i]);
  }
}

// rows of x f32[m, k] (row stride ldx) -> image [m, 2 * kp] halves + exponent per row.  One wave per row.
__global__ __launch_bounds__(256) void split_rows_kernel(const float *__restrict__ x, long long ldx, long long m, int k,
                                                        int kp, _Float16 *__restrict__ img, signed char *__restrict__ exps,
                                                        float *__restrict__ norms) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= m) return;
  const int lane = threadIdx.x & 63;
  const float *xr = x + (size_t)r * ldx;
  if (k == 768 && kp == 768) {
    // the t5-base / bert-base width: the row in registers, its three loads issued together (the loops below make three dependent
    // trips, then read the row again); same arithmetic in the same order: same bits
    float4 v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const float4 *>(xr + 4 * lane + 256 * j);
    float mx = 0.f, ss = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
      ss = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, ss))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off));
      ss += __shfl_xor(ss, off);
    }
    if (lane == 0 && norms) norms[r] = sqrtf(ss) * 1.0001f;
    const int e = pow2_exp(mx);
    _Float16 *o = img + (size_t)r * 2 * kp;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h4 hi, lo;
      split4(v[j], e, hi, lo);
      *reinterpret_cast<h4 *>(o + 4 * lane + 256 * j) = hi;
      *reinterpret_cast<h4 *>(o + kp + 4 * lane + 256 * j) = lo;
    }
    if (lane == 0) exps[r] = (signed char)e;
    return;
  }
  float mx = 0.f, ss = 0.f;
  for (int c = lane * 4; c < k; c += 256) {
    const float4 v = *reinterpret_cast<const float4 *>(xr + c);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    ss = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, ss))));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, off));
    ss += __shfl_xor(ss, off);
  }
  if (lane == 0 && norms) norms[r] = sqrtf(ss) * 1.0001f;  // rounded up: it bounds the next layer's outputs
  const int e = pow2_exp(mx);
  _Float16 *o = img + (size_t)r * 2 * kp;
  for (int c = lane * 4; c < kp; c += 256) {
    h4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
    if (c < k) split4(*reinterpret_cast<const float4 *>(xr + c), e, hi, lo);
    *reinterpret_cast<h4 *>(o + c) = hi;
    *reinterpret_cast<h4 *>(o + kp + c) = lo;
  }
  if (lane == 0) exps[r] = (signed char)e;
}

// T5LayerNorm (t5_ops.hip: rmsnorm_kernel, same arithmetic per element) written straight into the split image:
// y = w * (x / sqrt(mean(x^2) + eps)).  One wave per row; y is recomputed for the second pass (the row sits in L1).
__global__ __launch_bounds__(256) void rmsnorm_split_kernel(const float *__restrict__ x, long long ldx,
                                                           const float *__restrict__ w, float eps, long long rows, int dim,
                                                           int kp, _Float16 *__restrict__ img, signed char *__restrict__ exps,
                                                           float *__restrict__ norms) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 *xr = reinterpret_cast<const float4 *>(x + r * ldx);
  const float4 *wv = reinterpret_cast<const float4 *>(w);
  if (dim == 768 && kp == 768) {
    // the t5-base / bert-base width: the row stays in registers (three float4 per lane) -- one trip to memory instead of
    // three dependent ones (the latency path runs this kernel on a handful of rows: 5.1 us per call, most of it those trips).
    // Same arithmetic in the same order as the general form below: same bits.
    float4 v[3], g[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) v[j] = xr[lane + 64 * j], g[j] = wv[lane + 64 * j];
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ss = fmaf(v[j].x, v[j].x, ss); ss = fmaf(v[j].y, v[j].y, ss); ss = fmaf(v[j].z, v[j].z, ss); ss = fmaf(v[j].w, v[j].w, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float denom = sqrtf(ss / (float)dim + eps);
    float mx = 0.f, s2 = 0.f;
    float4 y[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      y[j] = make_float4(g[j].x * (v[j].x / denom), g[j].y * (v[j].y / denom), g[j].z * (v[j].z / denom), g[j].w * (v[j].w / denom));
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(y[j].x), fabsf(y[j].y))), fmaxf(fabsf(y[j].z), fabsf(y[j].w)));
      s2 = fmaf(y[j].x, y[j].x, fmaf(y[j].y, y[j].y, fmaf(y[j].z, y[j].z, fmaf(y[j].w, y[j].w, s2))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off));
      s2 += __shfl_xor(s2, off);
    }
    const int e = pow2_exp(mx);
    _Float16 *o = img + (size_t)r * 2 * kp;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h4 hi, lo;
      split4(y[j], e, hi, lo);
      *reinterpret_cast<h4 *>(o + 4 * (lane + 64 * j)) = hi;
      *reinterpret_cast<h4 *>(o + kp + 4 * (lane + 64 * j)) = lo;
    }
    if (lane == 0) {
      exps[r] = (signed char)e;
      norms[r] = sqrtf(s2) * 1.0001f;
    }
    return;
  }
  float ss = 0.f;
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 v = xr[i];
    ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
  const float denom = sqrtf(ss / (float)dim + eps);
  float mx = 0.f, s2 = 0.f;
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 v = xr[i], g = wv[i];
    const float4 y = make_float4(g.x * (v.x / denom), g.y * (v.y / denom), g.z * (v.z / denom), g.w * (v.w / denom));
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(y.x), fabsf(y.y))), fmaxf(fabsf(y.z), fabsf(y.w)));
    s2 = fmaf(y.x, y.x, fmaf(y.y, y.y, fmaf(y.z, y.z, fmaf(y.w, y.w, s2))));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, off));
    s2 += __shfl_xor(s2, off);
  }
  const int e = pow2_exp(mx);
  _Float16 *o = img + (size_t)r * 2 * kp;
  for (int i = lane; i < kp / 4; i += 64) {
    h4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
    if (i < dim / 4) {
      const float4 v = xr[i], g = wv[i];
      split4(make_float4(g.x * (v.x / denom), g.y * (v.y / denom), g.z * (v.z / denom), g.w * (v.w / denom)), e, hi, lo);
    }
    *reinterpret_cast<h4 *>(o + 4 * i) = hi;
    *reinterpret_cast<h4 *>(o + kp + 4 * i) = lo;
  }
  if (lane == 0) {
    exps[r] = (signed char)e;
    norms[r] = sqrtf(s2) * 1.0001f;
  }
}

// Where an output goes: f32 C (row stride ldc), or -- for a layer whose only consumer is the next GEMM (the FFN's
// relu(x Wi^T)) -- straight into a split image.  The image needs the row's exponent BEFORE the row is complete (its
// columns are spread over workgroups), so it comes from a bound instead of the row maximum:
//   |act(a.w + b)| <= ||a|| * max_n ||w_n|| + max|b|     (Cauchy-Schwarz; relu / gelu do not increase magnitudes)
// with ||a|| carried by the producer of A (anorm) and the weight-side constants in `obound`.  The bound sits a few
// binades above the row's real maximum, which costs range, not precision: hi keeps 11 bits down to 2^-29 of the
// bound, the pair 22 bits down to 2^-18 of it, and below that the absolute error is < 2^-40 of the bound.
struct SplitOut {
  _Float16 *img;       // [M, 2 * np] or nullptr
  signed char *exps;   // [M]
  float *norms;        // [M] or nullptr: bound on the output row's l2 norm (for a further split-out layer)
  const float *anorm;  // [M] l2 norms of the A rows
  int np;
  float wnorm_max, babs_max, onorm_scale;
  float alpha;         // OUT = 3 (the fused PAWA head): scale of the hidden states
  // ---- round 5: T5LayerNorm folded into the linear layers around it (modeling_t5.py:155-171 feeds exactly one projection) ----
  // rmsnorm(x) W^T = rsqrt(mean x^2 + eps) * (x (W (.) w_ln)^T): the CONSUMER multiplies the raw residual row (its split image) with
  // the weight that carries w_ln and scales the product per row; the PRODUCER of the residual stream (the GEMM that adds its result
  // to it) writes, next to the f32 rows, their split image and the sums of squares of their 16-column blocks.
  const float *rscale;  // consumer: [M] row scale, applied to the product before bias / activation (tile stream; nullptr: none)
  const float *rparts;  // consumer, latency kernels: [M][nparts] block sums of squares -> the scale is computed in the kernel
  int nparts;           //   (same formula, same order as mevi_row_rscale_f32)
  float rs_dim, rs_eps;
  float anorm_const;    // split output without per-row norms (anorm == nullptr): ||A row|| bound, e.g. sqrt(d) of a normed row
  float *ssq;           // producer: [M][N / 16] sum of squares of the OUTPUT row's 16-column blocks (N % 16 == 0)
  const float *xbound;  // producer with image: [M] bound on max |residual row|; the image's exponent = that of xbound + ||a|| wnorm_max
  float *obound;        // producer with image: [M] the new bound
};

// sum of squares of a row's 16-column block: the lane's four columns as an fma chain, then (kq0 + kq1) + (kq2 + kq3) -- the ONE
// order every producer (tile stream, latency kernels) and mevi_split_rows_ssq use, so a row's scale has the same bits in any batch
__device__ __forceinline__ float block_ssq(float v0, float v1, float v2, float v3) {
  float q = fmaf(v3, v3, fmaf(v2, v2, fmaf(v1, v1, v0 * v0)));
  q += __shfl_xor(q, 16);
  q += __shfl_xor(q, 32);
  return q;
}
// the row scale from its block sums: blocks added in index order
__device__ __forceinline__ float row_rscale(const float *__restrict__ parts, int nparts, float dim, float eps) {
  float ss = 0.f;
  if (nparts & 3) {       // narrow streams (the miniature models of the tests): rows of the table are not 16-byte aligned
    for (int i = 0; i < nparts; ++i) ss += parts[i];
  } else {
    for (int i = 0; i < nparts; i += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(parts + i);
      ss += v.x, ss += v.y, ss += v.z, ss += v.w;
    }
  }
  return 1.0f / sqrtf(ss / dim + eps);
}

__device__ __forceinline__ int out_exp(const SplitOut &so, int m) {
  return pow2_exp(fmaf(so.anorm[m], so.wnorm_max, so.babs_max) * 1.001f);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int OOB = (int)0x80000000;  // voffset beyond every descriptor below: the load returns 0, the store is dropped

__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7ffffff0, 0x00020000);
}

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
  if constexpr (ACT == 1) return fmaxf(v, 0.f);
  if constexpr (ACT == 2) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));  // erf GELU (BERT 'gelu')
  return v;
}

// Roles inside the tile stream: the MFMA's first operand (the stream's 2 x 128-row "A" tile) is W, the second (its
// 256-row "B" tile) the activations, so a lane ends up with ONE output row m = m0 + 128 wn + 32 ni + (lane & 31) per
// ni and, per accumulator register quad, FOUR CONSECUTIVE columns n = n0 + 128 grp + 64 wm + 32 mi + 8 q + 4 half + j:
// the epilogue moves 16 bytes per lane and instruction (32 stores per wave and tile instead of 128), the row exponent
// / norm are per-lane scalars.  All of its memory operations are buffer operations on a per-tile descriptor with a
// 32-bit offset (one VGPR of addressing; masked lanes get an out-of-range offset instead of a branch: a load under a
// per-element condition makes the compiler wait vmcnt(0) for each one).
template <int ACT, int OUT>   // OUT: 0 = f32 C, 1 = f32 C + residual, 2 = split image
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_split_kernel(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int n_mtiles, int n_ntiles, SplitOut so) {
  constexpr bool SPLIT_OUT = OUT == 2, HAS_RES = OUT == 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;  // as xcd_remap
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int lrow = lane & 31, half = lane >> 5;
  const int row_bytes = kp * 4;
  int head_m = 0, head_n = 0, tail_m = 0, tail_n = 0, n_pend = 0;  // tiles fetched ahead of their epilogue (<= 2)

  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    if (n_pend == 0) head_m = mt, head_n = nt;
    else tail_m = mt, tail_n = nt;
    ++n_pend;
    long long rows_left;
    if (w8 < 4) {  // waves 0-3 stage the 256 W rows (LDS rows [0, 256)), waves 4-7 the 256 activation rows
      s.src = reinterpret_cast<const char *>(W) + (size_t)nt * 256 * (size_t)row_bytes;
      rows_left = (long long)N - (long long)nt * 256;
    } else {
      s.src = reinterpret_cast<const char *>(A) + (size_t)mt * 256 * (size_t)row_bytes;
      rows_left = (long long)M - (long long)mt * 256;
    }
    if (rows_left > 256) rows_left = 256;
    s.bytes = (unsigned int)(rows_left * row_bytes);
    return true;
  };
  auto begin = [&]() {};
  auto emit = [&](f32x16 (&acc)[2][4]) {
    const int mt = head_m, nt = head_n;
    head_m = tail_m, head_n = tail_n;
    --n_pend;
    const int m0 = mt * 256 + 128 * wn + lrow;           // + 32 ni
    // the lane's columns: nt * 256 + 128 * grp + 64 * wm + 4 * half + 32 mi + 8 q (+ j)
    const bool interior = (mt + 1) * 256 <= M && (nt + 1) * 256 <= N;
    const int ldc4 = (int)ldc * 4, ldr4 = (int)ldr * 4;
    // per-tile descriptors (64-bit origin in SGPRs) + 32-bit lane offsets
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(SPLIT_OUT ? nullptr : C + (size_t)mt * 256 * ldc + (size_t)nt * 256);
    const __amdgpu_buffer_rsrc_t rr =
        tile_rsrc(residual ? residual + (size_t)mt * 256 * ldr + (size_t)nt * 256 : nullptr);
    const __amdgpu_buffer_rsrc_t ri =
        tile_rsrc(SPLIT_OUT ? so.img + (size_t)mt * 256 * 2 * so.np + (size_t)nt * 256 : nullptr);
    const __amdgpu_buffer_rsrc_t rw = tile_rsrc(ew + nt * 256);
    const __amdgpu_buffer_rsrc_t rb = tile_rsrc(bias ? bias + nt * 256 : nullptr);
    const int ncol = 128 * grp + 64 * wm + 4 * half;     // column of the lane's first quad inside the tile
    const int nvalid = N - nt * 256;                      // columns of this tile that exist (edge tiles)
    int em[4], eo[SPLIT_OUT ? 4 : 1];
    bool mok[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int m = m0 + 32 * ni;
      mok[ni] = interior || m < M;
      em[ni] = ea[min(m, M - 1)];
      if constexpr (SPLIT_OUT) {
        const float bound = fmaf(so.anorm[min(m, M - 1)], so.wnorm_max, so.babs_max);
        eo[ni] = pow2_exp(bound * 1.001f);
        if (nt == 0 && grp == 0 && wm == 0 && half == 0 && mok[ni]) {  // one writer per row
          so.exps[m] = (signed char)eo[ni];
          if (so.norms) so.norms[m] = bound * so.onorm_scale;
        }
      }
    }
    // Every load is issued BEFORE the stores it does not depend on: vmcnt retires in issue order, so a load placed
    // after a store waits for that store's write acknowledgement (microseconds under load).  The column exponents /
    // bias of all eight quads are fetched up front (the fragment registers are dead here), the residual rows of
    // quad i + 1 before the stores of quad i.
    unsigned int wq[8];   // four int8 exponents per quad
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = ncol + 32 * (i >> 2) + 8 * (i & 3);
      wq[i] = __builtin_amdgcn_raw_buffer_load_b32(rw, (interior || c < nvalid) ? c : OOB, 0, 0);
    }
    u32x4 bq[2], res[2][HAS_RES ? 4 : 1];
    auto load_res = [&](int i, u32x4 &b, u32x4 (&r)[HAS_RES ? 4 : 1]) {
      const int c = ncol + 32 * (i >> 2) + 8 * (i & 3);
      const bool cok = interior || c < nvalid;
      b = u32x4{0u, 0u, 0u, 0u};
      if (bias) b = __builtin_amdgcn_raw_buffer_load_b128(rb, cok ? c * 4 : OOB, 0, 0);
      if constexpr (HAS_RES) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int rowt = 128 * wn + 32 * ni + lrow;
          r[ni] = __builtin_amdgcn_raw_buffer_load_b128(rr, (cok && mok[ni]) ? rowt * ldr4 + c * 4 : OOB, 0, 0);
        }
      }
    };
    load_res(0, bq[0], res[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mi = i >> 2, q = i & 3;
      const int c = ncol + 32 * mi + 8 * q;
      const bool cok = interior || c < nvalid;
      if (i + 1 < 8) load_res(i + 1, bq[(i + 1) & 1], res[(i + 1) & 1]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int rowt = 128 * wn + 32 * ni + lrow;      // row inside the tile
        const bool ok = cok && mok[ni];
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = ldexpf(acc[mi][ni][4 * q + j], -(em[ni] + ((int)(wq[i] << (24 - 8 * j)) >> 24))) + __uint_as_float(bq[i & 1][j]);
          x = act_fn<ACT>(x);
          if constexpr (HAS_RES) x += __uint_as_float(res[i & 1][ni][j]);
          v[j] = x;
        }
        if constexpr (SPLIT_OUT) {
          h4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xs = ldexpf(v[j], eo[ni]);
            hi[j] = (_Float16)xs;
            lo[j] = (_Float16)(xs - (float)hi[j]);
          }
          const int off = ok ? rowt * (so.np * 4) + c * 2 : OOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ri, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ri, off, so.np * 2, 0);
        } else {
          const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, ok ? rowt * ldc4 + c * 4 : OOB, 0, 0);
        }
      }
    }
  };
  split_tile_stream(row_bytes, kp * 2, kp / 32, lds, next, begin, emit);
}

// Few outputs (the latency path): one WORKGROUP per 32 x 32 outputs, the same MFMA sequence as the tile stream (W first,
// activations second; per 16 k: a_lo w_hi, a_hi w_hi, a_hi w_lo), so a row has the same bits here and there.
// What bounds this shape is the latency of one dependent chain per output tile, so the design is: as many CUs as there are
// tiles (a t5-base projection of one query: 48 .. 192), and per CU as many bytes in flight as LDS holds.
//   * fragments straight from global memory touch 32 rows x 32 B per instruction (32 cache lines 2 Kp bytes apart):
//     tools/probes/skinny_probe.hip -- 25 of 27 us at K = 768 were those loads, 6 us the MFMA chain;
//   * here the four waves stream the tile's operands with LDS-DMA (buffer_load_dwordx4 ... lds; eight lanes per 128-B row
//     piece, source-side swizzle) into a ring of SK_RING chunks of 64 k (16 KiB: a_hi, a_lo, w_hi, w_lo x 32 rows), seven
//     chunks = 112 KiB in flight; wave w stages slab w, wave 0 alone reads fragments and multiplies.
constexpr int SK_RING = 8;
constexpr int SK_SLAB = 32 * 32;                  // floats: 32 rows x 128 B
constexpr int SK_CHUNK = 4 * SK_SLAB;
__global__ __launch_bounds__(256) void gemm_split_skinny_kernel(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, SplitOut so) {
  __shared__ __attribute__((aligned(16))) float sm[SK_RING * SK_CHUNK];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  // staging: wave 0 a_hi, 1 a_lo, 2 w_hi, 3 w_lo; piece i of a slab = rows 8 i .. 8 i + 7, lane -> row 8 i + (lane >> 3),
  // LDS slot lane & 7 of that row, which holds the row's 16-byte piece  slot ^ key(row),  key(r) = (r >> 1) & 7
  const bool is_w = wave >= 2;
  const int rows_left = (is_w ? N - n0 : M - m0) - 1;
  const __amdgpu_buffer_rsrc_t rsrc =
      tile_rsrc((is_w ? W + (size_t)n0 * 2 * kp : A + (size_t)m0 * 2 * kp) + ((wave & 1) ? kp : 0));
  int voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * i + (lane >> 3);
    voff[i] = min(r, rows_left) * 4 * kp + 16 * ((lane & 7) ^ ((r >> 1) & 7));
  }
  const int nchunk = (kp + 63) / 64;
  // kp is a multiple of 32 and >= 64: the last chunk may start 32 k early (clamped) -- its first two k-steps were then
  // already multiplied and are skipped.  Chunks past the end are issued out of bounds (no traffic, zeros into a free
  // ring slot): every iteration issues four pieces, so one vmcnt constant tells when a chunk has landed.
  auto issue = [&](int c) {
    float *dst = sm + (c & (SK_RING - 1)) * SK_CHUNK + wave * SK_SLAB;
    const int soff = 2 * min(64 * c, kp - 64);
    const bool past = c >= nchunk;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + 256 * i), 16,
                                               past ? OOB : voff[i], soff, 0, 0);
  };
  // the epilogue's operands, fetched ahead of the stream (they would otherwise be a dependent round trip at the end)
  const int m = m0 + lrow;
  const bool mok = m < M;
  const int mc = min(m, M - 1);
  const int em = ea[mc];
  int eo = 0;
  if (so.img) {
    eo = out_exp(so, mc);
    if (wave == 0 && mok && n0 == 0 && half == 0) {
      so.exps[m] = (signed char)eo;
      if (so.norms) so.norms[m] = fmaf(so.anorm[m], so.wnorm_max, so.babs_max) * so.onorm_scale;
    }
  }
  int pw[4];
  f32x4 pb[4], pr[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = min(n0 + 8 * q + 4 * half, N - 4);
    pw[q] = *reinterpret_cast<const int *>(ew + n);
    pb[q] = bias ? *reinterpret_cast<const f32x4 *>(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    pr[q] = residual ? *reinterpret_cast<const f32x4 *>(residual + (size_t)mc * ldr + n) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int foff[4];
#pragma unroll
  for (int s_ = 0; s_ < 4; ++s_) foff[s_] = lrow * 32 + 4 * ((2 * s_ + half) ^ ((lrow >> 1) & 7));
#pragma unroll
  for (int c = 0; c < SK_RING - 1; ++c) issue(c);
  for (int c = 0; c < nchunk; ++c) {
    // chunk c landed (this wave's pieces: all but the six chunks issued after it), then everybody's; wave 0 is done with c - 1
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(c + SK_RING - 1);
    if (wave == 0) {
      const int k0 = 64 * c;
      const int skip = k0 > kp - 64 ? (k0 - (kp - 64)) / 16 : 0;
      const float *b = sm + (c & (SK_RING - 1)) * SK_CHUNK;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) {
        if (s_ < skip) continue;
        const f16x8 ah = *reinterpret_cast<const f16x8 *>(b + 0 * SK_SLAB + foff[s_]);
        const f16x8 al = *reinterpret_cast<const f16x8 *>(b + 1 * SK_SLAB + foff[s_]);
        const f16x8 wh = *reinterpret_cast<const f16x8 *>(b + 2 * SK_SLAB + foff[s_]);
        const f16x8 wl = *reinterpret_cast<const f16x8 *>(b + 3 * SK_SLAB + foff[s_]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the empty tail pieces: nothing may target LDS past the loop
  if (wave != 0 || !mok) return;
  const int a8 = act;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + 8 * q + 4 * half;       // n, N, ldc multiples of 4: a quad is whole or absent, and 16-byte aligned
    if (n < N) {
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float t = ldexpf(acc[4 * q + j], -(em + (int)(signed char)(pw[q] >> (8 * j))));
        if (bias) t += pb[q][j];
        t = a8 == 1 ? act_fn<1>(t) : (a8 == 2 ? act_fn<2>(t) : t);
        if (residual) t += pr[q][j];
        v[j] = t;
      }
      if (so.img) {
        f16x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xs = ldexpf(v[j], eo);
          hi[j] = (_Float16)xs;
          lo[j] = (_Float16)(xs - (float)hi[j]);
        }
        _Float16 *o = so.img + (size_t)m * 2 * so.np + n;
        *reinterpret_cast<f16x4 *>(o) = hi;
        *reinterpret_cast<f16x4 *>(o + so.np) = lo;
      } else {
        *reinterpret_cast<f32x4 *>(C + (size_t)m * ldc + n) = v;
      }
    }
  }
}

// ---- the same two kernels on v_mfma_f32_16x16x32_f16 (mfma_split_stream16.h; the default) -------------------------------------
// Roles as above: the MFMA's first operand is W, the second the activations.  A lane ends up, per 16 x 16 block (mi, ni), with
// ONE output row m = m0 + 128 wn + 16 ni + (lane & 15) and FOUR CONSECUTIVE columns n = n0 + 128 grp + 64 wm + 16 mi + 4 (lane >> 4)
// + j: 16 bytes per lane and block, 64 contiguous bytes per row and store instruction.  The epilogue walks the 32 blocks in
// eight steps of four (mi, half of the rows), the residual rows of step i + 1 loaded before the stores of step i.
// OUT = 3, the PAWA head fused (modeling_t5.py:1683-1684: lm_logits = sequence_output . (adaptor_weight + lm_head_weight)): the W
// rows are (column c, d) pairs, 768 = three tiles per column; instead of storing its 256 x 256 block of head-matrix elements
// x = acc + bias the epilogue multiplies it with the rows' hidden states (`residual` = s f32 [M, 768], row stride ldr) and
// writes, per wave, row and tile, ONE partial sum -- C = part f32 [n_ntiles][M][4] (4 = the waves (grp, wm) of a row) -- in
// the order t5_ops.hip::adaptive_logits_rows768_kernel reproduces for table rows; mevi_logits_finish_f32 adds the twelve
// partials of a (row, column).  The head matrices of 70 k beams (55 GB at K = 256) are never written.
// NI: 16-row activation blocks per wave -- tiles of TM = 32 NI activation rows x 256 W rows (8: the 256 x 256 tile above; 4 / 2:
// 128 / 64 activation rows for GEMMs whose 256-row tiles would not fill the device, split_tile_stream16<NI>; same bits per row).
// OUT = 4 (round 5): f32 C + residual AND the split image of the result (exponent from so.xbound + ||a|| wnorm_max) AND the sums of
// squares of its 16-column blocks (so.ssq) -- the producer side of the folded T5LayerNorm (SplitOut).
template <int ACT, int OUT, int NI = 8>   // OUT: 0 = f32 C, 1 = f32 C + residual, 2 = split image, 3 = fused head (above), 4 = f32 + residual + image + block sums
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_split16_kernel(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int n_mtiles, int n_ntiles, SplitOut so) {
  constexpr bool SPLIT_OUT = OUT == 2, LOGITS = OUT == 3, BOTH = OUT == 4, HAS_RES = OUT == 1 || OUT == 3 || OUT == 4;
  constexpr int TM = 32 * NI, PR = NI / 2, STEPS = 4 * PR;   // tile height; row-block pairs per block column; epilogue steps
  static_assert(!LOGITS || NI == 8, "the fused head runs on 256-row tiles");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int nwg = n_mtiles * n_ntiles;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int range_base = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;  // as xcd_remap
  const int range_len = q8 + (xcd < r8 ? 1 : 0);
  int item = blockIdx.x >> 3;
  const int t = threadIdx.x;
  const int lane = t & 63;
  const int w8 = __builtin_amdgcn_readfirstlane(t >> 6);
  const int grp = w8 >> 2, wm = (w8 >> 1) & 1, wn = w8 & 1;
  const int row_bytes = kp * 4;
  int head_m = 0, head_n = 0, tail_m = 0, tail_n = 0, n_pend = 0;  // tiles fetched ahead of their epilogue (<= 2)

  auto next = [&](H1Src &s) -> bool {
    if (item >= range_len) return false;
    int mt, nt;
    supertile_order<4, 8>(range_base + item, n_ntiles, n_mtiles, nt, mt);
    item += per_xcd;
    if (n_pend == 0) head_m = mt, head_n = nt;
    else tail_m = mt, tail_n = nt;
    ++n_pend;
    long long rows_left;
    if (w8 < 4) {  // waves 0-3 stage the 256 W rows (LDS rows [0, 256)), waves 4-7 the 256 activation rows
      s.src = reinterpret_cast<const char *>(W) + (size_t)nt * 256 * (size_t)row_bytes;
      rows_left = (long long)N - (long long)nt * 256;
    } else {
      s.src = reinterpret_cast<const char *>(A) + (size_t)mt * TM * (size_t)row_bytes;
      rows_left = (long long)M - (long long)mt * TM;
      if (rows_left > TM) rows_left = TM;
    }
    if (rows_left > 256) rows_left = 256;
    s.bytes = (unsigned int)(rows_left * row_bytes);
    return true;
  };
  auto begin = [&]() {};
  auto emit = [&](f32x4 (&acc)[4][NI]) {
    const int mt = head_m, nt = head_n;
    head_m = tail_m, head_n = tail_n;
    --n_pend;
    // the lane coordinates are re-derived per tile from opaque copies: left visible, the compiler hoists the 64 per-block store /
    // residual offsets (tile-invariant) out of the tile loop and spills them (17-25 registers per lane in the residual variants)
    int r16 = lane & 15, kq = lane >> 4;
    asm volatile("" : "+v"(r16), "+v"(kq));
    const int m0 = mt * TM + 16 * NI * wn + r16;         // + 16 ni
    const bool interior = (mt + 1) * TM <= M && (nt + 1) * 256 <= N;
    const int ldc4 = (int)ldc * 4, ldr4 = (int)ldr * 4;
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(SPLIT_OUT ? nullptr : C + (size_t)mt * TM * ldc + (size_t)nt * 256);
    const __amdgpu_buffer_rsrc_t rr =   // LOGITS: the hidden states' columns d = 256 (nt mod 3) .. + 255
        tile_rsrc(residual ? residual + (size_t)mt * TM * ldr + (size_t)(LOGITS ? nt % 3 : nt) * 256 : nullptr);
    const __amdgpu_buffer_rsrc_t ri =
        tile_rsrc((SPLIT_OUT || BOTH) ? so.img + (size_t)mt * TM * 2 * so.np + (size_t)nt * 256 : nullptr);
    const __amdgpu_buffer_rsrc_t rw = tile_rsrc(ew + nt * 256);
    const __amdgpu_buffer_rsrc_t rb = tile_rsrc(bias ? bias + nt * 256 : nullptr);
    const int ncol = 128 * grp + 64 * wm + 4 * kq;        // column of the lane's quad in block column mi = 0 (+ 16 mi)
    const int nvalid = N - nt * 256;                      // columns of this tile that exist (edge tiles)
    // per-row scalars of the lane's NI rows, packed four int8 per register: A-row exponents, output exponents (split out)
    unsigned int emp[2] = {0u, 0u}, eop[2] = {0u, 0u};
    float rsv[NI];                                       // the rows' scales (folded norm: so.rscale), 1 otherwise
    const int mlim = interior ? 0x7fffffff : M;          // row m exists iff m < mlim
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int m = m0 + 16 * ni;
      emp[ni >> 2] |= ((unsigned int)(unsigned char)ea[min(m, M - 1)]) << (8 * (ni & 3));
      rsv[ni] = so.rscale ? so.rscale[min(m, M - 1)] : 1.0f;
      if constexpr (BOTH) {
        const float bound = fmaf(so.anorm ? so.anorm[min(m, M - 1)] : so.anorm_const, so.wnorm_max, so.xbound[min(m, M - 1)]);
        const int e = pow2_exp(bound * 1.001f);
        eop[ni >> 2] |= ((unsigned int)(unsigned char)(signed char)e) << (8 * (ni & 3));
        if (nt == 0 && grp == 0 && wm == 0 && kq == 0 && m < mlim) {  // one writer per row
          so.exps[m] = (signed char)e;
          so.obound[m] = bound * 1.0001f;
        }
      }
      if constexpr (SPLIT_OUT) {
        const float bound = fmaf(so.anorm ? so.anorm[min(m, M - 1)] : so.anorm_const, so.wnorm_max, so.babs_max);
        const int e = pow2_exp(bound * 1.001f);
        eop[ni >> 2] |= ((unsigned int)(unsigned char)(signed char)e) << (8 * (ni & 3));
        if (nt == 0 && grp == 0 && wm == 0 && kq == 0 && m < mlim) {  // one writer per row
          so.exps[m] = (signed char)e;
          if (so.norms) so.norms[m] = bound * so.onorm_scale;
        }
      }
    }
    auto sx8 = [](unsigned int packed, int i) { return (int)(packed << (24 - 8 * i)) >> 24; };   // sign-extended byte i
    // every load is issued BEFORE the stores it does not depend on (vmcnt retires in issue order): the column exponents of the
    // four block columns up front; bias of block column mi + 1 and the residual rows of step i + 1 before the stores of step i.
    // A step = two blocks (one block column mi, rows 16 (2 p) and 16 (2 p + 1)): STEPS = 2 NI steps per tile (sixteen at NI = 8).
    unsigned int wq[4];   // four int8 exponents per block column
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int c = ncol + 16 * mi;
      wq[mi] = __builtin_amdgcn_raw_buffer_load_b32(rw, (interior || c < nvalid) ? c : OOB, 0, 0);
    }
    // Step order (round 5): ROW-block major -- step i = (row pair pr = i / 4, block column mi = i % 4) -- so that the four 64-byte
    // pieces of a row's 256 bytes in this wave's columns leave in four consecutive store instructions (block-column major had
    // them a quarter of the epilogue apart: half-written 128-byte lines waiting in L2 while 32 CUs of the XCD write 8 MB).
    // The bias quads of the four block columns are held for the whole tile (all four loaded up front).
    constexpr int RD = 1;                      // residual rows are requested RD steps ahead of their use (a ring of RD + 1 steps)
    u32x4 bq[4], res[RD + 1][HAS_RES ? 2 : 1];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const int c = ncol + 16 * mi;
      bq[mi] = u32x4{0u, 0u, 0u, 0u};
      if (bias) bq[mi] = __builtin_amdgcn_raw_buffer_load_b128(rb, (interior || c < nvalid) ? c * 4 : OOB, 0, 0);
    }
    auto load_step = [&](int i, u32x4 (&r)[HAS_RES ? 2 : 1]) {
      const int mi = i & 3, pr = i >> 2;
      const int c = ncol + 16 * mi;
      const bool cok = interior || c < nvalid;
      if constexpr (HAS_RES) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int ni = 2 * pr + n2;
          const int rowt = 16 * NI * wn + 16 * ni + r16;
          r[n2] = __builtin_amdgcn_raw_buffer_load_b128(rr, (cok && m0 + 16 * ni < mlim) ? rowt * ldr4 + c * 4 : OOB, 0, 0);
        }
      }
    };
#pragma unroll
    for (int i = 0; i < RD && i < STEPS; ++i) load_step(i, res[i]);
    float part[LOGITS ? 8 : 1];   // LOGITS: the lane's chain per row 16 ni + r16, sixteen terms in order (mi, j)
    if constexpr (LOGITS) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) part[ni] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
      const int mi = i & 3, pr = i >> 2;
      const int c = ncol + 16 * mi;
      const bool cok = interior || c < nvalid;
      if (i + RD < STEPS) load_step(i + RD, res[(i + RD) & RD]);
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int ni = 2 * pr + n2;
        const int rowt = 16 * NI * wn + 16 * ni + r16;   // row inside the tile
        const bool ok = cok && m0 + 16 * ni < mlim;
        const int em = sx8(emp[ni >> 2], ni & 3);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x = __fmul_rn(ldexpf(acc[mi][ni][j], -(em + sx8(wq[mi], j))), rsv[ni]) + __uint_as_float(bq[mi][j]);   // (no fma with the bias: the latency kernels round the product too)
          x = act_fn<ACT>(x);
          if constexpr (LOGITS) part[ni] = fmaf(__uint_as_float(res[i & RD][n2][j]) * so.alpha, x, part[ni]);
          else if constexpr (HAS_RES) x += __uint_as_float(res[i & RD][n2][j]);
          v[j] = x;
        }
        if constexpr (LOGITS) {
          (void)rowt; (void)ok;
          // pin the step's sums here: with no store to anchor them the compiler sinks all sixteen steps' arithmetic behind the
          // last step's loads and keeps 32 loaded quads alive (176 bytes of scratch per lane)
          asm volatile("" : "+v"(part[ni]));
        } else if constexpr (BOTH) {
          const int eo = sx8(eop[ni >> 2], ni & 3);
          h4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xs = ldexpf(v[j], eo);
            hi[j] = (_Float16)xs;
            lo[j] = (_Float16)(xs - (float)hi[j]);
          }
          const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, ok ? rowt * ldc4 + c * 4 : OOB, 0, 0);
          const int off = ok ? rowt * (so.np * 4) + c * 2 : OOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ri, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ri, off, so.np * 2, 0);
          const float q = block_ssq(v[0], v[1], v[2], v[3]);
          if (kq == 0 && ok) so.ssq[(size_t)(mt * TM + rowt) * (size_t)(N >> 4) + (size_t)(nt * 16 + 8 * grp + 4 * wm + mi)] = q;
        } else if constexpr (SPLIT_OUT) {
          const int eo = sx8(eop[ni >> 2], ni & 3);
          h4 hi, lo;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xs = ldexpf(v[j], eo);
            hi[j] = (_Float16)xs;
            lo[j] = (_Float16)(xs - (float)hi[j]);
          }
          const int off = ok ? rowt * (so.np * 4) + c * 2 : OOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hi), ri, off, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, lo), ri, off, so.np * 2, 0);
        } else {
          const u32x4 o = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
          __builtin_amdgcn_raw_buffer_store_b128(o, rc, ok ? rowt * ldc4 + c * 4 : OOB, 0, 0);
        }
      }
      // keep the steps apart: left alone the scheduler hoists the residual loads of ALL steps to the top (64 registers of
      // loads in flight on top of the 128 accumulators: the residual variants spilled 17-25 registers per lane)
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (LOGITS) {
      // the wave's partial of each of its rows: (kq 0 + kq 1) + (kq 2 + kq 3); lane kq = 0 stores it
      int r16b = lane & 15;                   // a second opaque copy: the eight row indices are not kept alive through the steps
      asm volatile("" : "+v"(r16b));
      const int mb = mt * TM + 16 * NI * wn + r16b;
      float *cp = C + ((size_t)nt * M + mb) * 4 + 2 * grp + wm;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        float p = part[ni];
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        if (kq == 0 && mb + 16 * ni < mlim) cp[64 * ni] = p;
      }
    }
  };
  split_tile_stream16<NI>(row_bytes, kp * 2, kp / 32, lds, next, begin, emit);
}

// Few outputs (the latency path), 16x16x32 form: one workgroup per 32 x 32 outputs = 2 x 2 blocks, wave 0 multiplies; per 32 k
// and block a_lo w_hi, a_hi w_hi, a_hi w_lo -- the sequence of split_tile_stream16, hence the same bits.  Staging, ring and
// swizzle as gemm_split_skinny_kernel (the swizzle key (r >> 1) & 7 is conflict-free for this fragment shape too: the eight
// row pairs of a ds_read_b128 lane group land in eight different slots).
template <int TB>   // tile = 16 TB x 16 TB outputs: TB = 2 (four blocks, one per wave) or 1 (one block: below ~128 32 x 32 tiles the
                    // 16 x 16 form puts four times as many workgroups -- CUs streaming operands -- on the same GEMM)
__device__ __forceinline__ void skinny16_body(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, const SplitOut &so) {
  constexpr int T = 16 * TB;                        // tile edge
  constexpr int SLAB = T * 32;                      // floats: T rows x 128 B
  constexpr int CHUNK = 4 * SLAB;                   // a_hi, a_lo, w_hi, w_lo
  constexpr int PIECES = 2 * TB;                    // 8-row DMA pieces per slab
  __shared__ __attribute__((aligned(16))) float sm[SK_RING * CHUNK];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * T, m0 = blockIdx.y * T;
  const bool is_w = wave >= 2;
  const int rows_left = (is_w ? N - n0 : M - m0) - 1;
  const __amdgpu_buffer_rsrc_t rsrc =
      tile_rsrc((is_w ? W + (size_t)n0 * 2 * kp : A + (size_t)m0 * 2 * kp) + ((wave & 1) ? kp : 0));
  int voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int r = 8 * i + (lane >> 3);
    voff[i] = min(r, rows_left) * 4 * kp + 16 * ((lane & 7) ^ ((r >> 1) & 7));
  }
  const int nchunk = (kp + 63) / 64;
  auto issue = [&](int c) {
    float *dst = sm + (c & (SK_RING - 1)) * CHUNK + wave * SLAB;
    const int soff = 2 * min(64 * c, kp - 64);
    const bool past = c >= nchunk;
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(dst + 256 * i), 16,
                                               past ? OOB : voff[i], soff, 0, 0);
  };
  // Every wave stages one slab (above); TB = 2: it also owns one of the tile's four 16 x 16 blocks, wave w = block (mi, ni) =
  // (w >> 1, w & 1): W rows n0 + 16 mi + [0, 16), activation rows m0 + 16 ni + [0, 16).  A block is one accumulation chain of
  // three MFMAs per 32 k -- with one wave multiplying all four (the first version) the chain of K = 3072 took 25 us.
  // TB = 1: the single block is computed by every wave, stored by wave 0.
  const int mi = TB == 2 ? wave >> 1 : 0, ni = TB == 2 ? wave & 1 : 0;
  const bool storer = TB == 2 || wave == 0;
  // the epilogue's operands, fetched ahead of the stream: row m0 + 16 ni + r16, column quad n0 + 16 mi + 4 kq
  const int m = m0 + 16 * ni + r16;
  const bool mok = m < M;
  const int mc = min(m, M - 1);
  const int em = ea[mc];
  int eo = 0;
  // folded T5LayerNorm (SplitOut): the row's scale, given or from its block sums; a producer of the residual stream (C AND image)
  // takes the image's exponent from the bound it carries forward
  const float rs = so.rscale ? so.rscale[mc] : (so.rparts ? row_rscale(so.rparts + (size_t)mc * so.nparts, so.nparts, so.rs_dim, so.rs_eps) : 1.0f);
  const bool both = so.img && C;
  if (so.img) {
    const float an = so.anorm ? so.anorm[mc] : so.anorm_const;
    const float bound = both ? fmaf(an, so.wnorm_max, so.xbound[mc]) : fmaf(an, so.wnorm_max, so.babs_max);
    eo = pow2_exp(bound * 1.001f);
    if (storer && mi == 0 && mok && n0 == 0 && kq == 0) {   // one writer per row
      so.exps[m] = (signed char)eo;
      if (both) so.obound[m] = bound * 1.0001f;
      else if (so.norms) so.norms[m] = bound * so.onorm_scale;
    }
  }
  const int nq_ = min(n0 + 16 * mi + 4 * kq, N - 4);
  const int pw = *reinterpret_cast<const int *>(ew + nq_);
  const f32x4 pb = bias ? *reinterpret_cast<const f32x4 *>(bias + nq_) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 pr = residual ? *reinterpret_cast<const f32x4 *>(residual + (size_t)mc * ldr + nq_) : f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // fragment of rows 16 i + r16, unit uu of the chunk: piece 4 uu + kq of the 128-byte row, slot ^ key(row)
  int foff_w[2], foff_a[2];
#pragma unroll
  for (int uu = 0; uu < 2; ++uu) {
    const int rw = 16 * mi + r16, ra = 16 * ni + r16;
    foff_w[uu] = rw * 32 + 4 * ((4 * uu + kq) ^ ((rw >> 1) & 7));
    foff_a[uu] = ra * 32 + 4 * ((4 * uu + kq) ^ ((ra >> 1) & 7));
  }
  // Two chunks (128 k) per workgroup barrier: chunks c, c + 1 are read while c + 2 .. c + 7 fly (six chunks in flight per
  // workgroup: 96 KiB at TB = 2, 48 KiB at TB = 1, where two or three workgroups share a CU).
#pragma unroll
  for (int c = 0; c < SK_RING - 2; ++c) issue(c);
  for (int c = 0; c < nchunk; c += 2) {
    // chunks c and c + 1 landed (this wave's pieces: all but the four chunks issued after them), then everybody's; every wave
    // has read chunks c - 2 and c - 1, whose ring slots the two issues below overwrite
    if constexpr (TB == 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(c + SK_RING - 2);
    issue(c + SK_RING - 1);
    f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const float *b = sm + ((c + cc) & (SK_RING - 1)) * CHUNK;
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        ah[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + 0 * SLAB + foff_a[uu]);
        al[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + 1 * SLAB + foff_a[uu]);
        wh[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + 2 * SLAB + foff_w[uu]);
        wl[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + 3 * SLAB + foff_w[uu]);
      }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      if (c + cc >= nchunk) continue;                              // odd chunk count: the pair's second chunk does not exist
      const int k0 = 64 * (c + cc);
      const int skip = k0 > kp - 64 ? (k0 - (kp - 64)) / 32 : 0;   // units of the clamped last chunk already multiplied
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        if (uu < skip) continue;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[2 * cc + uu], al[2 * cc + uu], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[2 * cc + uu], ah[2 * cc + uu], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[2 * cc + uu], ah[2 * cc + uu], acc, 0, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the empty tail pieces: nothing may target LDS past the loop
  const int n = n0 + 16 * mi + 4 * kq;         // n, N, ldc multiples of 4: a quad is whole or absent, and 16-byte aligned
  if (so.ssq && storer) {   // block sums of the OUTPUT rows (producer of the residual stream): all 64 lanes take part in the shuffles
    float w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float x = __fmul_rn(ldexpf(acc[j], -(em + (int)(signed char)(pw >> (8 * j)))), rs);
      if (bias) x += pb[j];
      x = act == 1 ? act_fn<1>(x) : (act == 2 ? act_fn<2>(x) : x);
      if (residual) x += pr[j];
      w4[j] = x;
    }
    const float q = block_ssq(w4[0], w4[1], w4[2], w4[3]);
    if (kq == 0 && mok && n < N) so.ssq[(size_t)m * (size_t)(N >> 4) + (size_t)((n0 >> 4) + mi)] = q;
  }
  if (!storer || n >= N || !mok) return;
  const int a8 = act;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = __fmul_rn(ldexpf(acc[j], -(em + (int)(signed char)(pw >> (8 * j)))), rs);
    if (bias) x += pb[j];
    x = a8 == 1 ? act_fn<1>(x) : (a8 == 2 ? act_fn<2>(x) : x);
    if (residual) x += pr[j];
    v[j] = x;
  }
  if (both) {      // every lane of the block is here (storer waves only; rows / column quads past the edge left above): see below
    *reinterpret_cast<f32x4 *>(C + (size_t)m * ldc + n) = v;
  }
  if (so.img) {
    f16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xs = ldexpf(v[j], eo);
      hi[j] = (_Float16)xs;
      lo[j] = (_Float16)(xs - (float)hi[j]);
    }
    _Float16 *o = so.img + (size_t)m * 2 * so.np + n;
    *reinterpret_cast<f16x4 *>(o) = hi;
    *reinterpret_cast<f16x4 *>(o + so.np) = lo;
  } else {
    *reinterpret_cast<f32x4 *>(C + (size_t)m * ldc + n) = v;
  }
}

__global__ __launch_bounds__(256) void gemm_split16_skinny_kernel(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, SplitOut so) {
  skinny16_body<2>(A, ea, M, W, ew, N, kp, C, ldc, bias, residual, ldr, act, so);
}
__global__ __launch_bounds__(256) void gemm_split16_skinny_t16_kernel(
    const _Float16 *__restrict__ A, const signed char *__restrict__ ea, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, int kp, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, SplitOut so) {
  skinny16_body<1>(A, ea, M, W, ew, N, kp, C, ldc, bias, residual, ldr, act, so);
}

// The latency path's norm + GEMM in one kernel: y = act(rmsnorm(x) W^T + b) (+ residual) for a handful of rows at the t5-base
// width.  T5 feeds every T5LayerNorm into exactly one projection (q|k|v, the cross-attention q, wi); as two kernels the norm is a
// 3 us launch + boundary in front of a 5.5 us GEMM, sixty times per tower pass.  Here every workgroup (16 x 16 outputs, as
// skinny16_body<1>) normalises its 16 rows itself -- one wave per row, the arithmetic of rmsnorm_split_kernel's 768-wide path in
// the same order, so the (hi, lo) image it builds in LDS (two planes of [16][768] halves, rows 1664 B apart: 128 B of padding
// keep two rows on different bank halves, 16-byte chunks XORed by (row >> 1) & 7) is the image that kernel would have written,
// bit for bit -- while the W stream is already in flight (all of it at K = 768: the ring holds 16 chunks of 64 k).  The
// accumulation chain per output is skinny16_body's, so a row keeps its bits in any batch and on either path.
constexpr int NG_RING = 16;                      // W chunks of 64 k in LDS: [hi | lo][16 rows][128 B] = 4 KiB each
constexpr int NG_AROW = 1664;                    // bytes between rows of an A plane (768 halves + 128 B)
constexpr int NG_APLANE = 16 * NG_AROW;
constexpr size_t NG_LDS = (size_t)NG_RING * 4096 + 2 * NG_APLANE + 16 * 8;
__global__ __launch_bounds__(1024) void gemm_rmsnorm_split16_kernel(
    const float *__restrict__ X, long long ldx, const float *__restrict__ lnw, float eps, int M, const _Float16 *__restrict__ W,
    const signed char *__restrict__ ew, int N, float *__restrict__ C, long long ldc, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, SplitOut so) {
  constexpr int kp = 768;
  extern __shared__ __attribute__((aligned(16))) char ng[];
  char *ring = ng;                                 // NG_RING x 4 KiB
  char *ahi = ng + NG_RING * 4096, *alo = ahi + NG_APLANE;
  int *sexp = reinterpret_cast<int *>(alo + NG_APLANE);          // [16] row exponents
  float *snorm = reinterpret_cast<float *>(sexp + 16);           // [16] row norms (rounded up)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r16 = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
  // Sixteen waves: one per row of the norm (a wave per row is what fixes the order of its sums; four rows per wave in turn
  // made the norm 3 us of every workgroup).  W staging by waves 0-3: wave w moves plane (w >> 1) (hi | lo), rows 8 (w & 1) .. + 7
  // of every chunk, one 1-KiB piece per chunk and wave; wave 0 multiplies; waves 4-15 only keep the barriers company.
  const bool stager = wave < 4;
  const __amdgpu_buffer_rsrc_t rsrc = tile_rsrc(W + (size_t)n0 * 2 * kp + (((wave >> 1) & 1) ? kp : 0));
  const int wr = 8 * (wave & 1) + (lane >> 3);
  const int voff = min(wr, N - n0 - 1) * 4 * kp + 16 * ((lane & 7) ^ ((wr >> 1) & 7));
  constexpr int nchunk = kp / 64;
  // the epilogue's column operands first (oldest in the vmcnt queue: the counted waits below then only ever see W pieces)
  const int nq_ = min(n0 + 4 * kq, N - 4);
  const int pw = *reinterpret_cast<const int *>(ew + nq_);
  const f32x4 pb = bias ? *reinterpret_cast<const f32x4 *>(bias + nq_) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 pr = residual ? *reinterpret_cast<const f32x4 *>(residual + (size_t)min(m0 + r16, M - 1) * ldr + nq_)
                            : f32x4{0.f, 0.f, 0.f, 0.f};
  auto issue = [&](int c) {
    if (!stager) return;
    char *dst = ring + (c & (NG_RING - 1)) * 4096 + (wave >> 1) * 2048 + (wave & 1) * 1024;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)dst, 16, c >= nchunk ? OOB : voff,
                                             2 * 64 * (c < nchunk ? c : 0), 0, 0);
  };
#pragma unroll
  for (int c = 0; c < NG_RING - 2; ++c) issue(c);
  // the norm: wave w takes row w (clamped to the last real row: duplicates are never stored)
  {
    const float4 *wv = reinterpret_cast<const float4 *>(lnw);
    const float4 *xr = reinterpret_cast<const float4 *>(X + (size_t)min(m0 + wave, M - 1) * ldx);
    float4 g[3], v[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = wv[lane + 64 * j], v[j] = xr[lane + 64 * j];
    const int r = wave;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      ss = fmaf(v[j].x, v[j].x, ss); ss = fmaf(v[j].y, v[j].y, ss); ss = fmaf(v[j].z, v[j].z, ss); ss = fmaf(v[j].w, v[j].w, ss);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    const float denom = sqrtf(ss / (float)kp + eps);
    float mx = 0.f, s2 = 0.f;
    float4 y[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      y[j] = make_float4(g[j].x * (v[j].x / denom), g[j].y * (v[j].y / denom), g[j].z * (v[j].z / denom), g[j].w * (v[j].w / denom));
      mx = fmaxf(fmaxf(mx, fmaxf(fabsf(y[j].x), fabsf(y[j].y))), fmaxf(fabsf(y[j].z), fabsf(y[j].w)));
      s2 = fmaf(y[j].x, y[j].x, fmaf(y[j].y, y[j].y, fmaf(y[j].z, y[j].z, fmaf(y[j].w, y[j].w, s2))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mx = fmaxf(mx, __shfl_xor(mx, off));
      s2 += __shfl_xor(s2, off);
    }
    const int e = pow2_exp(mx);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      h4 hi, lo;
      split4(y[j], e, hi, lo);
      const int q4 = lane + 64 * j;                                     // elements 4 q4 .. 4 q4 + 3 = half of 16-byte chunk q4 >> 1
      const int off = r * NG_AROW + ((((q4 >> 1)) ^ ((r >> 1) & 7)) << 4) + ((q4 & 1) << 3);
      *reinterpret_cast<h4 *>(ahi + off) = hi;
      *reinterpret_cast<h4 *>(alo + off) = lo;
    }
    if (lane == 0) {
      sexp[r] = e;
      snorm[r] = sqrtf(s2) * 1.0001f;
    }
  }
  __syncthreads();
  // the epilogue's operands: row m0 + r16, column quad n0 + 4 kq
  const int m = m0 + r16;
  const bool mok = m < M;
  const int em = sexp[r16];
  int eo = 0;
  if (so.img) {
    eo = pow2_exp(fmaf(snorm[r16], so.wnorm_max, so.babs_max) * 1.001f);
    if (wave == 0 && mok && n0 == 0 && kq == 0) {
      so.exps[m] = (signed char)eo;
      if (so.norms) so.norms[m] = fmaf(snorm[r16], so.wnorm_max, so.babs_max) * so.onorm_scale;
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const int key = (r16 >> 1) & 7;
  const int aoff = r16 * NG_AROW, woff = r16 * 128;
  for (int c = 0; c < nchunk; c += 2) {
    // chunks c, c + 1 landed (this wave's pieces: all but the twelve issued after them), then everybody's; chunks c - 2, c - 1 read
    asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    issue(c + NG_RING - 2);
    issue(c + NG_RING - 1);
    if (wave != 0) continue;
    f16x8 ah[4], al[4], wh[4], wl[4];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const char *b = ring + ((c + cc) & (NG_RING - 1)) * 4096;
#pragma unroll
      for (int uu = 0; uu < 2; ++uu) {
        const int pc = 4 * uu + kq;                                        // 16-byte piece of the chunk's 128-byte row
        ah[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(ahi + aoff + (((8 * (c + cc) + pc) ^ key) << 4));
        al[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(alo + aoff + (((8 * (c + cc) + pc) ^ key) << 4));
        wh[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + woff + ((pc ^ key) << 4));
        wl[2 * cc + uu] = *reinterpret_cast<const f16x8 *>(b + 2048 + woff + ((pc ^ key) << 4));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], al[i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[i], ah[i], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[i], ah[i], acc, 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int n = n0 + 4 * kq;
  if (wave != 0 || n >= N || !mok) return;
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x = ldexpf(acc[j], -(em + (int)(signed char)(pw >> (8 * j))));
    if (bias) x += pb[j];
    x = act == 1 ? act_fn<1>(x) : (act == 2 ? act_fn<2>(x) : x);
    if (residual) x += pr[j];
    v[j] = x;
  }
  if (so.img) {
    f16x4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float xs = ldexpf(v[j], eo);
      hi[j] = (_Float16)xs;
      lo[j] = (_Float16)(xs - (float)hi[j]);
    }
    _Float16 *o = so.img + (size_t)m * 2 * so.np + n;
    *reinterpret_cast<f16x4 *>(o) = hi;
    *reinterpret_cast<f16x4 *>(o + so.np) = lo;
  } else {
    *reinterpret_cast<f32x4 *>(C + (size_t)m * ldc + n) = v;
  }
}

// rs[m] = rsqrt(mean(x_m^2) + eps) from the row's block sums (folded T5LayerNorm, SplitOut): one lane per row
__global__ __launch_bounds__(256) void row_rscale_kernel(const float *__restrict__ parts, long long m, int nparts, float dim, float eps,
                                                        float *__restrict__ out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r < m) out[r] = row_rscale(parts + (size_t)r * nparts, nparts, dim, eps);
}

// Where the residual stream STARTS (token embeddings): rows of x f32 [m, k] -> split image with the row's own exponent, its bound
// (the l2 norm, rounded up: >= max |x|) and the block sums of squares in block_ssq's order (a 16-column block = four lanes of the
// wave: lane & 3 plays kq).  k % 16 == 0, k <= 1024 here (one float4 per lane and 256 columns).  One wave per row.
__global__ __launch_bounds__(256) void split_rows_ssq_kernel(const float *__restrict__ x, long long ldx, long long m, int k, int kp,
                                                            _Float16 *__restrict__ img, signed char *__restrict__ exps,
                                                            float *__restrict__ bound, float *__restrict__ ssq) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= m) return;
  const int lane = threadIdx.x & 63;
  const float *xr = x + (size_t)r * ldx;
  float4 v[4];
  float mx = 0.f, ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * lane + 256 * j;
    v[j] = c < k ? *reinterpret_cast<const float4 *>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
    float q = fmaf(v[j].w, v[j].w, fmaf(v[j].z, v[j].z, fmaf(v[j].y, v[j].y, v[j].x * v[j].x)));
    q += __shfl_xor(q, 1);
    q += __shfl_xor(q, 2);
    if ((lane & 3) == 0 && c < k) ssq[(size_t)r * (k >> 4) + (c >> 4)] = q;
    ss += q;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, off));
    ss += __shfl_xor(ss, off);
  }
  const int e = pow2_exp(mx);
  _Float16 *o = img + (size_t)r * 2 * kp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < kp) {
      h4 hi = {0, 0, 0, 0}, lo = {0, 0, 0, 0};
      if (c < k) split4(v[j], e, hi, lo);
      *reinterpret_cast<h4 *>(o + c) = hi;
      *reinterpret_cast<h4 *>(o + kp + c) = lo;
    }
  }
  if (lane == 0) {
    exps[r] = (signed char)e;
    bound[r] = sqrtf(ss * 0.25f) * 1.001f + mx * 1e-6f;   // ss counts every block four times (all four lanes add the block's sum)
  }
}

// crossover measured with tools/bench_skinny_crossover.py (profiles/r02_skinny_crossover.txt): the latency kernel takes
// ~13 us (K = 768) per round of 256 workgroups, the tile stream ~50 us for any grid below one wave of tiles
constexpr long long SPLIT_SKINNY_MAX_OUTPUTS = 1500000;

// Activation blocks per wave (NI) of the tile stream for a GEMM of m rows x n_ntiles 256-column tiles on n_cu CUs: the tile
// height 32 NI whose rounds x time-per-tile is least.  A 128-row tile costs ~0.61, a 64-row tile ~0.41 of a 256-row tile's time
// (all W slabs still cross LDS; tools/bench_skinny_crossover.py, profiles/r05_gemm_tile_rows.txt); ties go to the larger tile.
// The choice changes no bit of any row.  MEVI_GEMM_TILE_ROWS=256|128|64 pins it (A/B).
inline int split_tile_blocks(int64_t m, int64_t n_ntiles, int n_cu, double *cost_out = nullptr) {
  static const int pinned = [] { const char *e = getenv("MEVI_GEMM_TILE_ROWS"); return e ? atoi(e) : 0; }();
  static const double t4 = [] { const char *e = getenv("MEVI_GEMM_TILE_T128"); return e ? atof(e) : 0.61; }();
  static const double t2 = [] { const char *e = getenv("MEVI_GEMM_TILE_T64"); return e ? atof(e) : 0.41; }();
  const double cost[3] = {1.0, t4, t2};
  int best = 8;
  double best_t = 0.0;
  for (int i = 0, ni = 8; i < 3; ++i, ni >>= 1) {
    if ((pinned == 256 || pinned == 128 || pinned == 64) && pinned != 32 * ni) continue;
    const int64_t tiles = ((m + 32 * ni - 1) / (32 * ni)) * n_ntiles;
    const double t = (double)((tiles + n_cu - 1) / n_cu) * cost[i];
    if (best_t == 0.0 || t < best_t * 0.97) best = ni, best_t = t;
  }
  if (cost_out) *cost_out = best_t;
  return best;
}

// Rows [0, m1) in 256-row tiles that fill WHOLE rounds of the device, the remaining rows as a GEMM of their own (with the tile
// height that suits them): 819 tiles of a 69 800 x 768 projection are 3.2 rounds of 256 CUs -- four rounds' time -- where three
// full rounds + 102 half-height tiles take 3.6.  Returns m1 (0: one launch).  A second launch costs ~6 us (launch + its
// prologue), 0.065 of a K = 768 tile; rows keep their bits (the tile heights agree bit for bit).  MEVI_GEMM_SPLIT_M=0: off (A/B).
inline int64_t split_rows_plan(int64_t m, int64_t n_ntiles, int n_cu, int kp) {
  static const bool off = [] { const char *e = getenv("MEVI_GEMM_SPLIT_M"); return (e && atoi(e) == 0) || getenv("MEVI_GEMM_TILE_ROWS"); }();
  if (off) return 0;
  const int64_t tiles8 = ((m + 255) / 256) * n_ntiles;
  const int64_t r8 = (tiles8 + n_cu - 1) / n_cu;
  if (r8 < 2) return 0;
  double single;
  (void)split_tile_blocks(m, n_ntiles, n_cu, &single);
  const double launch = 0.065 * 768.0 / (double)kp;
  int64_t best_m1 = 0;
  double best = single;
  for (int64_t R = r8 - 1; R >= 1 && R >= r8 - 2; --R) {
    const int64_t m1_tiles = R * n_cu / n_ntiles;
    if (m1_tiles == 0 || m1_tiles * 256 >= m) continue;
    double rest;
    (void)split_tile_blocks(m - m1_tiles * 256, n_ntiles, n_cu, &rest);
    const double t = (double)R + rest + launch;
    if (t < best * 0.97) best = t, best_m1 = m1_tiles * 256;
  }
  return best_m1;
}

}  // namespace
}  // namespace mevi

using namespace mevi;

// Kp: whole 32-wide slabs, at least two (the tile stream prefetches two super-units ahead)
extern "C" int64_t mevi_split_kp(int64_t k) { return k <= 64 ? 64 : (k + 31) / 32 * 32; }

extern "C" int mevi_split_rows_f16(const float *x, int64_t ldx, int64_t m, int64_t k, void *img, int8_t *exps,
                                   float *norms, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(m >= 0 && k > 0, MEVI_ERR_INVALID_ARG, "split_rows: bad shape");
  if (m == 0) return MEVI_OK;
  MEVI_REQUIRE(x && img && exps, MEVI_ERR_INVALID_ARG, "split_rows: null pointer");
  MEVI_REQUIRE(k % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)img % 16) == 0,
               MEVI_ERR_UNSUPPORTED, "split_rows: k, ldx must be multiples of 4 and x, img 16-byte aligned");
  MEVI_REQUIRE(k < (1LL << 22), MEVI_ERR_UNSUPPORTED, "split_rows: k too large");
  hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, x, (long long)ldx,
                     (long long)m, (int)k, (int)mevi_split_kp(k), reinterpret_cast<_Float16 *>(img), reinterpret_cast<signed char *>(exps), norms);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_rmsnorm_split_f16(const float *x, int64_t ldx, const float *w, float eps, int64_t rows, int64_t dim,
                                      void *img, int8_t *exps, float *norms, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0 && ldx % 4 == 0, MEVI_ERR_INVALID_ARG,
               "rmsnorm_split: dim/ld must be multiples of 4");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(x && w && img && exps && norms, MEVI_ERR_INVALID_ARG, "rmsnorm_split: null pointer");
  hipLaunchKernelGGL(rmsnorm_split_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, (long long)ldx, w,
                     eps, (long long)rows, (int)dim, (int)mevi_split_kp(dim), reinterpret_cast<_Float16 *>(img), reinterpret_cast<signed char *>(exps), norms);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

static int gemm_split_launch(const void *a_img, const int8_t *a_exp, const void *w_img, const int8_t *w_exp,
                             float *c, int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias,
                             const float *residual, int64_t ldr, int act, SplitOut so, void *stream_, int force_ni = 0) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(m >= 0 && n >= 0 && k > 0, MEVI_ERR_INVALID_ARG, "gemm_nt_split: bad shape");
  if (m == 0 || n == 0) return MEVI_OK;
  MEVI_REQUIRE(a_img && a_exp && w_img && w_exp && (c || so.img), MEVI_ERR_INVALID_ARG, "gemm_nt_split: null pointer");
  MEVI_REQUIRE(((uintptr_t)a_img % 16) == 0 && ((uintptr_t)w_img % 16) == 0, MEVI_ERR_INVALID_ARG,
               "gemm_nt_split: images must be 16-byte aligned");
  MEVI_REQUIRE(act >= 0 && act <= 2, MEVI_ERR_INVALID_ARG, "gemm_nt_split: act must be 0 (none), 1 (relu) or 2 (erf gelu)");
  MEVI_REQUIRE(m < (1LL << 31) && n < (1LL << 31) && k < (1LL << 20), MEVI_ERR_UNSUPPORTED, "gemm_nt_split: too large");
  MEVI_REQUIRE(n % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && ((uintptr_t)c % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
                   ((uintptr_t)bias % 16) == 0 && ((uintptr_t)w_exp % 4) == 0,
               MEVI_ERR_UNSUPPORTED, "gemm_nt_split: n, ldc, ldr must be multiples of 4 and c, residual, bias 16-byte (w_exp 4-byte) aligned");
  MEVI_REQUIRE(ldc < (1LL << 20) && ldr < (1LL << 20), MEVI_ERR_UNSUPPORTED, "gemm_nt_split: row stride too large");
  const int kp = (int)mevi_split_kp(k);
  const _Float16 *A = reinterpret_cast<const _Float16 *>(a_img), *W = reinterpret_cast<const _Float16 *>(w_img);
  // MEVI_GEMM_SKINNY_MAX: the crossover in outputs (tools/bench_gemm_split.py measures it; 0 = tile stream only)
  static const long long skinny_max = [] { const char *e = getenv("MEVI_GEMM_SKINNY_MAX"); return e ? atoll(e) : SPLIT_SKINNY_MAX_OUTPUTS; }();
  // MEVI_GEMM_MFMA=32: the 32x32x16 kernels (A/B switch; BOTH kernels change together -- a row's bits are the same in the
  // tile stream and in the latency kernel of one shape, not across shapes)
  static const bool shape32 = [] { const char *e = getenv("MEVI_GEMM_MFMA"); return e && atoi(e) == 32; }();
  if (so.rparts && so.rscale && m * n > skinny_max && force_ni == 0) {
    // the tile stream takes one scale per row (its lanes hold eight rows each): from the block sums, one launch for the whole GEMM
    hipLaunchKernelGGL(row_rscale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, so.rparts, (long long)m, so.nparts, so.rs_dim,
                       so.rs_eps, const_cast<float *>(so.rscale));
    so.rparts = nullptr;
  } else if (so.rparts && m * n <= skinny_max) {
    so.rscale = nullptr;      // the latency kernels compute it themselves (no extra launch)
  } else if (so.rparts && force_ni != 0) {
    so.rparts = nullptr;      // a part of a GEMM whose scales the first call has computed
  }
  if (m * n <= skinny_max) {
    // fewer 32 x 32 tiles than half the CUs: 16 x 16 tiles (MEVI_GEMM_SKINNY_TILE=32 keeps the large tile; same bits)
    static const bool tile32 = [] { const char *e = getenv("MEVI_GEMM_SKINNY_TILE"); return e && atoi(e) == 32; }();
    const bool small16 = !shape32 && !tile32 && ((n + 31) / 32) * ((m + 31) / 32) <= 128;
    const int T = small16 ? 16 : 32;
    typedef void (*skinny_t)(const _Float16 *, const signed char *, int, const _Float16 *, const signed char *, int, int, float *,
                             long long, const float *, const float *, long long, int, SplitOut);
    static const skinny_t skinny[3] = {gemm_split_skinny_kernel, gemm_split16_skinny_t16_kernel, gemm_split16_skinny_kernel};
    hipLaunchKernelGGL(skinny[shape32 ? 0 : (small16 ? 1 : 2)],
                       dim3((unsigned)((n + T - 1) / T), (unsigned)((m + T - 1) / T)), dim3(256), 0,
                       stream, A, reinterpret_cast<const signed char *>(a_exp), (int)m, W, reinterpret_cast<const signed char *>(w_exp), (int)n, kp, c,
                       (long long)ldc, bias, residual, (long long)ldr, act, so);
    MEVI_HIP_CHECK(hipGetLastError());
    return MEVI_OK;
  }
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const int64_t n_ntiles = (n + 255) / 256;
  if (!shape32 && force_ni == 0) {
    const int64_t m1 = split_rows_plan(m, n_ntiles, n_cu, kp);
    if (m1 > 0) {       // whole rounds of 256-row tiles, then the remaining rows on their own (every row-indexed pointer moves on by m1)
      const int st = gemm_split_launch(a_img, a_exp, w_img, w_exp, c, ldc, m1, n, k, bias, residual, ldr, act, so, stream_, 8);
      if (st != MEVI_OK) return st;
      SplitOut so2 = so;
      if (so.img) so2.img = so.img + (size_t)m1 * 2 * so.np, so2.exps = so.exps + m1;
      if (so.norms) so2.norms = so.norms + m1;
      if (so.anorm) so2.anorm = so.anorm + m1;
      if (so.rscale) so2.rscale = so.rscale + m1;
      if (so.rparts) so2.rparts = so.rparts + (size_t)m1 * so.nparts;
      if (so.ssq) so2.ssq = so.ssq + (size_t)m1 * (size_t)(n >> 4);
      if (so.xbound) so2.xbound = so.xbound + m1, so2.obound = so.obound + m1;
      return gemm_split_launch(A + (size_t)m1 * 2 * kp, a_exp + m1, w_img, w_exp, c ? c + (size_t)m1 * ldc : nullptr, ldc, m - m1, n, k, bias,
                               residual ? residual + (size_t)m1 * ldr : nullptr, ldr, act, so2, stream_, -1);
    }
  }
  const int ni = shape32 ? 8 : (force_ni > 0 ? force_ni : split_tile_blocks(m, n_ntiles, n_cu));
  const int tm = 32 * ni;
  const int64_t n_mtiles = (m + tm - 1) / tm;
  MEVI_REQUIRE(n_mtiles * n_ntiles <= 0x7fffffffLL, MEVI_ERR_UNSUPPORTED, "gemm_nt_split: grid too large");
  int64_t grid = n_cu / 8 * 8;  // persistent: one workgroup per CU, a multiple of the 8 XCDs
  if (grid < 8) grid = 8;
  const int64_t tiles = n_mtiles * n_ntiles;
  if (tiles < grid) grid = (tiles + 7) / 8 * 8;
  const size_t lds_bytes = ss_lds_bytes();
  const int a8 = act;
  typedef void (*kern_t)(const _Float16 *, const signed char *, int, const _Float16 *, const signed char *, int, int, float *,
                         long long, const float *, const float *, long long, int, int, SplitOut);
  static const kern_t table[3][3] = {{gemm_split_kernel<0, 0>, gemm_split_kernel<0, 1>, gemm_split_kernel<0, 2>},
                                     {gemm_split_kernel<1, 0>, gemm_split_kernel<1, 1>, gemm_split_kernel<1, 2>},
                                     {gemm_split_kernel<2, 0>, gemm_split_kernel<2, 1>, gemm_split_kernel<2, 2>}};
#define MEVI_SPLIT16_TABLE(NI_)                                                                                                  \
  {{gemm_split16_kernel<0, 0, NI_>, gemm_split16_kernel<0, 1, NI_>, gemm_split16_kernel<0, 2, NI_>},                             \
   {gemm_split16_kernel<1, 0, NI_>, gemm_split16_kernel<1, 1, NI_>, gemm_split16_kernel<1, 2, NI_>},                             \
   {gemm_split16_kernel<2, 0, NI_>, gemm_split16_kernel<2, 1, NI_>, gemm_split16_kernel<2, 2, NI_>}}
  static const kern_t table16[3][3][3] = {MEVI_SPLIT16_TABLE(8), MEVI_SPLIT16_TABLE(4), MEVI_SPLIT16_TABLE(2)};
#undef MEVI_SPLIT16_TABLE
  static const kern_t both16[3] = {gemm_split16_kernel<0, 4, 8>, gemm_split16_kernel<0, 4, 4>, gemm_split16_kernel<0, 4, 2>};
  const bool both = so.img && c;     // the producer of the residual stream: f32 rows + their image + block sums
  MEVI_REQUIRE(!both || (!shape32 && a8 == 0 && residual && so.ssq && so.xbound && so.obound && n % 16 == 0), MEVI_ERR_INVALID_ARG,
               "gemm_nt_split: residual-stream output needs the 16x16x32 kernels, no activation, a residual, n %% 16 == 0");
  MEVI_REQUIRE(!(shape32 && (so.rscale || so.rparts)), MEVI_ERR_UNSUPPORTED, "gemm_nt_split: row scales need the 16x16x32 kernels");
  const int o3 = so.img ? 2 : (residual ? 1 : 0);
  const kern_t fn = shape32 ? table[a8][o3] : (both ? both16[ni == 8 ? 0 : (ni == 4 ? 1 : 2)] : table16[ni == 8 ? 0 : (ni == 4 ? 1 : 2)][a8][o3]);
  MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds_bytes));
  hipLaunchKernelGGL(fn, dim3((unsigned)grid), dim3(PP_THREADS), lds_bytes, stream, A, reinterpret_cast<const signed char *>(a_exp), (int)m, W,
                     reinterpret_cast<const signed char *>(w_exp), (int)n, kp, c, (long long)ldc, bias, residual, (long long)ldr, (int)n_mtiles,
                     (int)n_ntiles, so);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_gemm_nt_split_f32(const void *a_img, const int8_t *a_exp, const void *w_img, const int8_t *w_exp,
                                      float *c, int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias,
                                      const float *residual, int64_t ldr, int act, void *stream) {
  SplitOut so = {};
  return gemm_split_launch(a_img, a_exp, w_img, w_exp, c, ldc, m, n, k, bias, residual, ldr, act, so, stream);
}

// The PAWA head fused (gemm_split16_kernel<0, 3>): part f32 [n / 256][m][4], then mevi_logits_finish_f32 (t5_ops.hip).
// Returns MEVI_ERR_UNSUPPORTED where the unfused pair (this GEMM with bias into f32 + mevi_adaptive_logits_rows_f32) has to
// be used: the latency kernels' shapes and MEVI_GEMM_MFMA=32 -- the bits are the same either way.
extern "C" int mevi_gemm_nt_split_head_supported(int64_t m, int64_t n, int64_t k, int64_t dim) {
  static const long long skinny_max = [] { const char *e = getenv("MEVI_GEMM_SKINNY_MAX"); return e ? atoll(e) : SPLIT_SKINNY_MAX_OUTPUTS; }();
  static const bool shape32 = [] { const char *e = getenv("MEVI_GEMM_MFMA"); return e && atoi(e) == 32; }();
  return dim == 768 && n > 0 && n % 768 == 0 && m > 0 && m * n > skinny_max && !shape32 && k > 0;
}

extern "C" int mevi_gemm_nt_split_head_f32(const void *a_img, const int8_t *a_exp, const void *w_img, const int8_t *w_exp,
                                           int64_t m, int64_t n, int64_t k, const float *bias, const float *s, int64_t lds_,
                                           float alpha, int64_t dim, float *part, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(mevi_gemm_nt_split_head_supported(m, n, k, dim), MEVI_ERR_UNSUPPORTED,
               "gemm_nt_split_head: shape %lld x %lld (dim %lld) not on the fused path", (long long)m, (long long)n, (long long)dim);
  MEVI_REQUIRE(a_img && a_exp && w_img && w_exp && s && part, MEVI_ERR_INVALID_ARG, "gemm_nt_split_head: null pointer");
  MEVI_REQUIRE(((uintptr_t)a_img % 16) == 0 && ((uintptr_t)w_img % 16) == 0 && ((uintptr_t)s % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                   ((uintptr_t)w_exp % 4) == 0 && lds_ % 4 == 0 && lds_ < (1LL << 20),
               MEVI_ERR_INVALID_ARG, "gemm_nt_split_head: alignment");
  MEVI_REQUIRE(m < (1LL << 31) && n < (1LL << 31) && k < (1LL << 20), MEVI_ERR_UNSUPPORTED, "gemm_nt_split_head: too large");
  const int kp = (int)mevi_split_kp(k);
  const int64_t n_mtiles = (m + 255) / 256, n_ntiles = n / 256;
  MEVI_REQUIRE(n_mtiles * n_ntiles <= 0x7fffffffLL, MEVI_ERR_UNSUPPORTED, "gemm_nt_split_head: grid too large");
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  int64_t grid = n_cu / 8 * 8;
  if (grid < 8) grid = 8;
  const int64_t tiles = n_mtiles * n_ntiles;
  if (tiles < grid) grid = (tiles + 7) / 8 * 8;
  SplitOut so = {};
  so.alpha = alpha;
  const size_t lds_bytes = ss_lds_bytes();
  MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_split16_kernel<0, 3>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((gemm_split16_kernel<0, 3>), dim3((unsigned)grid), dim3(PP_THREADS), lds_bytes, stream,
                     reinterpret_cast<const _Float16 *>(a_img), reinterpret_cast<const signed char *>(a_exp), (int)m,
                     reinterpret_cast<const _Float16 *>(w_img), reinterpret_cast<const signed char *>(w_exp), (int)n, kp, part, 0LL, bias, s,
                     (long long)lds_, (int)n_mtiles, (int)n_ntiles, so);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_gemm_nt_split_to_split(const void *a_img, const int8_t *a_exp, const float *a_norm, const void *w_img,
                                           const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                           const float *bias, float bias_abs_max, int act, void *out_img,
                                           int8_t *out_exp, float *out_norm, void *stream) {
  MEVI_REQUIRE(m == 0 || (a_norm && out_img && out_exp), MEVI_ERR_INVALID_ARG, "gemm_nt_split_to_split: null pointer");
  MEVI_REQUIRE(w_norm_max >= 0.f && bias_abs_max >= 0.f, MEVI_ERR_INVALID_ARG, "gemm_nt_split_to_split: negative bound");
  SplitOut so = {};
  so.img = reinterpret_cast<_Float16 *>(out_img);
  so.exps = reinterpret_cast<signed char *>(out_exp);
  so.norms = out_norm;
  so.anorm = a_norm;
  so.np = (int)mevi_split_kp(n);
  so.wnorm_max = w_norm_max;
  so.babs_max = bias_abs_max;
  so.onorm_scale = sqrtf((float)n) * 1.0001f;  // ||row||_2 <= sqrt(n) * max|element|
  return gemm_split_launch(a_img, a_exp, w_img, w_exp, nullptr, 0, m, n, k, bias, nullptr, 0, act, so, stream);
}

// ---- T5LayerNorm folded into the linear layers around it (round 5; SplitOut) ------------------------------------------------------
// The residual stream travels as (f32 rows, split image, bound on max |row|, block sums of squares [M][N / 16]):
//   mevi_split_rows_ssq_f16             where it starts (token embeddings)
//   mevi_gemm_nt_split_residual_stream  x' = x + a W^T: the f32 rows AND their image, bound, block sums, in the GEMM's epilogue
//   mevi_gemm_nt_split_normed_*         act(rsqrt(mean x^2 + eps) (x W'^T) + b): W' = W (.) w_ln, x = the stream's image
// mevi_rmsnorm_split_f16 (one pass over the rows per norm: read 4 B, write 4 B per element) is not launched on this path.
extern "C" int mevi_gemm_norm_fold_supported(int64_t n_stream) {
  static const bool shape32 = [] { const char *e = getenv("MEVI_GEMM_MFMA"); return e && atoi(e) == 32; }();
  return !shape32 && n_stream >= 16 && n_stream % 16 == 0 && n_stream <= 1024;
}

extern "C" int mevi_split_rows_ssq_f16(const float *x, int64_t ldx, int64_t m, int64_t k, void *img, int8_t *exps, float *bound,
                                       float *ssq, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(m >= 0 && mevi_gemm_norm_fold_supported(k), MEVI_ERR_UNSUPPORTED, "split_rows_ssq: k must be a multiple of 16, <= 1024");
  if (m == 0) return MEVI_OK;
  MEVI_REQUIRE(x && img && exps && bound && ssq, MEVI_ERR_INVALID_ARG, "split_rows_ssq: null pointer");
  MEVI_REQUIRE(ldx % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)img % 16) == 0, MEVI_ERR_UNSUPPORTED, "split_rows_ssq: alignment");
  hipLaunchKernelGGL(split_rows_ssq_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, x, (long long)ldx, (long long)m, (int)k,
                     (int)mevi_split_kp(k), reinterpret_cast<_Float16 *>(img), reinterpret_cast<signed char *>(exps), bound, ssq);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_row_rscale_f32(const float *parts, int64_t m, int64_t nparts, float dim, float eps, float *out, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(m >= 0 && nparts > 0, MEVI_ERR_INVALID_ARG, "row_rscale: bad shape");
  if (m == 0) return MEVI_OK;
  MEVI_REQUIRE(parts && out && ((uintptr_t)parts % 16) == 0, MEVI_ERR_INVALID_ARG, "row_rscale: null / unaligned pointer");
  hipLaunchKernelGGL(row_rscale_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, stream, parts, (long long)m, (int)nparts, dim, eps, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

static int fold_consumer(SplitOut &so, const float *parts, int64_t nparts, float rs_dim, float rs_eps, float *rscale_ws) {
  MEVI_REQUIRE(parts && rscale_ws && nparts > 0 && ((uintptr_t)parts % 16) == 0, MEVI_ERR_INVALID_ARG,
               "gemm_nt_split_normed: block sums [m][nparts] (16-byte aligned) and a [m] scale buffer are required");
  so.rparts = parts, so.nparts = (int)nparts, so.rs_dim = rs_dim, so.rs_eps = rs_eps, so.rscale = rscale_ws;
  return MEVI_OK;
}

extern "C" int mevi_gemm_nt_split_normed_f32(const void *a_img, const int8_t *a_exp, const float *parts, int64_t nparts, float rs_dim,
                                             float rs_eps, float *rscale_ws, const void *w_img, const int8_t *w_exp, float *c,
                                             int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias, const float *residual,
                                             int64_t ldr, int act, void *stream) {
  SplitOut so = {};
  if (m > 0) { const int st = fold_consumer(so, parts, nparts, rs_dim, rs_eps, rscale_ws); if (st != MEVI_OK) return st; }
  return gemm_split_launch(a_img, a_exp, w_img, w_exp, c, ldc, m, n, k, bias, residual, ldr, act, so, stream);
}

extern "C" int mevi_gemm_nt_split_normed_to_split(const void *a_img, const int8_t *a_exp, const float *parts, int64_t nparts,
                                                  float rs_dim, float rs_eps, float *rscale_ws, float a_norm_bound, const void *w_img,
                                                  const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                                  const float *bias, float bias_abs_max, int act, void *out_img, int8_t *out_exp,
                                                  float *out_norm, void *stream) {
  MEVI_REQUIRE(m == 0 || (out_img && out_exp), MEVI_ERR_INVALID_ARG, "gemm_nt_split_normed_to_split: null pointer");
  MEVI_REQUIRE(w_norm_max >= 0.f && bias_abs_max >= 0.f && a_norm_bound >= 0.f, MEVI_ERR_INVALID_ARG, "gemm_nt_split_normed_to_split: negative bound");
  SplitOut so = {};
  if (m > 0) { const int st = fold_consumer(so, parts, nparts, rs_dim, rs_eps, rscale_ws); if (st != MEVI_OK) return st; }
  so.img = reinterpret_cast<_Float16 *>(out_img);
  so.exps = reinterpret_cast<signed char *>(out_exp);
  so.norms = out_norm;
  so.anorm = nullptr, so.anorm_const = a_norm_bound;     // ||rsqrt(.) x|| <= sqrt(d): the same bound for every row
  so.np = (int)mevi_split_kp(n);
  so.wnorm_max = w_norm_max;
  so.babs_max = bias_abs_max;
  so.onorm_scale = sqrtf((float)n) * 1.0001f;
  return gemm_split_launch(a_img, a_exp, w_img, w_exp, nullptr, 0, m, n, k, bias, nullptr, 0, act, so, stream);
}

extern "C" int mevi_gemm_nt_split_residual_stream(const void *a_img, const int8_t *a_exp, const float *a_norm, float a_norm_const,
                                                  const float *parts, int64_t nparts, float rs_dim, float rs_eps, float *rscale_ws,
                                                  const void *w_img, const int8_t *w_exp, float w_norm_max, float *c, int64_t ldc,
                                                  int64_t m, int64_t n, int64_t k, const float *residual, int64_t ldr,
                                                  const float *x_bound, void *out_img, int8_t *out_exp, float *out_bound,
                                                  float *out_ssq, void *stream) {
  MEVI_REQUIRE(mevi_gemm_norm_fold_supported(n), MEVI_ERR_UNSUPPORTED, "gemm_nt_split_residual_stream: n must be a multiple of 16, <= 1024");
  MEVI_REQUIRE(m == 0 || (c && residual && x_bound && out_img && out_exp && out_bound && out_ssq), MEVI_ERR_INVALID_ARG,
               "gemm_nt_split_residual_stream: null pointer");
  MEVI_REQUIRE(w_norm_max >= 0.f && a_norm_const >= 0.f, MEVI_ERR_INVALID_ARG, "gemm_nt_split_residual_stream: negative bound");
  SplitOut so = {};
  if (m > 0 && parts) { const int st = fold_consumer(so, parts, nparts, rs_dim, rs_eps, rscale_ws); if (st != MEVI_OK) return st; }
  so.img = reinterpret_cast<_Float16 *>(out_img);
  so.exps = reinterpret_cast<signed char *>(out_exp);
  so.anorm = a_norm, so.anorm_const = a_norm_const;
  so.np = (int)mevi_split_kp(n);
  so.wnorm_max = w_norm_max;
  so.ssq = out_ssq, so.xbound = x_bound, so.obound = out_bound;
  return gemm_split_launch(a_img, a_exp, w_img, w_exp, c, ldc, m, n, k, nullptr, residual, ldr, 0, so, stream);
}

// ---- norm + projection in one launch (the latency path; see gemm_rmsnorm_split16_kernel) -------------------------------------
// Supported: k == 768 (the t5-base / bert-base width), m <= 1024 rows, n % 16 == 0 not required (n % 4 == 0); the caller keeps
// the two-kernel form (mevi_rmsnorm_split_f16 + mevi_gemm_nt_split_*) elsewhere -- same results, bit for bit, either way.
// The fused kernel multiplies in skinny16_body<1>'s order (16 x 16 x 32 MFMA, 16 x 16 tiles), so it stands in for the two-kernel form
// only where gemm_split_launch would pick that kernel: not under MEVI_GEMM_MFMA=32 / MEVI_GEMM_SKINNY_TILE=32, and inside
// MEVI_GEMM_SKINNY_MAX -- the same static switches, read the same way.
extern "C" int mevi_gemm_rmsnorm_supported(int64_t m, int64_t n, int64_t k) {
  static const long long skinny_max = [] { const char *e = getenv("MEVI_GEMM_SKINNY_MAX"); return e ? atoll(e) : SPLIT_SKINNY_MAX_OUTPUTS; }();
  static const bool shape32 = [] { const char *e = getenv("MEVI_GEMM_MFMA"); return e && atoi(e) == 32; }();
  static const bool tile32 = [] { const char *e = getenv("MEVI_GEMM_SKINNY_TILE"); return e && atoi(e) == 32; }();
  return !shape32 && !tile32 && k == 768 && m >= 1 && n >= 4 && n % 4 == 0 && ((n + 31) / 32) * ((m + 31) / 32) <= 128 &&
         m * n <= skinny_max;
}

static int gemm_rmsnorm_launch(const float *x, int64_t ldx, const float *ln_w, float eps, const void *w_img, const int8_t *w_exp,
                               float *c, int64_t ldc, int64_t m, int64_t n, int64_t k, const float *bias, const float *residual,
                               int64_t ldr, int act, SplitOut so, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(mevi_gemm_rmsnorm_supported(m, n, k), MEVI_ERR_UNSUPPORTED, "gemm_rmsnorm: shape %lld x %lld x %lld not on the fused path",
               (long long)m, (long long)n, (long long)k);
  MEVI_REQUIRE(x && ln_w && w_img && w_exp && (c || so.img), MEVI_ERR_INVALID_ARG, "gemm_rmsnorm: null pointer");
  MEVI_REQUIRE(act >= 0 && act <= 2, MEVI_ERR_INVALID_ARG, "gemm_rmsnorm: act must be 0, 1 or 2");
  MEVI_REQUIRE(ldx % 4 == 0 && ldc % 4 == 0 && ldr % 4 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)ln_w % 16) == 0 &&
                   ((uintptr_t)w_img % 16) == 0 && ((uintptr_t)c % 16) == 0 && ((uintptr_t)residual % 16) == 0 &&
                   ((uintptr_t)bias % 16) == 0 && ((uintptr_t)w_exp % 4) == 0,
               MEVI_ERR_UNSUPPORTED, "gemm_rmsnorm: strides must be multiples of 4 and pointers 16-byte (w_exp 4-byte) aligned");
  MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rmsnorm_split16_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)NG_LDS));
  hipLaunchKernelGGL(gemm_rmsnorm_split16_kernel, dim3((unsigned)((n + 15) / 16), (unsigned)((m + 15) / 16)), dim3(1024), NG_LDS, stream,
                     x, (long long)ldx, ln_w, eps, (int)m, reinterpret_cast<const _Float16 *>(w_img),
                     reinterpret_cast<const signed char *>(w_exp), (int)n, c, (long long)ldc, bias, residual, (long long)ldr, act, so);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_gemm_nt_rmsnorm_split_f32(const float *x, int64_t ldx, const float *ln_w, float eps, const void *w_img,
                                              const int8_t *w_exp, float *c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                                              const float *bias, const float *residual, int64_t ldr, int act, void *stream) {
  SplitOut so = {};
  return gemm_rmsnorm_launch(x, ldx, ln_w, eps, w_img, w_exp, c, ldc, m, n, k, bias, residual, ldr, act, so, stream);
}

extern "C" int mevi_gemm_nt_rmsnorm_split_to_split(const float *x, int64_t ldx, const float *ln_w, float eps, const void *w_img,
                                                   const int8_t *w_exp, float w_norm_max, int64_t m, int64_t n, int64_t k,
                                                   const float *bias, float bias_abs_max, int act, void *out_img, int8_t *out_exp,
                                                   float *out_norm, void *stream) {
  MEVI_REQUIRE(out_img && out_exp, MEVI_ERR_INVALID_ARG, "gemm_rmsnorm_to_split: null pointer");
  MEVI_REQUIRE(w_norm_max >= 0.f && bias_abs_max >= 0.f, MEVI_ERR_INVALID_ARG, "gemm_rmsnorm_to_split: negative bound");
  SplitOut so = {};
  so.img = reinterpret_cast<_Float16 *>(out_img);
  so.exps = reinterpret_cast<signed char *>(out_exp);
  so.norms = out_norm;
  so.anorm = nullptr;     // the kernel computes the rows' norms itself
  so.np = (int)mevi_split_kp(n);
  so.wnorm_max = w_norm_max;
  so.babs_max = bias_abs_max;
  so.onorm_scale = sqrtf((float)n) * 1.0001f;
  return gemm_rmsnorm_launch(x, ldx, ln_w, eps, w_img, w_exp, nullptr, 0, m, n, k, bias, nullptr, 0, act, so, stream);
}
