// Small-shape T5 / adaptor operators (everything that is not a GEMM): wave-per-row kernels
// with coalesced float4 access and wavefront (64-lane) reductions.  All f32.
//
//   rmsnorm          T5LayerNorm                      modeling_t5.py:155-171
//   add_layernorm    torch LayerNorm(x + y + c)       nn.TransformerDecoderLayer post-LN (modeling_t5.py:1252-1255)
//   gather_rows      nn.Embedding / beam reorder      modeling_t5.py:718, generation_utils.py:927-934
//   attention        T5Attention / nn.MultiheadAttention for <= 256 keys
//                    modeling_t5.py:374-410 (no 1/sqrt(d) scaling, additive bias + mask, fp32 softmax)
//   adaptive_logits  PAWA head, valid columns only    modeling_t5.py:1607, 1677-1689
//   scale            hidden * d_model^-0.5            modeling_t5.py:1607
#include "common.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace mevi {
namespace {

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// one wave per row, 4 rows per workgroup
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float *__restrict__ x, long long ldx,
                                                     const float *__restrict__ w, float eps, long long rows,
                                                     int dim, float *__restrict__ out, long long ldo) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 *xr = reinterpret_cast<const float4 *>(x + r * ldx);
  float ss = 0.f;
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 v = xr[i];
    ss = fmaf(v.x, v.x, ss); ss = fmaf(v.y, v.y, ss); ss = fmaf(v.z, v.z, ss); ss = fmaf(v.w, v.w, ss);
  }
  ss = wave_sum(ss);
  const float denom = sqrtf(ss / (float)dim + eps);
  float4 *o = reinterpret_cast<float4 *>(out + r * ldo);
  const float4 *wv = reinterpret_cast<const float4 *>(w);
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 v = xr[i], g = wv[i];
    o[i] = make_float4(g.x * (v.x / denom), g.y * (v.y / denom), g.z * (v.z / denom), g.w * (v.w / denom));
  }
}

// out = LayerNorm(x + y + c) * w + b   (y, c optional; biased variance; eps inside the sqrt)
__global__ __launch_bounds__(256) void add_layernorm_kernel(const float *__restrict__ x, long long ldx,
                                                           const float *__restrict__ y, long long ldy,
                                                           const float *__restrict__ c,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ b, float eps,
                                                           long long rows, int dim, float *__restrict__ out,
                                                           long long ldo) {
  extern __shared__ __attribute__((aligned(16))) float srow[];  // 4 rows x dim
  const int wv = threadIdx.x >> 6;
  const long long r = (long long)blockIdx.x * 4 + wv;
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  if (dim == 768) {
    // t5-base / bert-base width: the row in registers, every load issued before the first use (the general form below makes
    // twelve dependent trips per operand: 11.9 us per call on the latency path); same element -> lane map and the same
    // order of additions: same bits
    float v[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) v[j] = x[r * ldx + lane + 64 * j];
    if (y) {
#pragma unroll
      for (int j = 0; j < 12; ++j) v[j] += y[r * ldy + lane + 64 * j];
    }
    if (c) {
#pragma unroll
      for (int j = 0; j < 12; ++j) v[j] += c[lane + 64 * j];
    }
    float wj[12], bj[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) wj[j] = w[lane + 64 * j], bj[j] = b[lane + 64 * j];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) sum += v[j];
    const float mean = wave_sum(sum) / (float)dim;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const float d = v[j] - mean;
      var = fmaf(d, d, var);
    }
    var = wave_sum(var) / (float)dim;
    const float inv = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int j = 0; j < 12; ++j) out[r * ldo + lane + 64 * j] = (v[j] - mean) * inv * wj[j] + bj[j];
    return;
  }
  float *s = srow + (size_t)wv * dim;
  float sum = 0.f;
  for (int i = lane; i < dim; i += 64) {
    float v = x[r * ldx + i];
    if (y) v += y[r * ldy + i];
    if (c) v += c[i];
    s[i] = v;
    sum += v;
  }
  const float mean = wave_sum(sum) / (float)dim;
  float var = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float d = s[i] - mean;
    var = fmaf(d, d, var);
  }
  var = wave_sum(var) / (float)dim;
  const float inv = 1.0f / sqrtf(var + eps);
  for (int i = lane; i < dim; i += 64) out[r * ldo + i] = (s[i] - mean) * inv * w[i] + b[i];
}

__global__ __launch_bounds__(256) void gather_rows_i64_kernel(const float *__restrict__ table, long long ldt,
                                                             const long long *__restrict__ idx, long long n,
                                                             int dim, float *__restrict__ out, long long ldo) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const float4 *s = reinterpret_cast<const float4 *>(table + idx[r] * ldt);
  float4 *d = reinterpret_cast<float4 *>(out + r * ldo);
  const int d4 = dim / 4;
  if (d4 <= 256) {   // rows of up to 1024 floats: every load before the first store (a load -> store loop is one round trip per piece)
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < d4) v[j] = s[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < d4) d[lane + 64 * j] = v[j];
    return;
  }
  for (int i = lane; i < d4; i += 64) d[i] = s[i];
}

// out[idx[r]] = src[r]: the inverse of gather_rows (padding-free token batches back into the padded layout)
__global__ __launch_bounds__(256) void scatter_rows_i64_kernel(const float *__restrict__ src, long long lds_,
                                                              const long long *__restrict__ idx, long long n,
                                                              int dim, float *__restrict__ out, long long ldo) {
  const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n) return;
  const int lane = threadIdx.x & 63;
  const float4 *s = reinterpret_cast<const float4 *>(src + r * lds_);
  float4 *d = reinterpret_cast<float4 *>(out + idx[r] * ldo);
  const int d4 = dim / 4;
  if (d4 <= 256) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < d4) v[j] = s[lane + 64 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (lane + 64 * j < d4) d[lane + 64 * j] = v[j];
    return;
  }
  for (int i = lane; i < d4; i += 64) d[i] = s[i];
}

__global__ __launch_bounds__(256) void scale_kernel(const float *__restrict__ x, float alpha, long long n,
                                                   float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = x[i] * alpha;
}

struct AttnArgs {
  const float *q, *k, *v;
  float *out;
  long long q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int nb, tq, tk, H, dh, kv_div;
  const float *bias;   // [H, bias_rows, bias_ld] or null; row = q_pos0 + t, col = key
  int bias_rows, bias_ld;
  int q_pos0;
  const long long *key_mask;  // [nb / kv_div, tk] (1 = attend) or null
  int causal;                 // key j allowed iff j <= q_pos0 + t
  float scale;                // multiplies q before the dot product (1 for T5)
  const long long *seq_off;   // packed sequences (attention_varlen_kernel): rows seq_off[b] .. seq_off[b+1]-1, else null
  const long long *kv_off;    // packed K|V only (cross-attention): keys of kv batch bk = rows kv_off[bk] .. kv_off[bk+1]-1
  const int *key_rows;        // attention_few_keys_kernel only: key j of row b lives in K|V batch row key_rows[b * tk + j] (else b / kv_div)
  // Output as the split image the o-projection reads (gemm_split.hip) instead of f32: o_bs / o_ts are then in HALVES of
  // the image [rows, 2 * onp], element (row, c) = hi at c, lo at onp + c, both scaled by 2^oexp -- ONE exponent for all rows,
  // from a bound on |V| the host derives from the layer's constants (ops.attention*: split_bound), because a row's heads
  // are written by different waves.  Costs range, not precision (see SplitOut in gemm_split.hip).
  _Float16 *oimg;
  int onp, oexp;
};

// context element at `off` = row offset + h * dh + column.  Called by whole lane PAIRS (2i, 2i + 1) holding consecutive
// columns (every call site: column = lane or lane + 64 under a bound that is a multiple of 4): for the image the pair
// exchanges its halves so that each lane issues ONE 4-byte store -- the even lane both hi, the odd lane both lo -- instead
// of two 2-byte ones (the 2-byte form cost the cross-attention kernel 15 %).
__device__ __forceinline__ void put_ctx(const AttnArgs &a, size_t off, float v) {
  if (a.oimg) {
    const float xs = ldexpf(v, a.oexp);
    const _Float16 hi = (_Float16)xs;
    const _Float16 lo = (_Float16)(xs - (float)hi);
    const unsigned int mine = (unsigned int)__builtin_bit_cast(unsigned short, hi) |
                              ((unsigned int)__builtin_bit_cast(unsigned short, lo) << 16);
    const unsigned int other = (unsigned int)__shfl_xor((int)mine, 1);
    const bool odd = threadIdx.x & 1;
    // even lane: (hi_even, hi_odd) at hi[off]; odd lane: (lo_even, lo_odd) at lo[off - 1]
    const unsigned int word = odd ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
    *reinterpret_cast<unsigned int *>(a.oimg + (odd ? off - 1 + a.onp : off)) = word;
  } else {
    a.out[off] = v;
  }
}
__device__ __forceinline__ void put_ctx4(const AttnArgs &a, size_t off, float4 v) {
  if (a.oimg) {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    const float x[4] = {ldexpf(v.x, a.oexp), ldexpf(v.y, a.oexp), ldexpf(v.z, a.oexp), ldexpf(v.w, a.oexp)};
    half4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hi[i] = (_Float16)x[i];
      lo[i] = (_Float16)(x[i] - (float)hi[i]);
    }
    *reinterpret_cast<half4 *>(a.oimg + off) = hi;
    *reinterpret_cast<half4 *>(a.oimg + off + a.onp) = lo;
  } else {
    *reinterpret_cast<float4 *>(a.out + off) = v;
  }
}
// A whole 64-wide head row of the context from LDS (`row`: 64 floats, written by this wave): f32 -- lane = column -- or the
// image with lanes 0-31 storing the hi pairs of columns (2l, 2l+1) and lanes 32-63 the lo pairs: each half-wave covers one
// contiguous 128-byte line (put_ctx's even / odd lanes interleave two lines inside every lane quad).
__device__ __forceinline__ void put_ctx_row64(const AttnArgs &a, size_t off, const float *row) {
  const int lane = threadIdx.x & 63;
  if (a.oimg) {
    const int c = 2 * (lane & 31);
    const float x0 = ldexpf(row[c], a.oexp), x1 = ldexpf(row[c + 1], a.oexp);
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    const bool lo = lane >= 32;
    const _Float16 w0 = lo ? (_Float16)(x0 - (float)h0) : h0, w1 = lo ? (_Float16)(x1 - (float)h1) : h1;
    const unsigned int word = (unsigned int)__builtin_bit_cast(unsigned short, w0) |
                              ((unsigned int)__builtin_bit_cast(unsigned short, w1) << 16);
    *reinterpret_cast<unsigned int *>(a.oimg + off + c + (lo ? a.onp : 0)) = word;
  } else {
    a.out[off + lane] = row[lane];
  }
}

constexpr int ATT_KPL = 4;  // keys per lane: key j lives on lane j & 63, slot j >> 6 (tk <= 256)

// softmax over the wave's keys (ATT_KPL per lane): s -> p in place
__device__ __forceinline__ void attn_softmax(float (&s)[ATT_KPL], int lane, int tk) {
  float m = s[0];
#pragma unroll
  for (int i = 1; i < ATT_KPL; ++i) m = fmaxf(m, s[i]);
  m = wave_max(m);
  float e[ATT_KPL], sum = 0.f;
#pragma unroll
  for (int i = 0; i < ATT_KPL; ++i) {
    e[i] = (64 * i + lane < tk) ? expf(s[i] - m) : 0.f;
    sum += e[i];
  }
  // the reference sums the row left to right; a wave reduction is a different association of the same
  // non-negative terms (covered by the 5e-5 tolerance of the T5 parity tests, as for tk <= 64)
  sum = wave_sum(sum);
#pragma unroll
  for (int i = 0; i < ATT_KPL; ++i) s[i] = e[i] / sum;
}

// one wave per (batch, head, query token); lane l owns keys l, l + 64, ... (tk <= 256)
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)a.nb * a.H * a.tq;
  if (wid >= total) return;
  const int lane = threadIdx.x & 63;
  const int t = (int)(wid % a.tq);
  const int h = (int)((wid / a.tq) % a.H);
  const int b = (int)(wid / ((long long)a.tq * a.H));
  const int bk = b / a.kv_div;
  // packed K|V (kv_off): every key is real and there are only tk = the sequence's length of them; the masked keys of
  // the padded form contribute exact zeros, so both forms give the same bits
  const long long kr0 = a.kv_off ? a.kv_off[bk] : 0;
  const int tk = a.kv_off ? (int)(a.kv_off[bk + 1] - kr0) : a.tk;
  const float *kbase = a.k + (a.kv_off ? (size_t)kr0 * a.k_ts : (size_t)bk * a.k_bs) + (size_t)h * a.dh;
  const float *q = a.q + (size_t)b * a.q_bs + (size_t)t * a.q_ts + (size_t)h * a.dh;
  const int qpos = a.q_pos0 + t;

  float s[ATT_KPL];
#pragma unroll
  for (int i = 0; i < ATT_KPL; ++i) {
    const int key = 64 * i + lane;
    s[i] = -INFINITY;
    if (key < tk) {
      const float *kr = kbase + (size_t)key * a.k_ts;
      float acc = 0.f;
      for (int d = 0; d < a.dh; d += 4) {
        const float4 kv = *reinterpret_cast<const float4 *>(kr + d);
        const float4 qv = *reinterpret_cast<const float4 *>(q + d);
        acc = fmaf(qv.x * a.scale, kv.x, acc);
        acc = fmaf(qv.y * a.scale, kv.y, acc);
        acc = fmaf(qv.z * a.scale, kv.z, acc);
        acc = fmaf(qv.w * a.scale, kv.w, acc);
      }
      float add = 0.f;
      if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + key];
      if (a.key_mask && !a.kv_off && a.key_mask[(size_t)bk * a.tk + key] == 0) add += -1e9f;
      if (a.causal && key > qpos) add += -1e9f;
      s[i] = acc + add;
    }
  }
  attn_softmax(s, lane, tk);

  const size_t o_off = (size_t)b * a.o_bs + (size_t)t * a.o_ts + (size_t)h * a.dh;
  const float *vb = a.v + (a.kv_off ? (size_t)kr0 * a.v_ts : (size_t)bk * a.v_bs) + (size_t)h * a.dh;
  // every lane takes part in the shuffles (a lane outside dh must still SOURCE p for its key)
  float acc0 = 0.f, acc1 = 0.f;  // output dims lane and lane + 64 (dh <= 128)
#pragma unroll
  for (int i = 0; i < ATT_KPL; ++i) {
    const int jend = tk - 64 * i < 64 ? tk - 64 * i : 64;
    for (int j = 0; j < jend; ++j) {
      const float pj = __shfl(s[i], j);
      const size_t row = (size_t)(64 * i + j) * a.v_ts;
      if (lane < a.dh) acc0 = fmaf(pj, vb[row + lane], acc0);
      if (lane + 64 < a.dh) acc1 = fmaf(pj, vb[row + lane + 64], acc1);
    }
  }
  if (lane < a.dh) put_ctx(a, o_off + lane, acc0);
  if (lane + 64 < a.dh) put_ctx(a, o_off + lane + 64, acc1);
}

// Self-attention of one (batch, head) per workgroup: Q, K, V tiles [tk x dh] staged ONCE in LDS
// (the wave-per-query kernel above re-reads K and V from L2 for every query row: 32x the traffic
// at S = 32).  K rows are padded by one float so that lane = key reads are bank-conflict free.
// Same arithmetic as attention_kernel (fmaf chain over d, expf softmax, fmaf chain over keys).
__global__ __launch_bounds__(256) void attention_tile_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int dh = a.dh, tk = a.tk, ldk = dh + 1;
  float *sq = sm;                    // [tk][dh]
  float *sk = sq + tk * dh;          // [tk][dh + 1]
  float *sv = sk + tk * ldk;         // [tk][dh]
  const float *qg = a.q + (size_t)b * a.q_bs + (size_t)h * dh;
  const float *kg = a.k + (size_t)b * a.k_bs + (size_t)h * dh;
  const float *vg = a.v + (size_t)b * a.v_bs + (size_t)h * dh;
  for (int i = t; i < tk * dh; i += 256) {
    const int r = i / dh, d = i - r * dh;
    sq[i] = qg[(size_t)r * a.q_ts + d] * a.scale;
    sk[r * ldk + d] = kg[(size_t)r * a.k_ts + d];
    sv[i] = vg[(size_t)r * a.v_ts + d];
  }
  __syncthreads();
  for (int tq = wave; tq < a.tq; tq += 4) {
    const int qpos = a.q_pos0 + tq;
    float s[ATT_KPL];
    const float *qr = sq + tq * dh;
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int key = 64 * i + lane;
      s[i] = -INFINITY;
      if (key < tk) {
        float acc = 0.f;
        const float *kr = sk + key * ldk;
        for (int d = 0; d < dh; ++d) acc = fmaf(qr[d], kr[d], acc);
        float add = 0.f;
        if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + key];
        if (a.key_mask && a.key_mask[(size_t)b * tk + key] == 0) add += -1e9f;
        if (a.causal && key > qpos) add += -1e9f;
        s[i] = acc + add;
      }
    }
    attn_softmax(s, lane, tk);
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int jend = tk - 64 * i < 64 ? tk - 64 * i : 64;
      for (int j = 0; j < jend; ++j) {
        const float pj = __shfl(s[i], j);
        if (lane < dh) acc0 = fmaf(pj, sv[(64 * i + j) * dh + lane], acc0);
        if (lane + 64 < dh) acc1 = fmaf(pj, sv[(64 * i + j) * dh + lane + 64], acc1);
      }
    }
    const size_t o_off = (size_t)b * a.o_bs + (size_t)tq * a.o_ts + (size_t)h * dh;
    if (lane < dh) put_ctx(a, o_off + lane, acc0);
    if (lane + 64 < dh) put_ctx(a, o_off + lane + 64, acc1);
  }
}

// Self-attention over PACKED sequences (the padding-free encoders): sequence b owns rows seq_off[b] .. seq_off[b+1]-1
// of q / k / v / out (row strides q_ts ...), every key is real, positions count from 0 inside the sequence.  One wave
// per (sequence, head) -- queries are ~10 tokens long, a workgroup per pair would idle most of its lanes -- with its own
// LDS region for Q, K, V.  The arithmetic is attention_tile_kernel's, statement for statement (k-ordered fmaf dots,
// attn_softmax over lanes = keys, key-ordered fmaf for P.V): on the padded layout the masked keys contribute exact
// zeros, so both kernels produce the same bits for the real rows.
__global__ __launch_bounds__(256) void attention_varlen_kernel(AttnArgs a, int max_len, int waves) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= waves) return;
  const long long pair = (long long)blockIdx.x * waves + wave;
  if (pair >= (long long)a.nb * a.H) return;
  const int b = (int)(pair / a.H), h = (int)(pair % a.H);
  const long long r0 = a.seq_off[b];
  const int tk = (int)(a.seq_off[b + 1] - r0);
  const int dh = a.dh, ldk = dh + 1, dh4 = dh >> 2;
  float *sq = sm + (size_t)wave * max_len * (3 * dh + 1);   // [tk][dh]
  float *sk = sq + (size_t)max_len * dh;                      // [tk][dh + 1]
  float *sv = sk + (size_t)max_len * ldk;                     // [tk][dh]
  const float *qg = a.q + (size_t)r0 * a.q_ts + (size_t)h * dh;
  const float *kg = a.k + (size_t)r0 * a.k_ts + (size_t)h * dh;
  const float *vg = a.v + (size_t)r0 * a.v_ts + (size_t)h * dh;
  for (int i = lane; i < tk * dh4; i += 64) {
    const int r = i / dh4, d = (i - r * dh4) * 4;
    const float4 q4 = *reinterpret_cast<const float4 *>(qg + (size_t)r * a.q_ts + d);
    const float4 k4 = *reinterpret_cast<const float4 *>(kg + (size_t)r * a.k_ts + d);
    const float4 v4 = *reinterpret_cast<const float4 *>(vg + (size_t)r * a.v_ts + d);
    float *q_ = sq + r * dh + d, *k_ = sk + r * ldk + d, *v_ = sv + r * dh + d;
    q_[0] = q4.x * a.scale; q_[1] = q4.y * a.scale; q_[2] = q4.z * a.scale; q_[3] = q4.w * a.scale;
    k_[0] = k4.x; k_[1] = k4.y; k_[2] = k4.z; k_[3] = k4.w;
    v_[0] = v4.x; v_[1] = v4.y; v_[2] = v4.z; v_[3] = v4.w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int tq = 0; tq < tk; ++tq) {
    float s[ATT_KPL];
    const float *qr = sq + tq * dh;
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int key = 64 * i + lane;
      s[i] = -INFINITY;
      if (key < tk) {
        float acc = 0.f;
        const float *kr = sk + key * ldk;
        for (int d = 0; d < dh; ++d) acc = fmaf(qr[d], kr[d], acc);
        float add = 0.f;
        if (a.bias) add = a.bias[((size_t)h * a.bias_rows + tq) * a.bias_ld + key];
        if (a.causal && key > tq) add += -1e9f;
        s[i] = acc + add;
      }
    }
    attn_softmax(s, lane, tk);
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int jend = tk - 64 * i < 64 ? tk - 64 * i : 64;
      for (int j = 0; j < jend; ++j) {
        const float pj = __shfl(s[i], j);
        if (lane < dh) acc0 = fmaf(pj, sv[(64 * i + j) * dh + lane], acc0);
        if (lane + 64 < dh) acc1 = fmaf(pj, sv[(64 * i + j) * dh + lane + 64], acc1);
      }
    }
    const size_t o_off = (size_t)(r0 + tq) * a.o_ts + (size_t)h * dh;
    if (lane < dh) put_ctx(a, o_off + lane, acc0);
    if (lane + 64 < dh) put_ctx(a, o_off + lane + 64, acc1);
  }
}

// attention_varlen_kernel for sequences of <= 64 tokens and DH-wide heads: lane j keeps key j's row in registers, the
// query row is read from LDS as broadcast float4s, two query rows advance together (two independent fmaf chains), and
// the P.V loop takes p_j through v_readlane.  Same operations in the same order per output element as the generic
// kernel (the k-ordered chain, attn_softmax, the key-ordered chain), so the bits are the same.
template <int DH>
__global__ __launch_bounds__(256) void attention_varlen_short_kernel(AttnArgs a, int max_len) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long pair = (long long)blockIdx.x * 4 + wave;
  if (pair >= (long long)a.nb * a.H) return;
  const int b = (int)(pair / a.H), h = (int)(pair % a.H);
  const long long r0 = a.seq_off[b];
  const int tk = (int)(a.seq_off[b + 1] - r0);
  constexpr int D4 = DH / 4;
  float *sq = sm + (size_t)wave * max_len * 2 * DH;   // [tk][DH], pre-scaled
  float *sv = sq + (size_t)max_len * DH;                // [tk][DH]
  const float *qg = a.q + (size_t)r0 * a.q_ts + (size_t)h * DH;
  const float *kg = a.k + (size_t)r0 * a.k_ts + (size_t)h * DH;
  const float *vg = a.v + (size_t)r0 * a.v_ts + (size_t)h * DH;
  for (int i = lane; i < tk * D4; i += 64) {
    const int r = i / D4, d = (i - r * D4) * 4;
    float4 q4 = *reinterpret_cast<const float4 *>(qg + (size_t)r * a.q_ts + d);
    q4.x *= a.scale; q4.y *= a.scale; q4.z *= a.scale; q4.w *= a.scale;
    *reinterpret_cast<float4 *>(sq + r * DH + d) = q4;
    *reinterpret_cast<float4 *>(sv + r * DH + d) = *reinterpret_cast<const float4 *>(vg + (size_t)r * a.v_ts + d);
  }
  float kreg[DH];
  if (lane < tk) {
#pragma unroll
    for (int d4 = 0; d4 < D4; ++d4) {
      const float4 k4 = *reinterpret_cast<const float4 *>(kg + (size_t)lane * a.k_ts + d4 * 4);
      kreg[4 * d4] = k4.x; kreg[4 * d4 + 1] = k4.y; kreg[4 * d4 + 2] = k4.z; kreg[4 * d4 + 3] = k4.w;
    }
  } else {
#pragma unroll
    for (int d = 0; d < DH; ++d) kreg[d] = 0.f;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the bias values of a row pair are fetched one pair ahead by unconditional loads (lane clamped to a real key): a load
  // under `lane < tk` right before its use is a branch + a full memory wait in every iteration
  const int blane = lane < tk ? lane : tk - 1;
  const float *bias_h = a.bias ? a.bias + (size_t)h * a.bias_rows * a.bias_ld + blane : nullptr;
  float nb_a = 0.f, nb_b = 0.f;
  if (a.bias) {
    nb_a = bias_h[0];
    nb_b = bias_h[(size_t)(1 < tk ? 1 : 0) * a.bias_ld];
  }
  for (int tq = 0; tq < tk; tq += 2) {
    const int t1 = tq + 1 < tk ? tq + 1 : tq;         // odd length: the last pair repeats a row
    const float bias_a = nb_a, bias_b = nb_b;
    if (a.bias) {
      const int n0 = tq + 2 < tk ? tq + 2 : tk - 1, n1 = tq + 3 < tk ? tq + 3 : tk - 1;
      nb_a = bias_h[(size_t)n0 * a.bias_ld];
      nb_b = bias_h[(size_t)n1 * a.bias_ld];
    }
    const float4 *qa = reinterpret_cast<const float4 *>(sq + tq * DH);
    const float4 *qb = reinterpret_cast<const float4 *>(sq + t1 * DH);
    float acc_a = 0.f, acc_b = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < D4; ++d4) {
      const float4 x = qa[d4], y = qb[d4];
      acc_a = fmaf(x.x, kreg[4 * d4], acc_a);     acc_b = fmaf(y.x, kreg[4 * d4], acc_b);
      acc_a = fmaf(x.y, kreg[4 * d4 + 1], acc_a); acc_b = fmaf(y.y, kreg[4 * d4 + 1], acc_b);
      acc_a = fmaf(x.z, kreg[4 * d4 + 2], acc_a); acc_b = fmaf(y.z, kreg[4 * d4 + 2], acc_b);
      acc_a = fmaf(x.w, kreg[4 * d4 + 3], acc_a); acc_b = fmaf(y.w, kreg[4 * d4 + 3], acc_b);
    }
    float sa[ATT_KPL], sb[ATT_KPL];
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) sa[i] = sb[i] = -INFINITY;
    if (lane < tk) {
      float add_a = 0.f, add_b = 0.f;
      if (a.bias) add_a = bias_a, add_b = bias_b;
      if (a.causal && lane > tq) add_a += -1e9f;
      if (a.causal && lane > t1) add_b += -1e9f;
      sa[0] = acc_a + add_a;
      sb[0] = acc_b + add_b;
    }
    attn_softmax(sa, lane, tk);
    attn_softmax(sb, lane, tk);
    float oa = 0.f, ob = 0.f;
    for (int j = 0; j < tk; ++j) {
      const float pa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sa[0]), j));
      const float pb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sb[0]), j));
      const float vj = lane < DH ? sv[j * DH + lane] : 0.f;
      oa = fmaf(pa, vj, oa);
      ob = fmaf(pb, vj, ob);
    }
    // the rows' q are dead: their contexts wait there.  Storing them here put the next pair's bias loads BEHIND the stores
    // in the wave's in-order memory counter -- a write acknowledgement (microseconds) per row pair: the kernel ran at 560 us
    // per call with its loads alone taking 127 us and its arithmetic nothing measurable (profiles/r03_seq2seq_experiments.txt)
    if (lane < DH) {
      sq[tq * DH + lane] = oa;
      if (t1 != tq) sq[t1 * DH + lane] = ob;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  static_assert(DH == 64, "put_ctx_row64");
  for (int r = 0; r < tk; ++r) put_ctx_row64(a, (size_t)(r0 + r) * a.o_ts + (size_t)h * DH, sq + r * DH);
}

// Decode-step attention over a handful of cached keys (tq == 1, tk <= 8: the decoder's and the adaptor's
// self-attention): one wave takes EIGHT (row, head) pairs -- lane 8g + j scores key j of pair g, the softmax runs
// inside the 8-lane groups, then the 64 lanes are the output dims of one pair after the other.  The
// wave-per-pair kernel used 6 of 64 lanes and was dispatch bound (61 k waves, 73 us per call, 30 % of the NCI
// generate time).  Same arithmetic, bit for bit: the group butterfly (xor 4, 2, 1) is the association the 64-lane
// butterfly has on <= 8 non-zero lanes.
__global__ __launch_bounds__(256) void attention_few_keys_kernel(AttnArgs a) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long npair = (long long)a.nb * a.H;
  if (wid * 8 >= npair) return;
  const int lane = threadIdx.x & 63;
  const int g = lane >> 3, j = lane & 7;
  const int dh = a.dh, tk = a.tk, qpos = a.q_pos0;
  long long pair = wid * 8 + g;
  const bool pair_ok = pair < npair;
  if (!pair_ok) pair = npair - 1;
  const int h = (int)(pair % a.H);
  const int b = (int)(pair / a.H);
  float s = -INFINITY;
  if (j < tk) {
    const long long bk = a.key_rows ? a.key_rows[(size_t)b * tk + j] : b / a.kv_div;   // ancestor-indexed cache: no re-ordered copy
    const float *q = a.q + (size_t)b * a.q_bs + (size_t)h * dh;
    const float *kr = a.k + (size_t)bk * a.k_bs + (size_t)j * a.k_ts + (size_t)h * dh;
    float acc = 0.f;
    for (int d = 0; d < dh; d += 4) {
      const float4 kv = *reinterpret_cast<const float4 *>(kr + d);
      const float4 qv = *reinterpret_cast<const float4 *>(q + d);
      acc = fmaf(qv.x * a.scale, kv.x, acc);
      acc = fmaf(qv.y * a.scale, kv.y, acc);
      acc = fmaf(qv.z * a.scale, kv.z, acc);
      acc = fmaf(qv.w * a.scale, kv.w, acc);
    }
    float add = 0.f;
    if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + j];
    if (a.key_mask && a.key_mask[(size_t)(b / a.kv_div) * tk + j] == 0) add += -1e9f;
    if (a.causal && j > qpos) add += -1e9f;
    s = acc + add;
  }
  float m = s;
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  const float e = j < tk ? expf(s - m) : 0.f;
  float sum = e;
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float p = e / sum;
  // context: the wave's 64 lanes are the output dims of pair 0, then pair 1, ...
  for (int gg = 0; gg < 8; ++gg) {
    const long long pr = wid * 8 + gg;
    if (pr >= npair) break;  // wave-uniform
    const int hh = (int)(pr % a.H);
    const int bb = (int)(pr / a.H);
    float acc0 = 0.f, acc1 = 0.f;
    for (int jj = 0; jj < tk; ++jj) {
      const long long vr = a.key_rows ? a.key_rows[(size_t)bb * tk + jj] : bb / a.kv_div;
      const float *vb = a.v + (size_t)vr * a.v_bs + (size_t)jj * a.v_ts + (size_t)hh * dh;
      const float pj = __shfl(p, 8 * gg + jj);
      if (lane < dh) acc0 = fmaf(pj, vb[lane], acc0);
      if (lane + 64 < dh) acc1 = fmaf(pj, vb[lane + 64], acc1);
    }
    const size_t o_off = (size_t)bb * a.o_bs + (size_t)hh * dh;
    if (lane < dh) put_ctx(a, o_off + lane, acc0);
    if (lane + 64 < dh) put_ctx(a, o_off + lane + 64, acc1);
  }
}

// The same kernel for 64- / 96-wide heads with the key rows and q TRANSPOSED THROUGH LDS.  Above, lane (g, j) walks its own
// 256-byte key row with float4 loads: every load instruction of the wave touches 64 different cache lines (16 instructions
// x 64 lines per wave; the context phase and q together ~200), so the per-key cost is the L1's line rate, not a bandwidth,
// and the 64-bit pair / H, pair % H per pair (24 software divisions per wave) are most of the fixed cost
// (tools/bench_attn_cached.py at 69.8 k rows x 12 heads: 0.22 ms + 0.11 ms per cached position).  Here
//   * the wave fetches the key rows of its 8 TK live (pair, key) slots as four lanes x 16 bytes per row and quarter (16 rows
//     per instruction), half a row in flight at a time, passes them through a [64 rows][16 + 4] LDS slab (the padding makes
//     the b128 reads of 16 consecutive rows conflict-free) and every lane then runs the SAME sequential fmaf chain over its
//     row: identical bits;
//   * (row, head) of the eight pairs come from one 32-bit division and increments; the V row pointers of the context phase
//     are computed once by the score lanes and read back with readlane; TK is a template argument (no per-key branches).
template <int TK, int DH>   // DH = 64 (T5 heads) or 96 (the adaptor's eight heads over 768)
__global__ __launch_bounds__(256, 5) void attention_few_keys_staged_kernel(AttnArgs a) {
  constexpr int KS = 20, QS = DH + 4, NQH = DH / 32;   // NQH: 16-float pieces of a key row per half
  constexpr int NSLOT = 8 * TK, N_IT = (NSLOT + 15) / 16;
  __shared__ __attribute__((aligned(16))) float ks_all[4][64 * KS];
  __shared__ __attribute__((aligned(16))) float qs_all[4][8 * QS];
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const long long wid = (long long)blockIdx.x * 4 + w;
  const long long npair = (long long)a.nb * a.H;
  if (wid * 8 >= npair) return;
  float *ks = ks_all[w], *qs = qs_all[w];
  const int lane = threadIdx.x & 63;
  const int g = lane >> 3, j = lane & 7;
  const int qpos = a.q_pos0;
  const int pr0 = (int)(wid * 8);
  const int ngg = (int)((npair - wid * 8) < 8 ? (npair - wid * 8) : 8);   // live pairs of this wave (wave-uniform)
  int bb[8], hh[8];
  {
    int cb = pr0 / a.H, ch = pr0 - cb * a.H;
#pragma unroll
    for (int gg = 0; gg < 8; ++gg) {
      bb[gg] = gg < ngg ? cb : a.nb - 1;       // dead pairs repeat the last one (their results are not stored)
      hh[gg] = gg < ngg ? ch : a.H - 1;
      if (++ch == a.H) ch = 0, ++cb;
    }
  }
  int b = bb[0], h = hh[0];
#pragma unroll
  for (int gg = 1; gg < 8; ++gg)
    if (g == gg) b = bb[gg], h = hh[gg];
  const int jc = j < TK ? j : 0;
  int bk = 0;
  if (j < TK) bk = a.key_rows ? a.key_rows[(size_t)b * TK + j] : b / a.kv_div;
  const float *kr = a.k + (size_t)bk * a.k_bs + (size_t)jc * a.k_ts + (size_t)h * DH;
  const float *vr = a.v + (size_t)bk * a.v_bs + (size_t)jc * a.v_ts + (size_t)h * DH;
  const int vr_lo = (int)(unsigned long long)vr, vr_hi = (int)((unsigned long long)vr >> 32);
  // q rows of the wave's eight pairs, 256 contiguous bytes each
#pragma unroll
  for (int gg = 0; gg < 8; ++gg) {
    qs[gg * QS + lane] = a.q[(size_t)bb[gg] * a.q_bs + (size_t)hh[gg] * DH + lane];
    if (DH > 64 && lane + 64 < DH) qs[gg * QS + 64 + lane] = a.q[(size_t)bb[gg] * a.q_bs + (size_t)hh[gg] * DH + 64 + lane];
  }
  // slot s = 16 it + (lane >> 2) of the wave's 8 TK live (pair, key) slots -> score lane 8 (s / TK) + s % TK
  const int piece = lane & 3;
  int srow[N_IT];
  const float *sp[N_IT];
#pragma unroll
  for (int it = 0; it < N_IT; ++it) {
    const int sl = 16 * it + (lane >> 2);
    srow[it] = sl < NSLOT ? 8 * (sl / TK) + sl % TK : -1;
    const int from = srow[it] < 0 ? 0 : srow[it];
    const unsigned long long src = ((unsigned long long)(unsigned int)__shfl((int)((unsigned long long)kr >> 32), from) << 32) |
                                   (unsigned int)__shfl((int)(unsigned long long)kr, from);
    sp[it] = reinterpret_cast<const float *>(src) + 4 * piece;
  }
  // Loads in as few dependent rounds as the registers allow (the waves spend 71 % of their life parked on s_waitcnt,
  // profiles/r04_attention_pmc.txt): the bias / mask terms and the V rows of the first VB pairs go out with the key rows, both
  // halves of the key rows together at DH = 64, and every later group of V rows before the previous group's chains.
  float add = 0.f;
  if (j < TK) {
    if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + j];
    if (a.key_mask && a.key_mask[(size_t)(b / a.kv_div) * TK + j] == 0) add += -1e9f;
    if (a.causal && j > qpos) add += -1e9f;
  }
  constexpr int VB = TK > 5 ? 2 : 4;
  constexpr int NG = 8 / VB;
  constexpr bool SMALL = DH == 64 && TK <= 4;   // registers for everything at once (96 per lane at five waves per SIMD)
  constexpr int NVB = SMALL ? 2 : 1;            // groups of V rows in flight
  float vv[NVB][VB][TK], vw[NVB][VB][DH > 64 ? TK : 1];   // columns lane and (DH = 96) 64 + lane
  auto load_v = [&](int g0, int buf) {
#pragma unroll
    for (int gi = 0; gi < VB; ++gi)
#pragma unroll
      for (int jj = 0; jj < TK; ++jj) {   // a dead pair's pointer is the last live pair's: loaded, not stored
        const unsigned long long vp = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(vr_hi, 8 * (g0 + gi) + jj) << 32) |
                                      (unsigned int)__builtin_amdgcn_readlane(vr_lo, 8 * (g0 + gi) + jj);
        vv[buf][gi][jj] = reinterpret_cast<const float *>(vp)[lane];
        if constexpr (DH > 64) vw[buf][gi][jj] = lane + 64 < DH ? reinterpret_cast<const float *>(vp)[64 + lane] : 0.f;
      }
  };
  constexpr int NBK = SMALL ? 1 : 2;        // rounds of key-row loads: the whole row at once, or half a row
  constexpr int NQR = 2 * NQH / NBK;        // 16-float pieces per round
  float acc = 0.f;
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    float4 kf[NQR][N_IT];
#pragma unroll
    for (int it = 0; it < N_IT; ++it)
#pragma unroll
      for (int qi = 0; qi < NQR; ++qi)
        kf[qi][it] = srow[it] >= 0 ? *reinterpret_cast<const float4 *>(sp[it] + 16 * (NQR * nb + qi)) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (nb == 0) load_v(0, 0);
#pragma unroll
    for (int qi = 0; qi < NQR; ++qi) {
      const int qt = NQR * nb + qi;
#pragma unroll
      for (int it = 0; it < N_IT; ++it)
        if (srow[it] >= 0) *reinterpret_cast<float4 *>(&ks[srow[it] * KS + 4 * piece]) = kf[qi][it];
      __builtin_amdgcn_wave_barrier();
      if (j < TK) {
#pragma unroll
        for (int d = 0; d < 16; d += 4) {
          const float4 kv = *reinterpret_cast<const float4 *>(&ks[lane * KS + d]);
          const float4 qv = *reinterpret_cast<const float4 *>(&qs[g * QS + 16 * qt + d]);
          acc = fmaf(qv.x * a.scale, kv.x, acc);
          acc = fmaf(qv.y * a.scale, kv.y, acc);
          acc = fmaf(qv.z * a.scale, kv.z, acc);
          acc = fmaf(qv.w * a.scale, kv.w, acc);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  float s = -INFINITY;
  if (j < TK) s = acc + add;
  float m = s;
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  const float e = j < TK ? expf(s - m) : 0.f;
  float sum = e;
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float p = e / sum;
  // context: V rows of VB pairs at a time (VB TK rows of 256 B, one row per instruction; their pointers pass through SGPRs,
  // which is what limits VB), the next group's loads ahead of this group's chains
#pragma unroll
  for (int gr = 0; gr < NG; ++gr) {
    const int g0 = gr * VB;
    if (g0 >= ngg) break;
    if (NVB == 2 && gr + 1 < NG && g0 + VB < ngg) load_v(g0 + VB, (gr + 1) & 1);
    if (NVB == 1 && gr > 0) load_v(g0, 0);
    constexpr int one = NVB - 1;
#pragma unroll
    for (int gi = 0; gi < VB; ++gi) {
      const int gg = g0 + gi;
      if (gg >= ngg) break;
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int jj = 0; jj < TK; ++jj) {
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, p), 8 * gg + jj));
        acc0 = fmaf(pj, vv[gr & one][gi][jj], acc0);
        if constexpr (DH > 64) acc1 = fmaf(pj, vw[gr & one][gi][jj], acc1);
      }
      const size_t o_off = (size_t)bb[gg] * a.o_bs + (size_t)hh[gg] * DH;
      put_ctx(a, o_off + lane, acc0);
      if (DH > 64 && lane + 64 < DH) put_ctx(a, o_off + 64 + lane, acc1);
    }
  }
}

// Decode-step cross-attention where kv_div rows (the beams of one query) share K and V: one workgroup per
// (query, head) stages K [tk][dh + 4], V [tk][dh] and the group's q rows in LDS ONCE and its waves take the
// beams in turn.  The wave-per-row kernel re-read the shared K|V from L2 for every beam (10x the traffic).
// Same arithmetic as attention_kernel, lane for lane.
__global__ __launch_bounds__(256) void attention_group_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int bk = blockIdx.x / a.H, h = blockIdx.x % a.H;  // query group, head
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int dh = a.dh, ldk = dh + 4;  // 16-byte aligned rows; 4r mod 64 banks: float4 reads conflict free
  const long long kr0 = a.kv_off ? a.kv_off[bk] : 0;     // packed K|V: only the real keys exist (see attention_kernel)
  const int tk = a.kv_off ? (int)(a.kv_off[bk + 1] - kr0) : a.tk;
  float *sk = sm;                      // [tk][dh + 4]   (regions sized for a.tk = the longest sequence)
  float *sv = sk + a.tk * ldk;         // [tk][dh]
  float *sq = sv + a.tk * dh;          // [kv_div][dh], pre-scaled
  const float *kg = a.k + (a.kv_off ? (size_t)kr0 * a.k_ts : (size_t)bk * a.k_bs) + (size_t)h * dh;
  const float *vg = a.v + (a.kv_off ? (size_t)kr0 * a.v_ts : (size_t)bk * a.v_bs) + (size_t)h * dh;
  const int dq = dh / 4;
  for (int i = t; i < tk * dq; i += 256) {
    const int r = i / dq, c4 = (i - r * dq) * 4;
    *reinterpret_cast<float4 *>(sk + r * ldk + c4) = *reinterpret_cast<const float4 *>(kg + (size_t)r * a.k_ts + c4);
    *reinterpret_cast<float4 *>(sv + r * dh + c4) = *reinterpret_cast<const float4 *>(vg + (size_t)r * a.v_ts + c4);
  }
  for (int i = t; i < a.kv_div * dq; i += 256) {
    const int r = i / dq, c4 = (i - r * dq) * 4;
    const int b = bk * a.kv_div + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (b < a.nb) v = *reinterpret_cast<const float4 *>(a.q + (size_t)b * a.q_bs + (size_t)h * dh + c4);
    *reinterpret_cast<float4 *>(sq + r * dh + c4) = make_float4(v.x * a.scale, v.y * a.scale, v.z * a.scale, v.w * a.scale);
  }
  __syncthreads();
  const int qpos = a.q_pos0;  // tq == 1
  if (tk <= 32) {
    // Few keys (a query of <= 32 tokens): a wave takes 64 / G rows at once, G = 8 | 16 | 32 lanes per row, lane = (row, key).
    // One row per wave pass left 64 - tk lanes idle in the score loop, and that loop's LDS reads -- 32 x 1 KiB per row --
    // are what the kernel was bound by.  Same arithmetic bit for bit: the in-group butterflies (xor G/2 .. 1) are the
    // association attn_softmax's 64-lane butterflies have when lanes >= G hold zeros (as attention_few_keys_kernel).
    const int G = tk <= 8 ? 8 : (tk <= 16 ? 16 : 32), rpw = 64 / G;
    const int g = lane / G, j = lane & (G - 1);
    for (int r0 = wave * rpw; r0 < a.kv_div; r0 += 4 * rpw) {
      const int r = r0 + g;
      const bool ok = r < a.kv_div && bk * a.kv_div + r < a.nb && j < tk;
      float sj = -INFINITY;
      if (ok) {
        const float *q = sq + r * dh;
        const float *kr = sk + j * ldk;
        float acc = 0.f;
        for (int d = 0; d < dh; d += 4) {
          const float4 kv = *reinterpret_cast<const float4 *>(kr + d);
          const float4 qv = *reinterpret_cast<const float4 *>(q + d);
          acc = fmaf(qv.x, kv.x, acc);
          acc = fmaf(qv.y, kv.y, acc);
          acc = fmaf(qv.z, kv.z, acc);
          acc = fmaf(qv.w, kv.w, acc);
        }
        float add = 0.f;
        if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + j];
        if (a.key_mask && !a.kv_off && a.key_mask[(size_t)bk * tk + j] == 0) add += -1e9f;
        if (a.causal && j > qpos) add += -1e9f;
        sj = acc + add;
      }
      float m = sj;
      for (int off = G >> 1; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      const float e = ok ? expf(sj - m) : 0.f;
      float sum = e;
      for (int off = G >> 1; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      const float pr = e / sum;
      for (int gg = 0; gg < rpw; ++gg) {           // the wave's 64 lanes are the output dims of one row after the other
        const int rr = r0 + gg, b = bk * a.kv_div + rr;
        if (rr >= a.kv_div || b >= a.nb) break;    // wave-uniform
        float acc0 = 0.f, acc1 = 0.f;
        for (int jj = 0; jj < tk; ++jj) {
          const float pj = __shfl(pr, gg * G + jj);
          if (lane < dh) acc0 = fmaf(pj, sv[jj * dh + lane], acc0);
          if (lane + 64 < dh) acc1 = fmaf(pj, sv[jj * dh + lane + 64], acc1);
        }
        const size_t o_off = (size_t)b * a.o_bs + (size_t)h * dh;
        if (lane < dh) put_ctx(a, o_off + lane, acc0);
        if (lane + 64 < dh) put_ctx(a, o_off + lane + 64, acc1);
      }
    }
    return;
  }
  for (int r = wave; r < a.kv_div; r += 4) {
    const int b = bk * a.kv_div + r;
    if (b >= a.nb) break;
    const float *q = sq + r * dh;
    float s[ATT_KPL];
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int key = 64 * i + lane;
      s[i] = -INFINITY;
      if (key < tk) {
        const float *kr = sk + key * ldk;
        float acc = 0.f;
        for (int d = 0; d < dh; d += 4) {
          const float4 kv = *reinterpret_cast<const float4 *>(kr + d);
          const float4 qv = *reinterpret_cast<const float4 *>(q + d);
          acc = fmaf(qv.x, kv.x, acc);
          acc = fmaf(qv.y, kv.y, acc);
          acc = fmaf(qv.z, kv.z, acc);
          acc = fmaf(qv.w, kv.w, acc);
        }
        float add = 0.f;
        if (a.bias) add = a.bias[((size_t)h * a.bias_rows + qpos) * a.bias_ld + key];
        if (a.key_mask && !a.kv_off && a.key_mask[(size_t)bk * tk + key] == 0) add += -1e9f;
        if (a.causal && key > qpos) add += -1e9f;
        s[i] = acc + add;
      }
    }
    attn_softmax(s, lane, tk);
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int i = 0; i < ATT_KPL; ++i) {
      const int jend = tk - 64 * i < 64 ? tk - 64 * i : 64;
      for (int j = 0; j < jend; ++j) {
        const float pj = __shfl(s[i], j);
        if (lane < dh) acc0 = fmaf(pj, sv[(64 * i + j) * dh + lane], acc0);
        if (lane + 64 < dh) acc1 = fmaf(pj, sv[(64 * i + j) * dh + lane + 64], acc1);
      }
    }
    const size_t o_off = (size_t)b * a.o_bs + (size_t)h * dh;
    if (lane < dh) put_ctx(a, o_off + lane, acc0);
    if (lane + 64 < dh) put_ctx(a, o_off + lane + 64, acc1);
  }
}

// Whole-sequence self-attention on the f32 matrix cores, for 64 < tk = tq <= 128 and 64-wide heads (the
// 128-token passages of gen_doc_embedding): one (batch, head) per workgroup, wave w owns query rows
// 32w .. 32w+31.  Scores and context run as v_mfma_f32_32x32x2_f32 chains (exact f32 products, the scalar kernels
// above spent 44 % of the passage tower's time at 2.8 TFLOP/s).
//
// Both products are computed TRANSPOSED so that the probabilities never leave the registers:
//   S^T = K . Q^T   (A = K from LDS, B = the wave's Q rows from registers): a lane holds ONE query (lane & 31) and the
//                    keys 32n + (r&3) + 8(r>>2) + 4*half of it -- half of the row, the other half in lane ^ 32;
//   softmax         in registers: 64 values per lane, one cross-half exchange for the maximum, one for the sum;
//   O^T = V^T . P^T (A = V^T from LDS, B = P^T): the B operand wants, per step, one key from the lanes of half 0 and one
//                    from half 1 -- any pairing the A operand follows, so the step (n, r) takes exactly the two keys
//                    the lanes already hold in sc[n][r] (keys k0 and k0 + 4) and reads V rows k0 | k0 + 4 for them.
// The first version wrote the scores to LDS ([128][129] over Q|K), ran the softmax there and read P back as the A operand:
// 99 KiB of LDS = one workgroup per CU, three passes over LDS per probability, 795 us per call for 512 passages x 12 heads
// (32 TFLOP/s).  Now LDS holds K and V, [128][64] each (64 KiB: two workgroups per CU), Q passes through the V region on its
// way to the registers (row-contiguous global loads; fragments straight from global memory are 32 rows x 16 B per
// instruction and took half the kernel), the output goes through the K region for row-contiguous stores.
// Summation order per output (fixed, so a passage has the same bits in any batch and in the packed / padded layouts):
// scores over d in pairs (2j, 2j+1); row sum over the lane's registers in (n, r) order, then the other half; context over
// the key pairs (k0, k0 + 4) in (n, r) order.
constexpr int AM_S = 128, AM_D = 64;
constexpr size_t AM_LDS = (size_t)(2 * AM_S * AM_D + AM_S) * sizeof(float);
typedef float am_f32x16 __attribute__((ext_vector_type(16)));
// K, the Q rows on their way to the registers and the output staging are [rows][64] floats with the 16-byte chunk index
// XORed by the row (row & 15): the 16 lanes a ds_read_b128 serves per cycle read one chunk column of 16 consecutive rows
// from 16 different chunks = all 64 banks
__device__ __forceinline__ int am_sw4(int row, int c4) { return row * AM_D + 4 * (c4 ^ (row & 15)); }

// NB = number of 32-key blocks that hold real keys (compile-time: a block test per MFMA put a branch and a full LDS wait
// in front of every one of them)
template <int NB>
__device__ __forceinline__ void am_wave(const AttnArgs &a, const float *sk, const float *sv, const float *smask, float *so,
                                        const float (&qf)[AM_D / 2], int w, int h, int tk, int tq, size_t og) {
  const int lane = threadIdx.x & 63, lrow = lane & 31, half = lane >> 5;
  const int qi = 32 * w + lrow;
  am_f32x16 sc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[n][r] = 0.f;
  // S^T = K . Q^T: the lanes of half 0 / 1 supply head dims [0, 32) / [32, 64), step j = dims (j, 32 + j)
  const bool active = 32 * w < tq;    // wave-uniform
#pragma unroll
  for (int j4 = 0; j4 < (active ? AM_D / 8 : 0); ++j4) {
    float4 kf[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) kf[n] = *reinterpret_cast<const float4 *>(sk + am_sw4(32 * n + lrow, 8 * half + j4));
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      sc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n].x, qf[4 * j4], sc[n], 0, 0, 0);
      sc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n].y, qf[4 * j4 + 1], sc[n], 0, 0, 0);
      sc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n].z, qf[4 * j4 + 2], sc[n], 0, 0, 0);
      sc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[n].w, qf[4 * j4 + 3], sc[n], 0, 0, 0);
    }
  }
  __syncthreads();       // every wave is done with K: the output staging may overwrite it
  if (!active) return;   // no workgroup barrier below
  // scores + bias + masks (C/D map: col = lane & 31 = query, row = (r&3) + 8*(r>>2) + 4*half = key inside block n).
  // The bias values are fetched by unconditional loads (indices clamped), all of them in flight together: a load under a
  // per-element condition is a branch + a full wait each -- 64 dependent round trips per lane.
  const int qpos = a.q_pos0 + qi;
  float badd[NB][16];
  if (a.bias) {
    const int qrow = qpos < a.bias_rows ? qpos : a.bias_rows - 1;
    const float *brow = a.bias + ((size_t)h * a.bias_rows + qrow) * a.bias_ld;
    if ((a.bias_ld & 3) == 0 && a.bias_ld >= 32 * NB && ((uintptr_t)a.bias & 15) == 0) {   // 16-byte loads stay inside the row
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 x = *reinterpret_cast<const float4 *>(brow + 32 * n + 8 * g + 4 * half);
          badd[n][4 * g] = x.x; badd[n][4 * g + 1] = x.y; badd[n][4 * g + 2] = x.z; badd[n][4 * g + 3] = x.w;
        }
    } else {
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * n + (r & 3) + 8 * (r >> 2) + 4 * half;
          badd[n][r] = brow[key < a.bias_ld ? key : a.bias_ld - 1];
        }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) badd[n][r] = 0.f;
  }
  float m = -INFINITY;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 mk = *reinterpret_cast<const float4 *>(smask + 32 * n + 8 * g + 4 * half);  // 0 | -1e9 (masked key, padded layout)
      const float mk4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e, key = 32 * n + 8 * g + 4 * half + e;
        float add = mk4[e];
        if (a.bias) add += badd[n][r];
        add += (a.causal && key > qpos) ? -1e9f : 0.f;
        const float v = key < tk ? sc[n][r] + add : -INFINITY;
        sc[n][r] = v;
        m = fmaxf(m, v);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = expf(sc[n][r] - m);
      sc[n][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  // O^T = V^T . P^T: step (n, r) = keys (k0, k0 + 4), k0 = 32n + (r&3) + 8(r>>2) -- the two keys the lanes hold in sc[n][r]
  am_f32x16 o[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
  const float *va = sv + 4 * half * AM_D + lrow;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float vf[4][2];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c) vf[e][c] = va[(32 * n + 8 * g + e) * AM_D + 32 * c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = sc[n][4 * g + e] / sum;
#pragma unroll
        for (int c = 0; c < 2; ++c) o[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[e][c], p, o[c], 0, 0, 0);
      }
    }
  // O^T: col = lane & 31 = query, row = head dim -> through the wave's own rows of the K region, then whole rows out
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(so + am_sw4(lrow, 8 * c + 2 * g + half)) =
          make_float4(o[c][4 * g], o[c][4 * g + 1], o[c][4 * g + 2], o[c][4 * g + 3]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int nrow = tq - 32 * w < 32 ? tq - 32 * w : 32;
  const int sub = lane >> 4, c4 = lane & 15;   // four rows per pass, 16 lanes x 16 B each
  for (int r4 = 0; r4 < nrow; r4 += 4) {
    const int rr = r4 + sub;
    if (rr < nrow)
      put_ctx4(a, og + (size_t)(32 * w + rr) * a.o_ts + 4 * c4, *reinterpret_cast<const float4 *>(so + am_sw4(rr, c4)));
  }
}

__global__ __launch_bounds__(256, 2) void attention_mfma_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float *sk = sm;                     // [128][64], chunk-swizzled
  float *sv = sk + AM_S * AM_D;       // [128][64]; holds Q (chunk-swizzled) until the waves have their fragments
  float *smask = sv + AM_S * AM_D;    // [128] additive key mask
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  // padded layout: sequence b = rows [b][0 .. tk) with a key mask; packed layout (seq_off): rows seq_off[b] ..
  // seq_off[b+1]-1, every key real -- only the blocks that hold real rows / keys are computed then, and since the
  // masked keys of the padded form contribute exact zeros, both give the same bits
  const long long r0 = a.seq_off ? a.seq_off[b] : 0;
  const int tk = a.seq_off ? (int)(a.seq_off[b + 1] - r0) : a.tk;  // == tq
  const int tq = tk;
  const long long *key_mask = a.seq_off ? nullptr : a.key_mask;
  const float *qg = a.q + (a.seq_off ? (size_t)r0 * a.q_ts : (size_t)b * a.q_bs) + (size_t)h * AM_D;
  const float *kg = a.k + (a.seq_off ? (size_t)r0 * a.k_ts : (size_t)b * a.k_bs) + (size_t)h * AM_D;
  const float *vg = a.v + (a.seq_off ? (size_t)r0 * a.v_ts : (size_t)b * a.v_bs) + (size_t)h * AM_D;
  const size_t og = (a.seq_off ? (size_t)r0 * a.o_ts : (size_t)b * a.o_bs) + (size_t)h * AM_D;
  if (t < AM_S) smask[t] = (key_mask && t < tk && key_mask[(size_t)b * tk + t] == 0) ? -1e9f : 0.f;
  // all three operands with row-contiguous loads (16 lanes per 256-byte row); V waits in registers while Q uses its region
  float4 v4[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = t + 256 * it, r = i >> 4, c4 = i & 15;
    float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), k4 = q4;
    v4[it] = q4;
    if (r < tk) {  // rows past tk are zero
      q4 = *reinterpret_cast<const float4 *>(qg + (size_t)r * a.q_ts + 4 * c4);
      k4 = *reinterpret_cast<const float4 *>(kg + (size_t)r * a.k_ts + 4 * c4);
      v4[it] = *reinterpret_cast<const float4 *>(vg + (size_t)r * a.v_ts + 4 * c4);
    }
    *reinterpret_cast<float4 *>(sk + am_sw4(r, c4)) = k4;
    *reinterpret_cast<float4 *>(sv + am_sw4(r, c4)) = make_float4(q4.x * a.scale, q4.y * a.scale, q4.z * a.scale, q4.w * a.scale);
  }
  __syncthreads();
  float qf[AM_D / 2];                 // the lane's Q fragments: qf[j] = scale * Q[qi][32 half + j]
  {
    const int qi = 32 * w + lrow;
#pragma unroll
    for (int j4 = 0; j4 < AM_D / 8; ++j4) {
      const float4 x = *reinterpret_cast<const float4 *>(sv + am_sw4(qi, 8 * half + j4));
      qf[4 * j4] = x.x; qf[4 * j4 + 1] = x.y; qf[4 * j4 + 2] = x.z; qf[4 * j4 + 3] = x.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = t + 256 * it, r = i >> 4, c4 = i & 15;
    *reinterpret_cast<float4 *>(sv + r * AM_D + 4 * c4) = v4[it];
  }
  __syncthreads();
  float *so = sm + w * 32 * AM_D;
  switch ((tk + 31) / 32) {
    case 1: am_wave<1>(a, sk, sv, smask, so, qf, w, h, tk, tq, og); break;
    case 2: am_wave<2>(a, sk, sv, smask, so, qf, w, h, tk, tq, og); break;
    case 3: am_wave<3>(a, sk, sv, smask, so, qf, w, h, tk, tq, og); break;
    default: am_wave<4>(a, sk, sv, smask, so, qf, w, h, tk, tq, og); break;
  }
}

// ---- the same attention on the f16 matrix cores in split precision (the default wherever the split GEMM runs) -----------------
// attention_mfma_kernel is bound by the f32 matrix rate (64 cycles per v_mfma_f32_32x32x2f32: 1.14 ms per call for 2048
// passages x 12 heads, 12 % of the passage tower).  Here Q, K, V and P are (hi, lo) f16 pairs -- hi = f16(x 2^e), lo = f16(x 2^e
// - hi), 22 significant bits, as the linear layers' operands (gemm_split.hip) -- and a product is three v_mfma_f32_32x32x16_f16
// (lo.hi + hi.hi + hi.lo, f32 accumulate): 16 k per 32-cycle instruction instead of 2 k per 64-cycle one.
//   scaling      one power of two per operand and (sequence, head): max |x| 2^e in [2^13, 2^14) (exact; no f16 overflow whatever
//                the checkpoint), P by 2^14; undone on the f32 accumulators (ldexp, exact);
//   S^T = K.Q^T  A = K fragments from LDS ([key][64] halves, hi and lo planes, 16-byte chunks XORed by (key >> 1) & 7: conflict
//                free), B = the wave's Q rows in registers; four k-steps of 16 head dims x three products per key block;
//   softmax      as attention_mfma_kernel (same registers: C/D map col = lane & 31 = query, row = key);
//   O^T = V^T.P^T  the B operand of k-step (n, u) -- keys 32 n + 16 u + [0, 16) -- is taken from the registers the lane already
//                holds: slot j of half h = register 8 u + j = key 32 n + 16 u + 8 (j >> 2) + 4 h + (j & 3); V is staged TRANSPOSED,
//                [dim][key slots in that order] halves (chunks XORed by dim & 15), so the A fragment of a lane (dim, h) is one
//                16-byte read.
// LDS: K planes 32 KiB (re-used for the output rows), V^T planes 32 KiB (Q passes through them first), key mask: 64.5 KiB, two
// workgroups per CU, as before.  Masked / absent keys contribute exact zeros, so packed and padded layouts agree bit for bit;
// the summation order per output is fixed (dims in k-steps of 16, keys in k-steps of 16, per step lo.hi, hi.hi, hi.lo).
// Accuracy: operand images 2^-22 relative, products exact in f32 -- the f32 kernel's own accumulation error is the same order;
// MEVI_ATTN_PASSAGE=f32 keeps attention_mfma_kernel (A/B switch).
typedef _Float16 ah_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ah_f16x4 __attribute__((ext_vector_type(4)));
constexpr size_t AH_LDS = (size_t)(4 * AM_S * AM_D) * sizeof(_Float16) + (AM_S + 16) * sizeof(float);
// byte offset of the 8-byte piece (dims 4 c4 .. 4 c4 + 3) of row r in a [rows][64] halves plane
__device__ __forceinline__ int ah_row8(int r, int c4) { return r * 128 + (((c4 >> 1) ^ ((r >> 1) & 7)) << 4) + ((c4 & 1) << 3); }
// byte offset of the 16-byte fragment piece (dims 8 c8 .. 8 c8 + 7) of row r
__device__ __forceinline__ int ah_row16(int r, int c8) { return r * 128 + ((c8 ^ ((r >> 1) & 7)) << 4); }
// byte offset of chunk c (16 bytes = 8 key slots) of dim row d in a [64][128] halves plane
__device__ __forceinline__ int ah_vt16(int d, int c) { return d * 256 + ((c ^ (d & 15)) << 4); }

__device__ __forceinline__ int ah_exp_for(float m) {   // e with m 2^e in [2^13, 2^14); 0 for m = 0 / inf / nan
  if (!(m > 0.f) || isinf(m)) return 0;
  int e;
  (void)frexpf(m, &e);
  const int s_ = 14 - e;
  return s_ > 100 ? 100 : (s_ < -100 ? -100 : s_);
}

template <int NB>
__device__ __forceinline__ void ah_wave(const AttnArgs &a, const char *kh, const char *kl, const char *vth, const char *vtl,
                                        const float *smask, float *so, const ah_f16x8 (&qh)[4], const ah_f16x8 (&ql)[4],
                                        int e_s, int e_o, int w, int h, int tk, int tq, size_t og) {
  const int lane = threadIdx.x & 63, lrow = lane & 31, half = lane >> 5;
  const int qi = 32 * w + lrow;
  am_f32x16 sc[NB];
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[n][r] = 0.f;
  const bool active = 32 * w < tq;    // wave-uniform
  if (active) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      ah_f16x8 fh[NB], fl[NB];
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        fh[n] = *reinterpret_cast<const ah_f16x8 *>(kh + ah_row16(32 * n + lrow, 2 * st + half));
        fl[n] = *reinterpret_cast<const ah_f16x8 *>(kl + ah_row16(32 * n + lrow, 2 * st + half));
      }
#pragma unroll
      for (int n = 0; n < NB; ++n) {
        sc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl[n], qh[st], sc[n], 0, 0, 0);
        sc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[n], qh[st], sc[n], 0, 0, 0);
        sc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh[n], ql[st], sc[n], 0, 0, 0);
      }
    }
  }
  __syncthreads();       // every wave is done with K: the output staging may overwrite it
  if (!active) return;   // no workgroup barrier below
  const int qpos = a.q_pos0 + qi;
  float badd[NB][16];
  if (a.bias) {
    const int qrow = qpos < a.bias_rows ? qpos : a.bias_rows - 1;
    const float *brow = a.bias + ((size_t)h * a.bias_rows + qrow) * a.bias_ld;
    if ((a.bias_ld & 3) == 0 && a.bias_ld >= 32 * NB && ((uintptr_t)a.bias & 15) == 0) {
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 x = *reinterpret_cast<const float4 *>(brow + 32 * n + 8 * g + 4 * half);
          badd[n][4 * g] = x.x; badd[n][4 * g + 1] = x.y; badd[n][4 * g + 2] = x.z; badd[n][4 * g + 3] = x.w;
        }
    } else {
#pragma unroll
      for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = 32 * n + (r & 3) + 8 * (r >> 2) + 4 * half;
          badd[n][r] = brow[key < a.bias_ld ? key : a.bias_ld - 1];
        }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) badd[n][r] = 0.f;
  }
  float m = -INFINITY;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 mk = *reinterpret_cast<const float4 *>(smask + 32 * n + 8 * g + 4 * half);
      const float mk4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e, key = 32 * n + 8 * g + 4 * half + e;
        float add = mk4[e];
        if (a.bias) add += badd[n][r];
        add += (a.causal && key > qpos) ? -1e9f : 0.f;
        const float v = key < tk ? ldexpf(sc[n][r], -e_s) + add : -INFINITY;
        sc[n][r] = v;
        m = fmaxf(m, v);
      }
    }
  m = fmaxf(m, __shfl_xor(m, 32));
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // exp(x) = 2^t (1 + c ln 2) with t = fl(x log2 e) and c the product's rounding error + x times log2 e's own low part:
      // six vector instructions, ~1 ulp (plain v_exp_f32 of fl(x log2 e) is off by |x| 2^-24 relative: 2e-6 at x = -38)
      const float x = fmaxf(sc[n][r] - m, -200.f);         // masked keys are -inf: clamp (2^-288 = 0), no inf - inf below
      const float t_ = x * 1.44269502162933349609375f;
      const float c_ = fmaf(x, 1.44269502162933349609375f, -t_) + x * 1.925963033500011e-8f;
      const float e2 = __builtin_amdgcn_exp2f(t_);
      const float e = fmaf(e2, c_ * 0.693147182464599609375f, e2);
      sc[n][r] = e;
      sum += e;
    }
  sum += __shfl_xor(sum, 32);
  const float inv = 16384.f / sum;               // P 2^14 = e (2^14 / sum): one division per row, not per probability
  am_f32x16 o[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[c][r] = 0.f;
#pragma unroll
  for (int n = 0; n < NB; ++n)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ah_f16x8 ph, pl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = sc[n][8 * u + j] * inv;                 // P 2^14: the pair keeps 22 bits down to p = 2^-17
        ph[j] = (_Float16)pv;
        pl[j] = (_Float16)(pv - (float)ph[j]);
      }
      const int ch = 2 * (2 * n + u) + half;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const ah_f16x8 vh = *reinterpret_cast<const ah_f16x8 *>(vth + ah_vt16(32 * c + lrow, ch));
        const ah_f16x8 vl = *reinterpret_cast<const ah_f16x8 *>(vtl + ah_vt16(32 * c + lrow, ch));
        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph, o[c], 0, 0, 0);
        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph, o[c], 0, 0, 0);
        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl, o[c], 0, 0, 0);
      }
    }
  // O^T: col = lane & 31 = query, row = head dim -> through the wave's own rows of the K region, then whole rows out
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4 *>(so + am_sw4(lrow, 8 * c + 2 * g + half)) =
          make_float4(ldexpf(o[c][4 * g], -e_o), ldexpf(o[c][4 * g + 1], -e_o), ldexpf(o[c][4 * g + 2], -e_o),
                      ldexpf(o[c][4 * g + 3], -e_o));
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int nrow = tq - 32 * w < 32 ? tq - 32 * w : 32;
  const int sub = lane >> 4, c4 = lane & 15;
  for (int r4 = 0; r4 < nrow; r4 += 4) {
    const int rr = r4 + sub;
    if (rr < nrow)
      put_ctx4(a, og + (size_t)(32 * w + rr) * a.o_ts + 4 * c4, *reinterpret_cast<const float4 *>(so + am_sw4(rr, c4)));
  }
}

__global__ __launch_bounds__(256, 2) void attention_h16_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smb[];
  char *kh = smb, *kl = smb + AM_S * AM_D * 2;                   // K hi | lo planes, [128][64] halves
  char *vth = smb + 2 * AM_S * AM_D * 2, *vtl = vth + AM_S * AM_D * 2;   // V^T hi | lo planes, [64][128] halves (Q first)
  float *smask = reinterpret_cast<float *>(smb + 4 * AM_S * AM_D * 2);   // [128] additive key mask, then 12 floats of maxima
  float *smax = smask + AM_S;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const long long r0 = a.seq_off ? a.seq_off[b] : 0;
  const int tk = a.seq_off ? (int)(a.seq_off[b + 1] - r0) : a.tk;  // == tq
  const int tq = tk;
  const long long *key_mask = a.seq_off ? nullptr : a.key_mask;
  const float *qg = a.q + (a.seq_off ? (size_t)r0 * a.q_ts : (size_t)b * a.q_bs) + (size_t)h * AM_D;
  const float *kg = a.k + (a.seq_off ? (size_t)r0 * a.k_ts : (size_t)b * a.k_bs) + (size_t)h * AM_D;
  const float *vg = a.v + (a.seq_off ? (size_t)r0 * a.v_ts : (size_t)b * a.v_bs) + (size_t)h * AM_D;
  const size_t og = (a.seq_off ? (size_t)r0 * a.o_ts : (size_t)b * a.o_bs) + (size_t)h * AM_D;
  long long mval = 1;   // key t's mask word: asked for here, stored behind the operand loads (not a round trip of its own)
  if (key_mask && t < tk) mval = key_mask[(size_t)b * tk + t];
  // row-contiguous loads: q / k as (row, dim quad) per thread and pass, v as (four consecutive keys, dim quad)
  float4 q4[8], k4[8], v4[2][4];
  float mq = 0.f, mk_ = 0.f, mv = 0.f;
  auto amax4 = [](float m_, const float4 &x) { return fmaxf(fmaxf(m_, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w))); };
  // All 24 loads first, unconditional (row index clamped to the last real row, zeroed afterwards): with `if (r < tk)` around a
  // load and its first use in one loop body every iteration waited for its own data -- nine memory round trips in a row per
  // workgroup (as attention_mfma16_kernel, profiles/r04_attention_pmc.txt).  Same values, same bits.
  const float4 zero4f = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tk > 0) {   // uniform
    const int last = tk - 1;
    const unsigned q_ts32 = (unsigned)a.q_ts, k_ts32 = (unsigned)a.k_ts, v_ts32 = (unsigned)a.v_ts;   // < 2^24 (checked at launch), rows < 128
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int i = t + 256 * it, r = min(i >> 4, last), c4 = i & 15;
      q4[it] = *reinterpret_cast<const float4 *>(qg + ((unsigned)r * q_ts32 + 4u * c4));
      k4[it] = *reinterpret_cast<const float4 *>(kg + ((unsigned)r * k_ts32 + 4u * c4));
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = t + 256 * it, kg4 = i >> 4, c4 = i & 15;
#pragma unroll
      for (int e = 0; e < 4; ++e) v4[it][e] = *reinterpret_cast<const float4 *>(vg + ((unsigned)min(4 * kg4 + e, last) * v_ts32 + 4u * c4));
    }
  }
  if (t < AM_S) smask[t] = mval == 0 ? -1e9f : 0.f;
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = (t + 256 * it) >> 4;
    if (r < tk) {  // rows past tk are zero
      const float4 x = q4[it];
      q4[it] = make_float4(x.x * a.scale, x.y * a.scale, x.z * a.scale, x.w * a.scale);
    } else {
      q4[it] = zero4f, k4[it] = zero4f;
    }
    mq = amax4(mq, q4[it]);
    mk_ = amax4(mk_, k4[it]);
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int kg4 = (t + 256 * it) >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (4 * kg4 + e >= tk) v4[it][e] = zero4f;
      mv = amax4(mv, v4[it][e]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mq = fmaxf(mq, __shfl_xor(mq, off));
    mk_ = fmaxf(mk_, __shfl_xor(mk_, off));
    mv = fmaxf(mv, __shfl_xor(mv, off));
  }
  if (lane == 0) smax[3 * w] = mq, smax[3 * w + 1] = mk_, smax[3 * w + 2] = mv;
  __syncthreads();
  mq = fmaxf(fmaxf(smax[0], smax[3]), fmaxf(smax[6], smax[9]));
  mk_ = fmaxf(fmaxf(smax[1], smax[4]), fmaxf(smax[7], smax[10]));
  mv = fmaxf(fmaxf(smax[2], smax[5]), fmaxf(smax[8], smax[11]));
  const int eq = ah_exp_for(mq), ek = ah_exp_for(mk_), ev = ah_exp_for(mv);
  auto split4 = [](const float4 &x, int e, ah_f16x4 &hi, ah_f16x4 &lo) {
    const float y[4] = {ldexpf(x.x, e), ldexpf(x.y, e), ldexpf(x.z, e), ldexpf(x.w, e)};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      hi[i] = (_Float16)y[i];
      lo[i] = (_Float16)(y[i] - (float)hi[i]);
    }
  };
  // K into its planes, Q through the V^T region
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int i = t + 256 * it, r = i >> 4, c4 = i & 15;
    ah_f16x4 hi, lo;
    split4(k4[it], ek, hi, lo);
    *reinterpret_cast<ah_f16x4 *>(kh + ah_row8(r, c4)) = hi;
    *reinterpret_cast<ah_f16x4 *>(kl + ah_row8(r, c4)) = lo;
    split4(q4[it], eq, hi, lo);
    *reinterpret_cast<ah_f16x4 *>(vth + ah_row8(r, c4)) = hi;
    *reinterpret_cast<ah_f16x4 *>(vtl + ah_row8(r, c4)) = lo;
  }
  __syncthreads();
  ah_f16x8 qh[4], ql[4];     // the lane's Q fragments: row 32 w + lrow, dims 16 st + 8 half .. + 7
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    qh[st] = *reinterpret_cast<const ah_f16x8 *>(vth + ah_row16(32 * w + lrow, 2 * st + half));
    ql[st] = *reinterpret_cast<const ah_f16x8 *>(vtl + ah_row16(32 * w + lrow, 2 * st + half));
  }
  __syncthreads();
  // V transposed: keys 4 kg4 .. + 3 of dim d = 4 c4 + i -> slots 8 (kg4 & 1) + 4 ((kg4 >> 1) & 1) + e of key block kg4 >> 2
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int i = t + 256 * it, kg4 = i >> 4, c4 = i & 15;
    const int chunk = 2 * (kg4 >> 2) + (kg4 & 1), inner = ((kg4 >> 1) & 1) << 3;
    const float vd[4][4] = {{v4[it][0].x, v4[it][1].x, v4[it][2].x, v4[it][3].x}, {v4[it][0].y, v4[it][1].y, v4[it][2].y, v4[it][3].y},
                            {v4[it][0].z, v4[it][1].z, v4[it][2].z, v4[it][3].z}, {v4[it][0].w, v4[it][1].w, v4[it][2].w, v4[it][3].w}};
#pragma unroll
    for (int d_ = 0; d_ < 4; ++d_) {
      ah_f16x4 hi, lo;
      split4(make_float4(vd[d_][0], vd[d_][1], vd[d_][2], vd[d_][3]), ev, hi, lo);
      const int d = 4 * c4 + d_;
      *reinterpret_cast<ah_f16x4 *>(vth + ah_vt16(d, chunk) + inner) = hi;
      *reinterpret_cast<ah_f16x4 *>(vtl + ah_vt16(d, chunk) + inner) = lo;
    }
  }
  __syncthreads();
  float *so = reinterpret_cast<float *>(smb) + w * 32 * AM_D;      // output rows of wave w: 8 KiB of the K planes
  const int e_s = eq + ek, e_o = ev + 14;
  switch ((tk + 31) / 32) {
    case 1: ah_wave<1>(a, kh, kl, vth, vtl, smask, so, qh, ql, e_s, e_o, w, h, tk, tq, og); break;
    case 2: ah_wave<2>(a, kh, kl, vth, vtl, smask, so, qh, ql, e_s, e_o, w, h, tk, tq, og); break;
    case 3: ah_wave<3>(a, kh, kl, vth, vtl, smask, so, qh, ql, e_s, e_o, w, h, tk, tq, og); break;
    default: ah_wave<4>(a, kh, kl, vth, vtl, smask, so, qh, ql, e_s, e_o, w, h, tk, tq, og); break;
  }
}

// Attention of SMALL groups on the f32 matrix cores in 16 x 16 blocks (v_mfma_f32_16x16x4_f32): one wave per (group, head),
// a group = the <= 32 tokens of a sequence (self-attention of the query encoders, mode 0: rows attend to the rows of their own
// sequence) or the <= 32 rows that share one K|V (decode-step cross-attention, mode 1: the kv_div beams of a query against
// its <= 32 encoder positions; kv_div = 1: the towers' single decoder position).  A typical group -- 11 tokens, or 10 beams x
// 11 keys -- is ONE block pair = 32 matrix instructions of 32 cycles; 32 x 32 blocks (am_wave<1>, tried first: 401 us per call
// for the encoders' self-attention) cost 64 instructions of 64 cycles for the same group, the scalar kernels (549 us) a
// 64-step fmaf chain per (row, key) with most lanes idle.
//   S^T = K . Q^T   per (key block, query block): A = K (lane: key l & 15, dims 4 s + (l >> 4)), B = Q^T -- a lane ends up with
//                    keys 16 kb + 4 (l >> 4) + r (r = 0..3) of ONE query (l & 15); the softmax reduces over r and lanes l ^ 16, l ^ 32
//   O^T = V^T . P^T per (query block, 16-dim tile c): step r multiplies the four keys the lanes of a column hold in
//                    register r (keys 4 (l >> 4) + r) -- any pairing the A operand follows, as in attention_mfma_kernel
// Blocks are independent and a masked / absent key contributes exact zeros, so the padded and the packed layouts agree bit for
// bit, whatever the padding.  Operands pass through LDS once (row-contiguous global loads; K and Q rows stored with their
// dims permuted d -> (d & 3) * 16 + (d >> 2) so that a lane's 16 fragment values are four b128 reads); Q passes through the V
// region on its way to the registers.  Summation order (fixed): scores over dims in groups (4 s .. 4 s + 3), row sums over
// the lane's registers then lanes ^16, ^32, context over keys in groups {r, 4 + r, 8 + r, 12 + r} per block.
typedef float a16_f4 __attribute__((ext_vector_type(4)));
constexpr int A16_LD = 68;                               // floats per staged row: 16-byte aligned, 4 banks apart
constexpr int A16_WAVE_FLOATS = 2 * 32 * A16_LD + 32;    // K | V (Q first) | key mask
constexpr size_t A16_LDS = (size_t)4 * A16_WAVE_FLOATS * sizeof(float);
// what a wave of attention_mfma16_kernel works on (all wave-uniform)
struct A16Work {
  const float *qg, *kg, *vg;
  size_t ooff, q_rs, o_rs, k_rs, v_rs;
  const long long *mrow;
  float *sk, *sv, *smask;
  int nq, tk, qstep, h, KB, QB;
};

// KBM / QBM: key / query blocks of 16 the code is laid out for (1 or 2); a wave whose group needs one of each -- queries of up
// to 16 tokens, ten beams -- runs the <1, 1> body: no block loops, no wave-uniform branches around them, half the registers
template <int KBM, int QBM>
__device__ __forceinline__ void a16_body(const AttnArgs &a, const A16Work &wk) {
  const int lane = threadIdx.x & 63;
  const int n = lane & 15, kq = lane >> 4;
  const float *qg = wk.qg, *kg = wk.kg, *vg = wk.vg;
  const size_t ooff = wk.ooff, q_rs = wk.q_rs, o_rs = wk.o_rs, k_rs = wk.k_rs, v_rs = wk.v_rs;
  const long long *mrow = wk.mrow;
  float *sk = wk.sk, *sv = wk.sv, *smask = wk.smask;
  const int nq = wk.nq, tk = wk.tk, qstep = wk.qstep, h = wk.h;
  const int KB = KBM == 1 ? 1 : wk.KB, QB = QBM == 1 ? 1 : wk.QB;
  const int rows_staged = 16 * (KB > QB ? KB : QB);
  long long mval = 1;   // the key mask of key `lane` (padded layouts): read here, used after the staging loads are under way
  if (mrow && lane < tk) mval = mrow[lane];
  // staging: iteration `it` = rows 4 it .. 4 it + 3, 16 lanes x 16 B per row
  // (the kernel is bound by the instructions it issues -- 2 waves per SIMD keep it 55 % busy, profiles/r04_attention_pmc.txt --
  // so registers that are never read are not cleared, x 1.0 is not multiplied and the first product of every block takes the
  // constant 0 as its C operand)
  // Every global load first, then the LDS writes: with a row's loads and its writes in one loop body the writes wait for that
  // row's data before the next row's loads are issued -- four (eight) memory round trips in a row, 58 % of a wave's life.
  // The loads are unconditional (row index clamped to a real row; offsets in 32 bits from the wave-uniform bases): a K row
  // past tk only makes scores that are replaced by -inf, a Q row past nq a column that is never stored; V rows past tk are
  // zeroed when written (their weights are exact zeros).
  // the bias rows of the lane's query columns with the first loads, not after the score products (one more round trip there)
  const bool bias4 = a.bias && (a.bias_ld & 3) == 0 && a.bias_ld >= 16 * KB && ((uintptr_t)a.bias & 15) == 0;
  float4 bpre[QBM][KBM];
  if (bias4) {
#pragma unroll
    for (int qb = 0; qb < QBM; ++qb) {
      if (qb >= QB) break;
      const int qpos = a.q_pos0 + qstep * (16 * qb + n);
      const float *brow = a.bias + ((size_t)h * a.bias_rows + (qpos < a.bias_rows ? qpos : a.bias_rows - 1)) * a.bias_ld;
#pragma unroll
      for (int kb = 0; kb < KBM; ++kb) {
        if (kb >= KB) break;
        bpre[qb][kb] = *reinterpret_cast<const float4 *>(brow + 16 * kb + 4 * kq);
      }
    }
  }
  float4 v4[4 * (KBM > QBM ? KBM : QBM)], k4[4 * (KBM > QBM ? KBM : QBM)], q4[4 * (KBM > QBM ? KBM : QBM)];
  const bool unit_scale = a.scale == 1.0f;   // T5 (no 1/sqrt(d)); x * 1.0f == x bit for bit
  const int k_rs32 = (int)k_rs, v_rs32 = (int)v_rs, q_rs32 = (int)q_rs;
#pragma unroll
  for (int it = 0; it < 4 * (KBM > QBM ? KBM : QBM); ++it) {
    if (4 * it >= rows_staged) continue;   // wave-uniform; v4[it] is read for 4 it < 16 KB <= rows_staged only
    const int r = 4 * it + kq;
    if (tk > 0) {
      const int rk = min(r, tk - 1);
      k4[it] = *reinterpret_cast<const float4 *>(kg + (unsigned)(rk * k_rs32 + 4 * n));
      v4[it] = *reinterpret_cast<const float4 *>(vg + (unsigned)(rk * v_rs32 + 4 * n));
    } else {
      k4[it] = v4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    q4[it] = *reinterpret_cast<const float4 *>(qg + (unsigned)(min(r, nq - 1) * q_rs32 + 4 * n));   // nq >= 1 for a live wave
  }
  if (lane < 32) smask[lane] = mval == 0 ? -1e9f : 0.f;
#pragma unroll
  for (int it = 0; it < 4 * (KBM > QBM ? KBM : QBM); ++it) {
    if (4 * it >= rows_staged) continue;
    const int r = 4 * it + kq;
    float *kr = sk + r * A16_LD + n, *qr = sv + r * A16_LD + n;   // dim 4 n + e -> position e * 16 + n
    kr[0] = k4[it].x; kr[16] = k4[it].y; kr[32] = k4[it].z; kr[48] = k4[it].w;
    float4 x = q4[it];
    if (!unit_scale) x = make_float4(x.x * a.scale, x.y * a.scale, x.z * a.scale, x.w * a.scale);
    qr[0] = x.x; qr[16] = x.y; qr[32] = x.z; qr[48] = x.w;
    if (r >= tk) v4[it] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float qf[QBM][16];
#pragma unroll
  for (int qb = 0; qb < QBM; ++qb)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (qb < QB) x = *reinterpret_cast<const float4 *>(sv + (16 * qb + n) * A16_LD + kq * 16 + 4 * j);
      qf[qb][4 * j] = x.x; qf[qb][4 * j + 1] = x.y; qf[qb][4 * j + 2] = x.z; qf[qb][4 * j + 3] = x.w;
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 4 * (KBM > QBM ? KBM : QBM); ++it) {
    if (4 * it >= 16 * KB) continue;
    *reinterpret_cast<float4 *>(sv + (4 * it + kq) * A16_LD + 4 * n) = v4[it];
  }
  // S^T blocks
  a16_f4 sc[KBM][QBM];   // blocks (kb < KB, qb < QB) only are written and read
  const a16_f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KBM; ++kb) {
    if (kb >= KB) break;
    float kf[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 x = *reinterpret_cast<const float4 *>(sk + (16 * kb + n) * A16_LD + kq * 16 + 4 * j);
      kf[4 * j] = x.x; kf[4 * j + 1] = x.y; kf[4 * j + 2] = x.z; kf[4 * j + 3] = x.w;
    }
#pragma unroll
    for (int qb = 0; qb < QBM; ++qb) {
      if (qb >= QB) break;
      sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[0], qf[qb][0], zero4, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < 16; ++s) sc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s], qf[qb][s], sc[kb][qb], 0, 0, 0);
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // V is in place for the context phase
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // scores + bias + masks, softmax per query column
  float sum[QBM];
#pragma unroll
  for (int qb = 0; qb < QBM; ++qb) {
    sum[qb] = 1.f;
    if (qb >= QB) break;
    const int qpos = a.q_pos0 + qstep * (16 * qb + n);
    const float *brow = nullptr;
    if (a.bias) brow = a.bias + ((size_t)h * a.bias_rows + (qpos < a.bias_rows ? qpos : a.bias_rows - 1)) * a.bias_ld;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
      if (kb >= KB) break;
      const int k0 = 16 * kb + 4 * kq;
      float badd[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        if (bias4) {
          const float4 x = bpre[qb][kb];
          badd[0] = x.x; badd[1] = x.y; badd[2] = x.z; badd[3] = x.w;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) badd[r] = brow[k0 + r < a.bias_ld ? k0 + r : a.bias_ld - 1];
        }
      }
      const float4 mk = *reinterpret_cast<const float4 *>(smask + k0);
      const float mk4[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + r;
        float add = mk4[r] + badd[r];
        add += (a.causal && key > qpos) ? -1e9f : 0.f;
        const float v = key < tk ? sc[kb][qb][r] + add : -INFINITY;
        sc[kb][qb][r] = v;
        m = fmaxf(m, v);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sm_ = 0.f;
#pragma unroll
    for (int kb = 0; kb < KBM; ++kb) {
      if (kb >= KB) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(sc[kb][qb][r] - m);
        sc[kb][qb][r] = e;
        sm_ += e;
      }
    }
    sm_ += __shfl_xor(sm_, 16);
    sm_ += __shfl_xor(sm_, 32);
    sum[qb] = sm_;
  }
  // O^T = V^T . P^T
  a16_f4 o[QBM][4];    // blocks qb < QB only
  if (KBM > 1 && KB == 0) {   // a group without keys: the context is zero
#pragma unroll
    for (int qb = 0; qb < QBM; ++qb)
#pragma unroll
      for (int c = 0; c < 4; ++c) o[qb][c] = zero4;
  }
#pragma unroll
  for (int kb = 0; kb < KBM; ++kb) {
    if (kb >= KB) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float vf[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) vf[c] = sv[(16 * kb + 4 * kq + r) * A16_LD + 16 * c + n];
#pragma unroll
      for (int qb = 0; qb < QBM; ++qb) {
        if (qb >= QB) break;
        const float p = sc[kb][qb][r] / sum[qb];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          o[qb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c], p, (kb == 0 && r == 0) ? zero4 : o[qb][c], 0, 0, 0);
      }
    }
  }
  // a lane holds dims 16 c + 4 kq .. + 3 of query 16 qb + n
#pragma unroll
  for (int qb = 0; qb < QBM; ++qb) {
    if (qb >= QB) break;
    const int qi = 16 * qb + n;
    if (qi < nq) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        put_ctx4(a, ooff + (size_t)qi * o_rs + 16 * c + 4 * kq, make_float4(o[qb][c][0], o[qb][c][1], o[qb][c][2], o[qb][c][3]));
    }
  }
}

__global__ __launch_bounds__(256, 2) void attention_mfma16_kernel(AttnArgs a, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int D = 64;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float *sk = sm + (size_t)w * A16_WAVE_FLOATS, *sv = sk + 32 * A16_LD, *smask = sv + 32 * A16_LD;
  const long long ngroups = mode == 0 ? a.nb : a.nb / a.kv_div;
  const long long pair = (long long)blockIdx.x * 4 + w;
  const bool live = pair < ngroups * a.H;
  const long long gi = live ? pair / a.H : 0;
  const int h = live ? (int)(pair - gi * a.H) : 0;
  size_t qoff, ooff, q_rs, o_rs;
  const float *kb_, *vb_;
  size_t k_rs, v_rs;
  int nq, tk, qstep;
  const long long *mrow = nullptr;
  if (mode == 0) {
    if (a.seq_off) {
      const long long r0 = a.seq_off[gi];
      tk = (int)(a.seq_off[gi + 1] - r0);
      qoff = (size_t)r0 * a.q_ts, ooff = (size_t)r0 * a.o_ts;
      kb_ = a.k + (size_t)r0 * a.k_ts, vb_ = a.v + (size_t)r0 * a.v_ts;
    } else {
      tk = a.tk;
      qoff = (size_t)gi * a.q_bs, ooff = (size_t)gi * a.o_bs;
      kb_ = a.k + (size_t)gi * a.k_bs, vb_ = a.v + (size_t)gi * a.v_bs;
      if (a.key_mask) mrow = a.key_mask + (size_t)gi * a.tk;
    }
    q_rs = a.q_ts, o_rs = a.o_ts, nq = tk, qstep = 1;
  } else {
    qoff = (size_t)gi * a.kv_div * a.q_bs, ooff = (size_t)gi * a.kv_div * a.o_bs;
    q_rs = a.q_bs, o_rs = a.o_bs, nq = a.kv_div, qstep = 0;
    if (a.kv_off) {
      const long long r0 = a.kv_off[gi];
      tk = (int)(a.kv_off[gi + 1] - r0);
      kb_ = a.k + (size_t)r0 * a.k_ts, vb_ = a.v + (size_t)r0 * a.v_ts;
    } else {
      tk = a.tk;
      kb_ = a.k + (size_t)gi * a.k_bs, vb_ = a.v + (size_t)gi * a.v_bs;
      if (a.key_mask) mrow = a.key_mask + (size_t)gi * a.tk;
    }
  }
  k_rs = a.k_ts, v_rs = a.v_ts;
  if (!live) nq = tk = 0;
  const float *qg = a.q + qoff + (size_t)h * D;
  const float *kg = kb_ + (size_t)h * D, *vg = vb_ + (size_t)h * D;
  ooff += (size_t)h * D;
  A16Work wk;
  wk.qg = qg, wk.kg = kg, wk.vg = vg, wk.ooff = ooff, wk.q_rs = q_rs, wk.o_rs = o_rs, wk.k_rs = k_rs, wk.v_rs = v_rs;
  wk.mrow = mrow, wk.sk = sk, wk.sv = sv, wk.smask = smask, wk.nq = nq, wk.tk = tk, wk.qstep = qstep, wk.h = h;
  wk.KB = (tk + 15) >> 4, wk.QB = (nq + 15) >> 4;      // <= 2 each (wave-uniform)
  if (wk.KB == 1 && wk.QB == 1) a16_body<1, 1>(a, wk);
  else if (wk.KB + wk.QB > 0) a16_body<2, 2>(a, wk);
}

// logits[row, c] = sum_d s[row, d] * (T[trow, c*dim + d] + E[c, d]); one wave per (row, c); trow = row, or
// t_index[row] when the head matrices are looked up in a per-prefix table
__global__ __launch_bounds__(256) void adaptive_logits_kernel(const float *__restrict__ s, long long lds_,
                                                             const float *__restrict__ T, long long ldt,
                                                             const long long *__restrict__ t_index,
                                                             const float *__restrict__ E, long long rows,
                                                             int ncol, int dim, float *__restrict__ out) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * ncol) return;
  const int lane = threadIdx.x & 63;
  const long long r = wid / ncol;
  const int c = (int)(wid - r * ncol);
  const long long tr = t_index ? t_index[r] : r;
  const float4 *sv = reinterpret_cast<const float4 *>(s + r * lds_);
  const float4 *tv = reinterpret_cast<const float4 *>(T + tr * ldt + (size_t)c * dim);
  const float4 *ev = reinterpret_cast<const float4 *>(E + (size_t)c * dim);
  float acc = 0.f;
  for (int i = lane; i < dim / 4; i += 64) {
    const float4 a = sv[i], t4 = tv[i], e4 = ev[i];
    acc = fmaf(a.x, t4.x + e4.x, acc);
    acc = fmaf(a.y, t4.y + e4.y, acc);
    acc = fmaf(a.z, t4.z + e4.z, acc);
    acc = fmaf(a.w, t4.w + e4.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[r * ncol + c] = acc;
}

// logits[row, c] = sum_d (s[row, d] * alpha) * TE[trow, c*dim + d]: TE = the head matrices with lm_head's rows already in them
// (the bias of the GEMM that produced them, the same f32 add adaptive_logits_kernel makes), alpha = d_model^-0.5 of
// modeling_t5.py:1607 applied to the element as scale_kernel would.  One wave per (row, chunk of 64 columns): the row's hidden
// state is read once and stays in registers (adaptive_logits_kernel reads s, t and e per column: three times the L2 traffic,
// which is what bounds the table positions -- 32 .. 32768 prefixes shared by 70 k beams).  Per (row, c) the chain is that
// kernel's: lane l adds its float4 pieces l, l + 64, ... in order, x y z w, then the xor butterfly -- same bits.
template <int NI>
__global__ __launch_bounds__(256) void adaptive_logits_rows_kernel(const float *__restrict__ s, long long lds_, float alpha,
                                                                  const float *__restrict__ TE, long long ldt,
                                                                  const long long *__restrict__ t_index, long long rows,
                                                                  int ncol, int dim, int chunks, float *__restrict__ out) {
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * chunks) return;
  const int lane = threadIdx.x & 63;
  const long long r = wid / chunks;
  const int c0 = (int)(wid - r * chunks) * 64;
  const int c1 = min(ncol, c0 + 64);
  const int d4 = dim / 4;
  const long long tr = t_index ? t_index[r] : r;
  const float4 *sv = reinterpret_cast<const float4 *>(s + r * lds_);
  float4 a[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int i = lane + 64 * j;
    a[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < d4) {
      const float4 v = sv[i];
      a[j] = make_float4(v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha);
    }
  }
  const float *tb = TE + tr * ldt;
  float keep = 0.f;
  for (int c = c0; c < c1; c += 4) {
    float4 t4[4][NI];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float4 *tv = reinterpret_cast<const float4 *>(tb + (size_t)min(c + u, c1 - 1) * dim);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int i = lane + 64 * j;
        t4[u][j] = i < d4 ? tv[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < NI; ++j)
        if (lane + 64 * j < d4) {
          acc = fmaf(a[j].x, t4[u][j].x, acc);
          acc = fmaf(a[j].y, t4[u][j].y, acc);
          acc = fmaf(a[j].z, t4[u][j].z, acc);
          acc = fmaf(a[j].w, t4[u][j].w, acc);
        }
      acc = wave_sum(acc);
      if (lane == c + u - c0) keep = acc;
    }
  }
  if (c0 + lane < c1) out[r * ncol + c0 + lane] = keep;
}

// The same head at dim = 768 in the summation order of the FUSED form (gemm_split.hip, gemm_split16_kernel<0, 3>: the head GEMM
// whose epilogue multiplies its 256 x 256 tile with the hidden states instead of storing it), so that a prefix-table row and a
// row computed per beam give the same bits.  A column's 768 products are summed as that kernel's geometry dictates:
//   chain   C[third][grp][wm][kq]  = 16 fmaf in order (mi, j) over d = 256 third + 128 grp + 64 wm + 16 mi + 4 kq + j
//   wave    P[third][grp][wm]      = (C[0] + C[1]) + (C[2] + C[3])             (lanes ^16, ^32 of the MFMA layout)
//   tile    T[third]               = (P[0][0] + P[0][1]) + (P[1][0] + P[1][1]) (mevi_logits_finish_f32 for the fused form)
//   logit                          = (T[0] + T[1]) + T[2]
// Here lane L < 48 is the chain (third, grp, wm, kq) = (L >> 4, (L >> 3) & 1, (L >> 2) & 1, L & 3); the trees are xor 1, 2, 4, 8.
__global__ __launch_bounds__(256) void adaptive_logits_rows768_kernel(const float *__restrict__ s, long long lds_, float alpha,
                                                                     const float *__restrict__ TE, long long ldt,
                                                                     const long long *__restrict__ t_index, long long rows,
                                                                     int ncol, int chunks, float *__restrict__ out) {
  constexpr int dim = 768;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= rows * chunks) return;
  const int lane = threadIdx.x & 63;
  const long long r = wid / chunks;
  const int c0 = (int)(wid - r * chunks) * 64;
  const int c1 = min(ncol, c0 + 64);
  const long long tr = t_index ? t_index[r] : r;
  const bool on = lane < 48;
  const int d0 = on ? 256 * (lane >> 4) + 128 * ((lane >> 3) & 1) + 64 * ((lane >> 2) & 1) + 4 * (lane & 3) : 0;   // + 16 mi
  float4 a[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const float4 v = *reinterpret_cast<const float4 *>(s + r * lds_ + d0 + 16 * mi);
    a[mi] = make_float4(v.x * alpha, v.y * alpha, v.z * alpha, v.w * alpha);
  }
  const float *tb = TE + tr * ldt + d0;
  float keep = 0.f;
  for (int c = c0; c < c1; c += 4) {
    float4 t4[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float *tv = tb + (size_t)min(c + u, c1 - 1) * dim;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) t4[u][mi] = *reinterpret_cast<const float4 *>(tv + 16 * mi);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float acc = 0.f;
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        acc = fmaf(a[mi].x, t4[u][mi].x, acc);
        acc = fmaf(a[mi].y, t4[u][mi].y, acc);
        acc = fmaf(a[mi].z, t4[u][mi].z, acc);
        acc = fmaf(a[mi].w, t4[u][mi].w, acc);
      }
      if (!on) acc = 0.f;
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      acc += __shfl_xor(acc, 8);
      const float t1 = __shfl(acc, 16), t2 = __shfl(acc, 32), t0 = __shfl(acc, 0);
      const float z = (t0 + t1) + t2;
      if (lane == c + u - c0) keep = z;
    }
  }
  if (c0 + lane < c1) out[r * ncol + c0 + lane] = keep;
}

// The fused form's last step: part [ncol][3][rows][4] wave partials (thirds x (grp, wm)) -> out[row, c], in the order above.
__global__ __launch_bounds__(256) void logits_finish_kernel(const float *__restrict__ part, long long rows, int ncol,
                                                           float *__restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;     // (c, row), row fastest: the reads are contiguous
  if (i >= rows * ncol) return;
  const int c = (int)(i / rows);
  const long long r = i - (long long)c * rows;
  float t[3];
#pragma unroll
  for (int th = 0; th < 3; ++th) {
    const float4 p = *reinterpret_cast<const float4 *>(part + (((size_t)c * 3 + th) * rows + r) * 4);
    t[th] = (p.x + p.y) + (p.z + p.w);
  }
  out[r * ncol + c] = (t[0] + t[1]) + t[2];
}

inline unsigned blocks4(long long waves) { return (unsigned)((waves + 3) / 4); }

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" int mevi_rmsnorm_f32(const float *x, int64_t ldx, const float *w, float eps, int64_t rows, int64_t dim,
                                float *out, int64_t ldo, void *stream) {
  MEVI_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0, MEVI_ERR_INVALID_ARG,
               "rmsnorm: dim/ld must be multiples of 4");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(x && w && out, MEVI_ERR_INVALID_ARG, "rmsnorm: null pointer");
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(blocks4(rows)), dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, w,
                     eps, (long long)rows, (int)dim, out, (long long)ldo);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_add_layernorm_f32(const float *x, int64_t ldx, const float *y, int64_t ldy, const float *cvec,
                                      const float *w, const float *b, float eps, int64_t rows, int64_t dim,
                                      float *out, int64_t ldo, void *stream) {
  MEVI_REQUIRE(rows >= 0 && dim > 0 && dim <= 8192, MEVI_ERR_INVALID_ARG, "add_layernorm: bad shape");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(x && w && b && out, MEVI_ERR_INVALID_ARG, "add_layernorm: null pointer");
  hipLaunchKernelGGL(add_layernorm_kernel, dim3(blocks4(rows)), dim3(256), (size_t)4 * dim * sizeof(float),
                     (hipStream_t)stream, x, (long long)ldx, y, (long long)ldy, cvec, w, b, eps, (long long)rows,
                     (int)dim, out, (long long)ldo);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_gather_rows_f32(const float *table, int64_t ldt, const int64_t *idx, int64_t n, int64_t dim,
                                    float *out, int64_t ldo, void *stream) {
  MEVI_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0 && ldt % 4 == 0 && ldo % 4 == 0, MEVI_ERR_INVALID_ARG,
               "gather_rows: dim/ld must be multiples of 4");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(table && idx && out, MEVI_ERR_INVALID_ARG, "gather_rows: null pointer");
  hipLaunchKernelGGL(gather_rows_i64_kernel, dim3(blocks4(n)), dim3(256), 0, (hipStream_t)stream, table,
                     (long long)ldt, reinterpret_cast<const long long *>(idx), (long long)n, (int)dim, out,
                     (long long)ldo);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_scatter_rows_f32(const float *src, int64_t lds_, const int64_t *idx, int64_t n, int64_t dim,
                                     float *out, int64_t ldo, void *stream) {
  MEVI_REQUIRE(n >= 0 && dim > 0 && dim % 4 == 0 && lds_ % 4 == 0 && ldo % 4 == 0, MEVI_ERR_INVALID_ARG,
               "scatter_rows: dim/ld must be multiples of 4");
  if (n == 0) return MEVI_OK;
  MEVI_REQUIRE(src && idx && out, MEVI_ERR_INVALID_ARG, "scatter_rows: null pointer");
  hipLaunchKernelGGL(scatter_rows_i64_kernel, dim3(blocks4(n)), dim3(256), 0, (hipStream_t)stream, src, (long long)lds_,
                     reinterpret_cast<const long long *>(idx), (long long)n, (int)dim, out, (long long)ldo);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_scale_f32(const float *x, float alpha, int64_t n, float *out, void *stream) {
  if (n <= 0) return MEVI_OK;
  MEVI_REQUIRE(x && out, MEVI_ERR_INVALID_ARG, "scale: null pointer");
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, alpha,
                     (long long)n, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

// Where the context goes: f32 `out`, or (img != nullptr) the split image of the o-projection's operand (AttnArgs::oimg)
struct CtxImage {
  _Float16 *img;
  int np, exp;
};
static int ctx_image_check(const CtxImage &ci, int64_t heads, int64_t dh, int64_t o_bs, int64_t o_ts) {
  if (!ci.img) return MEVI_OK;
  MEVI_REQUIRE(ci.np >= heads * dh && ci.np % 32 == 0 && ci.exp >= -100 && ci.exp <= 100 && o_bs % 4 == 0 && o_ts % 4 == 0 &&
                   ((uintptr_t)ci.img % 8) == 0,
               MEVI_ERR_INVALID_ARG, "attention: bad split-image output (np %d, exp %d)", ci.np, ci.exp);
  return MEVI_OK;
}

// Whole sequences of <= 32 tokens with 64-wide heads: the matrix-core kernel (MEVI_ATTN_SHORT=chain keeps the scalar chains)
// Groups of <= 32 rows / keys with 64-wide heads: the matrix-core kernel (MEVI_ATTN_SHORT=chain keeps the scalar kernels)
static bool short_mfma() {
  static const bool chain = [] { const char *e = getenv("MEVI_ATTN_SHORT"); return e && strcmp(e, "chain") == 0; }();
  return !chain;
}
// passage-length self-attention: split-precision f16 matrix cores when the context goes out as a split image (= the caller runs
// the split GEMMs); the f32-MFMA kernel otherwise (MEVI_GEMM=exact) or with MEVI_ATTN_PASSAGE=f32
static int launch_passage(const AttnArgs &a, long long pairs, hipStream_t stream) {
  MEVI_REQUIRE(a.q_ts >= 0 && a.k_ts >= 0 && a.v_ts >= 0 && a.q_ts < (1 << 24) && a.k_ts < (1 << 24) && a.v_ts < (1 << 24),
               MEVI_ERR_UNSUPPORTED, "attention: token strides of 2^24 floats or more");
  static const bool f32only = [] { const char *e = getenv("MEVI_ATTN_PASSAGE"); return e && strcmp(e, "f32") == 0; }();
  if (a.oimg && !f32only) {
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_h16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)AH_LDS));
    hipLaunchKernelGGL(attention_h16_kernel, dim3((unsigned)pairs), dim3(256), AH_LDS, stream, a);
  } else {
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_mfma_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)AM_LDS));
    hipLaunchKernelGGL(attention_mfma_kernel, dim3((unsigned)pairs), dim3(256), AM_LDS, stream, a);
  }
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
static int launch_mfma16(const AttnArgs &a, int mode, long long pairs, hipStream_t stream) {
  // row offsets inside a group are formed in 32 bits (at most 32 rows of it)
  MEVI_REQUIRE(a.k_ts < (1 << 24) && a.v_ts < (1 << 24) && a.q_ts < (1 << 24) && a.q_bs < (1 << 24) && a.k_ts >= 0 && a.v_ts >= 0 &&
                   a.q_ts >= 0 && a.q_bs >= 0,
               MEVI_ERR_UNSUPPORTED, "attention: row strides of 2^24 floats or more");
  MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_mfma16_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)A16_LDS));
  hipLaunchKernelGGL(attention_mfma16_kernel, dim3((unsigned)((pairs + 3) / 4)), dim3(256), A16_LDS, stream, a, mode);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
// 64- and 96-wide heads take the LDS-transposed form; MEVI_ATTN_FEW_KEYS=direct keeps the per-lane row walk (A/B, same bits)
typedef void (*few_keys_fn)(AttnArgs);
static few_keys_fn few_keys_kernel(int64_t dh, int64_t tk) {
  static const bool direct = [] { const char *e = getenv("MEVI_ATTN_FEW_KEYS"); return e && strcmp(e, "direct") == 0; }();
  // tk = 8 (and 7 at 96-wide heads): the staged form runs out of registers -- measured 4x slower than the direct one there
  if ((dh != 64 && dh != 96) || direct || tk < 1 || tk > (dh == 64 ? 7 : 6)) return attention_few_keys_kernel;
#define MEVI_FK(DH_) {attention_few_keys_staged_kernel<1, DH_>, attention_few_keys_staged_kernel<2, DH_>, attention_few_keys_staged_kernel<3, DH_>, \
                      attention_few_keys_staged_kernel<4, DH_>, attention_few_keys_staged_kernel<5, DH_>, attention_few_keys_staged_kernel<6, DH_>, \
                      attention_few_keys_staged_kernel<7, DH_>}
  static const few_keys_fn table64[7] = MEVI_FK(64), table96[7] = MEVI_FK(96);
#undef MEVI_FK
  const few_keys_fn *table = dh == 64 ? table64 : table96;
  return table[tk - 1];
}

static int attention_launch(const float *q, int64_t q_bs, int64_t q_ts, const float *k, int64_t k_bs,
                                  int64_t k_ts, const float *v, int64_t v_bs, int64_t v_ts, float *out,
                                  int64_t o_bs, int64_t o_ts, int64_t nb, int64_t tq, int64_t tk, int64_t heads,
                                  int64_t dh, int64_t kv_div, const float *bias, int64_t bias_rows,
                                  int64_t bias_ld, int64_t q_pos0, const int64_t *key_mask, int causal,
                                  float scale, const int64_t *kv_off, CtxImage ci, void *stream) {
  MEVI_REQUIRE(nb >= 0 && tq > 0 && tk > 0 && heads > 0 && dh > 0 && kv_div > 0, MEVI_ERR_INVALID_ARG,
               "attention: bad shape");
  MEVI_REQUIRE(tk <= 64 * ATT_KPL, MEVI_ERR_UNSUPPORTED, "attention: tk=%lld > %d keys not supported", (long long)tk,
               64 * ATT_KPL);
  MEVI_REQUIRE(dh <= 128, MEVI_ERR_UNSUPPORTED, "attention: head dim %lld > 128 not supported", (long long)dh);
  MEVI_REQUIRE(dh % 4 == 0 && q_bs % 4 == 0 && q_ts % 4 == 0 && k_bs % 4 == 0 && k_ts % 4 == 0,
               MEVI_ERR_INVALID_ARG, "attention: dh and q/k strides must be multiples of 4");
  if (nb == 0) return MEVI_OK;
  MEVI_REQUIRE(q && k && v && (out || ci.img), MEVI_ERR_INVALID_ARG, "attention: null pointer");
  MEVI_REQUIRE(!bias || q_pos0 + tq <= bias_rows, MEVI_ERR_INVALID_ARG, "attention: bias table too small");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.oimg = ci.img; a.onp = ci.np; a.oexp = ci.exp;
  if (int st = ctx_image_check(ci, heads, dh, o_bs, o_ts)) return st;
  a.q_bs = q_bs; a.q_ts = q_ts; a.k_bs = k_bs; a.k_ts = k_ts; a.v_bs = v_bs; a.v_ts = v_ts; a.o_bs = o_bs; a.o_ts = o_ts;
  a.nb = (int)nb; a.tq = (int)tq; a.tk = (int)tk; a.H = (int)heads; a.dh = (int)dh; a.kv_div = (int)kv_div;
  a.bias = bias; a.bias_rows = (int)bias_rows; a.bias_ld = (int)bias_ld; a.q_pos0 = (int)q_pos0;
  a.key_mask = reinterpret_cast<const long long *>(key_mask); a.causal = causal; a.scale = scale;
  a.seq_off = nullptr;
  a.kv_off = reinterpret_cast<const long long *>(kv_off);
  a.key_rows = nullptr;
  const size_t tile_lds = (size_t)tk * (3 * dh + 1) * sizeof(float);
  // attention_mfma16_kernel moves rows as 16-byte pieces (q / k strides are checked above)
  const bool mfma16_aligned = v_bs % 4 == 0 && v_ts % 4 == 0 && o_bs % 4 == 0 && o_ts % 4 == 0 &&
                              (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0;
  if (!kv_off && kv_div == 1 && tq == tk && tk > 64 && tk <= AM_S && dh == AM_D) {  // passages: matrix cores
    return launch_passage(a, (long long)nb * heads, (hipStream_t)stream);
  } else if (!kv_off && kv_div == 1 && tq == tk && tk > 1 && tk <= 32 && dh == AM_D && short_mfma() && mfma16_aligned) {  // query-length sequences
    return launch_mfma16(a, 0, (long long)nb * heads, (hipStream_t)stream);
  } else if (tq == 1 && tk <= 32 && kv_div <= 32 && nb % kv_div == 0 && dh == AM_D && short_mfma() && mfma16_aligned &&
             (kv_off || key_mask || kv_div > 1 || tk > 8)) {   // a few rows against a few shared keys (cross-attention)
    return launch_mfma16(a, 1, (nb / kv_div) * heads, (hipStream_t)stream);
  } else if (!kv_off && tq == 1 && tk <= 8) {  // a handful of cached keys: eight (row, head) pairs per wave
    MEVI_REQUIRE(nb * heads < (1LL << 31) - 8, MEVI_ERR_UNSUPPORTED, "attention: too many (row, head) pairs");
    hipLaunchKernelGGL(few_keys_kernel(dh, tk), dim3(blocks4((nb * heads + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
  } else if (kv_div > 1 && tq == 1 && nb % kv_div == 0 && v_bs % 4 == 0 && v_ts % 4 == 0 &&
             ((size_t)tk * (2 * dh + 4) + (size_t)kv_div * dh) * sizeof(float) <= 65536) {
    // the beams of a query share K|V: stage them once per (query, head)
    hipLaunchKernelGGL(attention_group_kernel, dim3((unsigned)((nb / kv_div) * heads)), dim3(256),
                       ((size_t)tk * (2 * dh + 4) + (size_t)kv_div * dh) * sizeof(float), (hipStream_t)stream, a);
  } else if (!kv_off && kv_div == 1 && tq == tk && tq >= 8 && tile_lds <= 160 * 1024) {  // self-attention over a whole sequence
    if (tile_lds > 65536)  // dynamic LDS beyond 64 KiB must be opted into (128 passage tokens x 64: 97 KiB)
      MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_tile_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
    hipLaunchKernelGGL(attention_tile_kernel, dim3((unsigned)(nb * heads)), dim3(256), tile_lds, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(attention_kernel, dim3(blocks4(nb * heads * tq)), dim3(256), 0, (hipStream_t)stream, a);
  }
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

static int attention_varlen_launch(const float *q, int64_t q_ts, const float *k, int64_t k_ts, const float *v,
                                         int64_t v_ts, float *out, int64_t o_ts, const int64_t *seq_off, int64_t nseq,
                                         int64_t max_len, int64_t heads, int64_t dh, const float *bias,
                                         int64_t bias_rows, int64_t bias_ld, int causal, float scale, CtxImage ci, void *stream) {
  MEVI_REQUIRE(nseq >= 0 && max_len > 0 && heads > 0 && dh > 0, MEVI_ERR_INVALID_ARG, "attention_varlen: bad shape");
  MEVI_REQUIRE(max_len <= 64 * ATT_KPL, MEVI_ERR_UNSUPPORTED, "attention_varlen: %lld > %d keys not supported",
               (long long)max_len, 64 * ATT_KPL);
  MEVI_REQUIRE(dh <= 128 && dh % 4 == 0 && q_ts % 4 == 0 && k_ts % 4 == 0 && v_ts % 4 == 0, MEVI_ERR_INVALID_ARG,
               "attention_varlen: dh (<= 128) and the row strides must be multiples of 4");
  const size_t per_wave = (size_t)max_len * (3 * dh + 1) * sizeof(float);
  MEVI_REQUIRE(per_wave <= 160 * 1024, MEVI_ERR_UNSUPPORTED, "attention_varlen: %lld keys x %lld do not fit LDS",
               (long long)max_len, (long long)dh);
  if (nseq == 0) return MEVI_OK;
  MEVI_REQUIRE(q && k && v && (out || ci.img) && seq_off, MEVI_ERR_INVALID_ARG, "attention_varlen: null pointer");
  MEVI_REQUIRE(!bias || (max_len <= bias_rows && max_len <= bias_ld), MEVI_ERR_INVALID_ARG,
               "attention_varlen: bias table too small");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.oimg = ci.img; a.onp = ci.np; a.oexp = ci.exp;
  if (int st = ctx_image_check(ci, heads, dh, 0, o_ts)) return st;
  a.q_bs = a.k_bs = a.v_bs = a.o_bs = 0;
  a.q_ts = q_ts; a.k_ts = k_ts; a.v_ts = v_ts; a.o_ts = o_ts;
  a.nb = (int)nseq; a.tq = a.tk = (int)max_len; a.H = (int)heads; a.dh = (int)dh; a.kv_div = 1;
  a.bias = bias; a.bias_rows = (int)bias_rows; a.bias_ld = (int)bias_ld; a.q_pos0 = 0;
  a.key_mask = nullptr; a.causal = causal; a.scale = scale;
  a.seq_off = reinterpret_cast<const long long *>(seq_off);
  a.kv_off = nullptr;
  a.key_rows = nullptr;
  const long long pairs = (long long)nseq * heads;
  if (max_len > 64 && max_len <= AM_S && dh == AM_D) {   // passage-length sequences: matrix cores
    a.q_pos0 = 0;
    return launch_passage(a, pairs, (hipStream_t)stream);
  }
  if (max_len <= 32 && dh == AM_D && short_mfma() && o_ts % 4 == 0 &&
      (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0)
    return launch_mfma16(a, 0, pairs, (hipStream_t)stream);
  if (max_len <= 64 && dh == 64) {   // t5-base / bert-base heads, query-length sequences
    hipLaunchKernelGGL(attention_varlen_short_kernel<64>, dim3((unsigned)((pairs + 3) / 4)), dim3(256),
                       (size_t)4 * max_len * 2 * 64 * sizeof(float), (hipStream_t)stream, a, (int)max_len);
    MEVI_HIP_CHECK(hipGetLastError());
    return MEVI_OK;
  }
  // as many waves per workgroup as keep two workgroups' LDS regions on a CU (one when the region is large)
  int waves = (int)(80 * 1024 / per_wave);
  waves = waves < 1 ? 1 : waves > 4 ? 4 : waves;
  const size_t lds = per_wave * waves;
  if (lds > 65536)
    MEVI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(attention_varlen_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(attention_varlen_kernel, dim3((unsigned)((pairs + waves - 1) / waves)), dim3(256), lds,
                     (hipStream_t)stream, a, (int)max_len, waves);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

static int attention_cached_launch(const float *q, int64_t q_bs, const float *k, int64_t k_bs, int64_t k_ts, const float *v,
                                         int64_t v_bs, int64_t v_ts, float *out, int64_t o_bs, int64_t nb, int64_t tk,
                                         int64_t heads, int64_t dh, const int32_t *key_rows, const float *bias,
                                         int64_t bias_rows, int64_t bias_ld, int64_t q_pos0, int causal, float scale,
                                         CtxImage ci, void *stream) {
  MEVI_REQUIRE(nb >= 0 && tk > 0 && tk <= 8 && heads > 0 && dh > 0 && dh <= 128 && dh % 4 == 0, MEVI_ERR_UNSUPPORTED,
               "attention_cached: 1..8 cached positions, head width a multiple of 4 up to 128 (got tk %lld, dh %lld)",
               (long long)tk, (long long)dh);
  MEVI_REQUIRE(q_bs % 4 == 0 && k_bs % 4 == 0 && k_ts % 4 == 0 && v_bs % 4 == 0 && v_ts % 4 == 0, MEVI_ERR_INVALID_ARG,
               "attention_cached: strides must be multiples of 4");
  if (nb == 0) return MEVI_OK;
  MEVI_REQUIRE(q && k && v && (out || ci.img) && key_rows, MEVI_ERR_INVALID_ARG, "attention_cached: null pointer");
  MEVI_REQUIRE(!bias || (q_pos0 < bias_rows && tk <= bias_ld), MEVI_ERR_INVALID_ARG, "attention_cached: bias table too small");
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.out = out;
  a.oimg = ci.img; a.onp = ci.np; a.oexp = ci.exp;
  if (int st = ctx_image_check(ci, heads, dh, o_bs, 0)) return st;
  a.q_bs = q_bs; a.q_ts = 0; a.k_bs = k_bs; a.k_ts = k_ts; a.v_bs = v_bs; a.v_ts = v_ts; a.o_bs = o_bs; a.o_ts = 0;
  a.nb = (int)nb; a.tq = 1; a.tk = (int)tk; a.H = (int)heads; a.dh = (int)dh; a.kv_div = 1;
  a.bias = bias; a.bias_rows = (int)bias_rows; a.bias_ld = (int)bias_ld; a.q_pos0 = (int)q_pos0;
  a.key_mask = nullptr; a.causal = causal; a.scale = scale;
  a.seq_off = nullptr; a.kv_off = nullptr;
  a.key_rows = key_rows;
  MEVI_REQUIRE(nb * heads < (1LL << 31) - 8, MEVI_ERR_UNSUPPORTED, "attention_cached: too many (row, head) pairs");
  hipLaunchKernelGGL(few_keys_kernel(dh, tk), dim3(blocks4((nb * heads + 7) / 8)), dim3(256), 0, (hipStream_t)stream, a);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_attention_f32(const float *q, int64_t q_bs, int64_t q_ts, const float *k, int64_t k_bs,
                                  int64_t k_ts, const float *v, int64_t v_bs, int64_t v_ts, float *out,
                                  int64_t o_bs, int64_t o_ts, int64_t nb, int64_t tq, int64_t tk, int64_t heads,
                                  int64_t dh, int64_t kv_div, const float *bias, int64_t bias_rows,
                                  int64_t bias_ld, int64_t q_pos0, const int64_t *key_mask, int causal,
                                  float scale, const int64_t *kv_off, void *stream) {
  return attention_launch(q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, out, o_bs, o_ts, nb, tq, tk, heads, dh, kv_div, bias,
                          bias_rows, bias_ld, q_pos0, key_mask, causal, scale, kv_off, CtxImage{nullptr, 0, 0}, stream);
}

// The *_split_f16 forms write the context as the (hi, lo) f16 image of the o-projection's operand (gemm_split.hip) scaled by
// 2^out_exp: out_img [rows, 2 * img_np] halves; img_bs / img_ts are the batch / token strides in HALVES (img_ts = 2 * img_np
// for contiguous rows).  The caller guarantees |context| * 2^out_exp < 2^15 (a bound on |V|: the context is a convex
// combination of V rows) and fills the rows' exponent array with out_exp.
extern "C" int mevi_attention_split_f16(const float *q, int64_t q_bs, int64_t q_ts, const float *k, int64_t k_bs,
                                        int64_t k_ts, const float *v, int64_t v_bs, int64_t v_ts, void *out_img,
                                        int64_t img_np, int out_exp, int64_t img_bs, int64_t img_ts, int64_t nb, int64_t tq,
                                        int64_t tk, int64_t heads, int64_t dh, int64_t kv_div, const float *bias,
                                        int64_t bias_rows, int64_t bias_ld, int64_t q_pos0, const int64_t *key_mask,
                                        int causal, float scale, const int64_t *kv_off, void *stream) {
  MEVI_REQUIRE(out_img || nb == 0, MEVI_ERR_INVALID_ARG, "attention_split: null image");
  return attention_launch(q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, nullptr, img_bs, img_ts, nb, tq, tk, heads, dh, kv_div,
                          bias, bias_rows, bias_ld, q_pos0, key_mask, causal, scale, kv_off,
                          CtxImage{reinterpret_cast<_Float16 *>(out_img), (int)img_np, out_exp}, stream);
}

extern "C" int mevi_attention_varlen_f32(const float *q, int64_t q_ts, const float *k, int64_t k_ts, const float *v,
                                         int64_t v_ts, float *out, int64_t o_ts, const int64_t *seq_off, int64_t nseq,
                                         int64_t max_len, int64_t heads, int64_t dh, const float *bias,
                                         int64_t bias_rows, int64_t bias_ld, int causal, float scale, void *stream) {
  return attention_varlen_launch(q, q_ts, k, k_ts, v, v_ts, out, o_ts, seq_off, nseq, max_len, heads, dh, bias, bias_rows,
                                 bias_ld, causal, scale, CtxImage{nullptr, 0, 0}, stream);
}

extern "C" int mevi_attention_varlen_split_f16(const float *q, int64_t q_ts, const float *k, int64_t k_ts, const float *v,
                                               int64_t v_ts, void *out_img, int64_t img_np, int out_exp, int64_t img_ts,
                                               const int64_t *seq_off, int64_t nseq, int64_t max_len, int64_t heads,
                                               int64_t dh, const float *bias, int64_t bias_rows, int64_t bias_ld,
                                               int causal, float scale, void *stream) {
  MEVI_REQUIRE(out_img || nseq == 0, MEVI_ERR_INVALID_ARG, "attention_varlen_split: null image");
  return attention_varlen_launch(q, q_ts, k, k_ts, v, v_ts, nullptr, img_ts, seq_off, nseq, max_len, heads, dh, bias,
                                 bias_rows, bias_ld, causal, scale,
                                 CtxImage{reinterpret_cast<_Float16 *>(out_img), (int)img_np, out_exp}, stream);
}

extern "C" int mevi_attention_cached_f32(const float *q, int64_t q_bs, const float *k, int64_t k_bs, int64_t k_ts, const float *v,
                                         int64_t v_bs, int64_t v_ts, float *out, int64_t o_bs, int64_t nb, int64_t tk,
                                         int64_t heads, int64_t dh, const int32_t *key_rows, const float *bias,
                                         int64_t bias_rows, int64_t bias_ld, int64_t q_pos0, int causal, float scale,
                                         void *stream) {
  return attention_cached_launch(q, q_bs, k, k_bs, k_ts, v, v_bs, v_ts, out, o_bs, nb, tk, heads, dh, key_rows, bias, bias_rows,
                                 bias_ld, q_pos0, causal, scale, CtxImage{nullptr, 0, 0}, stream);
}

extern "C" int mevi_attention_cached_split_f16(const float *q, int64_t q_bs, const float *k, int64_t k_bs, int64_t k_ts,
                                               const float *v, int64_t v_bs, int64_t v_ts, void *out_img, int64_t img_np,
                                               int out_exp, int64_t img_bs, int64_t nb, int64_t tk, int64_t heads, int64_t dh,
                                               const int32_t *key_rows, const float *bias, int64_t bias_rows, int64_t bias_ld,
                                               int64_t q_pos0, int causal, float scale, void *stream) {
  MEVI_REQUIRE(out_img || nb == 0, MEVI_ERR_INVALID_ARG, "attention_cached_split: null image");
  return attention_cached_launch(q, q_bs, k, k_bs, k_ts, v, v_bs, v_ts, nullptr, img_bs, nb, tk, heads, dh, key_rows, bias,
                                 bias_rows, bias_ld, q_pos0, causal, scale,
                                 CtxImage{reinterpret_cast<_Float16 *>(out_img), (int)img_np, out_exp}, stream);
}

extern "C" int mevi_adaptive_logits_rows_f32(const float *s, int64_t lds_, float alpha, const float *te, int64_t ldt,
                                             const int64_t *t_index, int64_t rows, int64_t ncol, int64_t dim, float *out,
                                             void *stream) {
  MEVI_REQUIRE(rows >= 0 && ncol > 0 && dim > 0 && dim % 4 == 0 && dim <= 1024 && lds_ % 4 == 0 && ldt % 4 == 0 && ncol < (1 << 24),
               MEVI_ERR_INVALID_ARG, "adaptive_logits_rows: bad shape (dim a multiple of 4, at most 1024)");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(s && te && out, MEVI_ERR_INVALID_ARG, "adaptive_logits_rows: null pointer");
  MEVI_REQUIRE((((uintptr_t)s | (uintptr_t)te) & 15) == 0, MEVI_ERR_INVALID_ARG, "adaptive_logits_rows: s, te 16-byte aligned");
  const int chunks = (int)((ncol + 63) / 64);
  if (dim == 768) {   // the order of the fused head (see the kernel): a table row and a per-beam row give the same bits
    hipLaunchKernelGGL(adaptive_logits_rows768_kernel, dim3(blocks4(rows * chunks)), dim3(256), 0, (hipStream_t)stream, s,
                       (long long)lds_, alpha, te, (long long)ldt, reinterpret_cast<const long long *>(t_index), (long long)rows,
                       (int)ncol, chunks, out);
    MEVI_HIP_CHECK(hipGetLastError());
    return MEVI_OK;
  }
  const int ni = (int)((dim / 4 + 63) / 64);
  typedef void (*fn_t)(const float *, long long, float, const float *, long long, const long long *, long long, int, int, int,
                       float *);
  static const fn_t table[4] = {adaptive_logits_rows_kernel<1>, adaptive_logits_rows_kernel<2>, adaptive_logits_rows_kernel<3>,
                                adaptive_logits_rows_kernel<4>};
  hipLaunchKernelGGL(table[ni - 1], dim3(blocks4(rows * chunks)), dim3(256), 0, (hipStream_t)stream, s, (long long)lds_, alpha, te,
                     (long long)ldt, reinterpret_cast<const long long *>(t_index), (long long)rows, (int)ncol, (int)dim, chunks, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_logits_finish_f32(const float *part, int64_t rows, int64_t ncol, float *out, void *stream) {
  MEVI_REQUIRE(rows >= 0 && ncol > 0 && rows * ncol < (1LL << 40), MEVI_ERR_INVALID_ARG, "logits_finish: bad shape");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(part && out && ((uintptr_t)part & 15) == 0, MEVI_ERR_INVALID_ARG, "logits_finish: null or unaligned pointer");
  hipLaunchKernelGGL(logits_finish_kernel, dim3((unsigned)((rows * ncol + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                     (long long)rows, (int)ncol, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}

extern "C" int mevi_adaptive_logits_f32(const float *s, int64_t lds_, const float *t, int64_t ldt,
                                        const int64_t *t_index, const float *e, int64_t rows, int64_t ncol,
                                        int64_t dim, float *out, void *stream) {
  MEVI_REQUIRE(rows >= 0 && ncol > 0 && dim > 0 && dim % 4 == 0 && lds_ % 4 == 0 && ldt % 4 == 0,
               MEVI_ERR_INVALID_ARG, "adaptive_logits: bad shape");
  if (rows == 0) return MEVI_OK;
  MEVI_REQUIRE(s && t && e && out, MEVI_ERR_INVALID_ARG, "adaptive_logits: null pointer");
  hipLaunchKernelGGL(adaptive_logits_kernel, dim3(blocks4(rows * ncol)), dim3(256), 0, (hipStream_t)stream, s,
                     (long long)lds_, t, (long long)ldt, reinterpret_cast<const long long *>(t_index), e, (long long)rows,
                     (int)ncol, (int)dim, out);
  MEVI_HIP_CHECK(hipGetLastError());
  return MEVI_OK;
}
