// C[M, N] = act(A[M, K] . W[N, K]^T + bias[N]) + residual[M, N]   (all f32, row-major, K contiguous;
// act: none / relu / erf-gelu)
//
// The linear layers of the T5 stacks: q/k/v/o projections and wi/wo of every block
// (MEVI/transformers/modeling_t5.py:181-186, 217-220, 350-358, 412), the adaptor's packed
// in_proj / out_proj / linear1 / linear2 (torch.nn.TransformerDecoderLayer, modeling_t5.py:1252-1255)
// and adaptor_linear restricted to the valid columns (modeling_t5.py:1677-1682).
// torch.nn.Linear stores W as [out, in], so both operands are K-contiguous: exactly the shape of
// the shared ping-pong f32-MFMA loop (mfma_pp.h).  Every output is the sequential fmaf chain over k.
#include <cstdlib>

#include "mfma_pp.h"

namespace mevi {
namespace {

// NI = 2: 256 x 128 output tiles; NI = 1: 256 x 64 tiles for grids that would leave CUs idle (a decode-step
// GEMM of 5120 x 768 is 120 tiles of the first kind on 256 CUs).
template <int NI, bool KTAIL>
__global__ __launch_bounds__(PP_THREADS, 2) void gemm_nt_kernel(
    const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
    float *__restrict__ C, long long ldc, int M, int N, int K, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act, int n_ntiles, int n_mpairs) {
  constexpr int QT = 64 * NI;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int nwg = n_ntiles * n_mpairs;
  const int wg = xcd_remap(blockIdx.x, nwg);
  const int mpair = wg / n_ntiles;
  const int ntile = wg - mpair * n_ntiles;

  const int t = threadIdx.x;
  const int grp = __builtin_amdgcn_readfirstlane(t >> 8);
  const int tg = t & 255;
  const int lane = t & 63;
  const int wave = tg >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, half = lane >> 5;
  const int srow = tg >> 3, skq = (tg & 7) * 4;
  const int mrow0 = (mpair * 2 + grp) * BM;
  const int nrow0 = ntile * QT;

  const float *aptr[4];
  const float *wptr[NI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int r = mrow0 + srow + 32 * i;
    if (r > M - 1) r = M - 1;
    aptr[i] = A + (size_t)r * lda + skq;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int r = nrow0 + (QT / 2) * grp + srow + 32 * i;
    if (r > N - 1) r = N - 1;
    wptr[i] = W + (size_t)r * ldw + skq;
  }

  f32x16 acc[2][NI];
  pp_mainloop<NI, KTAIL>(aptr, wptr, K, lds, acc);

  // C/D map: col = lane&31 -> n (W row), row = (r&3) + 8*(r>>2) + 4*half -> m (A row)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nrow0 + 32 * NI * wn + 32 * ni + lrow;
    if (n >= N) continue;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mrow0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m < M) {
          float v = acc[mi][ni][r];
          if (bias) v += b;
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));  // erf GELU (BERT 'gelu')
          if (residual) v += residual[(size_t)m * ldr + n];
          C[(size_t)m * ldc + n] = v;
        }
      }
    }
  }
}

// Few outputs (single queries and small batches, the latency path): a 256 x 128 MFMA tile would spend a full tile's time on
// them.  One workgroup per (64 output columns, 4 rows): lane = column, wave = row; W and X slabs of SK_BK k-values go
// through LDS (register prefetch of the next slab), every output is the same sequential fmaf chain over k the MFMA
// kernels produce -- so a row gives the same bits whether it travels alone or inside a large batch.
constexpr int SK_BK = 128, SK_LD = SK_BK + 1;
// taken when m * n <= SK_MAX_OUTPUTS: measured ~0.1 us per 1000 outputs at K = 768 against ~57 us for one round of MFMA tiles
constexpr long long SK_MAX_OUTPUTS = 500000;

__global__ __launch_bounds__(256) void gemm_skinny_kernel(
    const float *__restrict__ A, long long lda, const float *__restrict__ W, long long ldw,
    float *__restrict__ C, long long ldc, int M, int N, int K, const float *__restrict__ bias,
    const float *__restrict__ residual, long long ldr, int act) {
  __shared__ float sw[64 * SK_LD];   // [column][k]
  __shared__ float sx[4 * SK_BK];    // [row][k]
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n0 = blockIdx.x * 64, m0 = blockIdx.y * 4;
  // staging roles: thread t loads W row (t >> 2), k-quads (t & 3) + 4 j (j < 8); threads 0..127 load X row (t >> 5), quad (t & 31)
  const int wr = t >> 2, wq = t & 3;
  const int xr = t >> 5, xq = t & 31;
  const int wrow = n0 + wr < N ? n0 + wr : N - 1;
  const float *wp = W + (size_t)wrow * ldw;
  float4 rw[8], rx;
  auto gload = [&](int k0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 4 * (wq + 4 * j);
      rw[j] = k < K ? *reinterpret_cast<const float4 *>(wp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int r = m0 + xr, k = k0 + 4 * xq;
    rx = (t < 128 && r < M && k < K) ? *reinterpret_cast<const float4 *>(A + (size_t)r * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto lstore = [&]() {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float *p = sw + wr * SK_LD + 4 * (wq + 4 * j);
      p[0] = rw[j].x; p[1] = rw[j].y; p[2] = rw[j].z; p[3] = rw[j].w;
    }
    if (t < 128) *reinterpret_cast<float4 *>(sx + xr * SK_BK + 4 * xq) = rx;
  };
  const int m = m0 + wave;
  float acc = 0.f;
  gload(0);
  for (int k0 = 0; k0 < K; k0 += SK_BK) {
    __syncthreads();                 // the previous slab has been consumed
    lstore();
    __syncthreads();
    if (k0 + SK_BK < K) gload(k0 + SK_BK);
    const int kn = K - k0 < SK_BK ? K - k0 : SK_BK;     // the chain stops at K
    if (m < M) {
      const float *wl = sw + lane * SK_LD, *xl = sx + wave * SK_BK;
      if (kn == SK_BK) {
#pragma unroll 16
        for (int kk = 0; kk < SK_BK; ++kk) acc = fmaf(xl[kk], wl[kk], acc);
      } else {
        for (int kk = 0; kk < kn; ++kk) acc = fmaf(xl[kk], wl[kk], acc);
      }
    }
  }
  const int n = n0 + lane;
  if (n >= N || m >= M) return;
  float v = acc;
  if (bias) v += bias[n];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  if (residual) v += residual[(size_t)m * ldr + n];
  C[(size_t)m * ldc + n] = v;
}

}  // namespace
}  // namespace mevi

using namespace mevi;

extern "C" int mevi_gemm_nt_f32(const float *a, int64_t lda, const float *w, int64_t ldw, float *c, int64_t ldc,
                                int64_t m, int64_t n, int64_t k, const float *bias, const float *residual,
                                int64_t ldr, int act, void *stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  MEVI_REQUIRE(m >= 0 && n >= 0 && k > 0, MEVI_ERR_INVALID_ARG, "gemm_nt: bad shape");
  if (m == 0 || n == 0) return MEVI_OK;
  MEVI_REQUIRE(a && w && c, MEVI_ERR_INVALID_ARG, "gemm_nt: null pointer");
  MEVI_REQUIRE(k % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, MEVI_ERR_UNSUPPORTED,
               "gemm_nt: k, lda, ldw must be multiples of 4 (k=%lld lda=%lld ldw=%lld)", (long long)k,
               (long long)lda, (long long)ldw);
  MEVI_REQUIRE(((uintptr_t)a % 16) == 0 && ((uintptr_t)w % 16) == 0, MEVI_ERR_INVALID_ARG,
               "gemm_nt: a/w must be 16-byte aligned");
  MEVI_REQUIRE(act >= 0 && act <= 2, MEVI_ERR_INVALID_ARG, "gemm_nt: act must be 0 (none), 1 (relu) or 2 (erf gelu)");
  MEVI_REQUIRE(m < (1LL << 31) && n < (1LL << 31) && k < (1LL << 24), MEVI_ERR_UNSUPPORTED, "gemm_nt: too large");
  static const int skinny_on = [] { const char *e = getenv("MEVI_GEMM_SKINNY"); return e ? atoi(e) : 1; }();
  if (m * n <= SK_MAX_OUTPUTS && skinny_on) {   // a handful of rows: latency path
    hipLaunchKernelGGL(gemm_skinny_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((m + 3) / 4)), dim3(256), 0, stream, a, (long long)lda, w,
                       (long long)ldw, c, (long long)ldc, (int)m, (int)n, (int)k, bias, residual, (long long)ldr, act);
    MEVI_HIP_CHECK(hipGetLastError());
    return MEVI_OK;
  }
  const int64_t n_mpairs = (m + 2 * BM - 1) / (2 * BM);
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    n_cu = (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  // A 256x64 tile takes half the time of a 256x128 one (measured), so on small grids pick the width whose last,
  // partly filled round of workgroups wastes less: wide = 2 * ceil(T / CUs) half-rounds, narrow = ceil(2T / CUs).
  const int64_t t_wide = n_mpairs * ((n + 127) / 128);
  const bool narrow = n > 64 && t_wide < 8 * (int64_t)n_cu &&
                      (2 * t_wide + n_cu - 1) / n_cu < 2 * ((t_wide + n_cu - 1) / n_cu);
  const int64_t n_ntiles = narrow ? (n + 63) / 64 : (n + 127) / 128;
  MEVI_REQUIRE(n_mpairs * n_ntiles <= 0x7fffffffLL, MEVI_ERR_UNSUPPORTED, "gemm_nt: grid too large");
  const size_t lds_bytes = narrow ? pp_lds_bytes<1>() : pp_lds_bytes<2>();
  const bool ktail = (k % BK) != 0;
  const void *fn = narrow ? (ktail ? reinterpret_cast<const void *>(gemm_nt_kernel<1, true>)
                                   : reinterpret_cast<const void *>(gemm_nt_kernel<1, false>))
                          : (ktail ? reinterpret_cast<const void *>(gemm_nt_kernel<2, true>)
                                   : reinterpret_cast<const void *>(gemm_nt_kernel<2, false>));
  MEVI_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  long long lda_ = lda, ldw_ = ldw, ldc_ = ldc, ldr_ = ldr;
  int m_ = (int)m, n_ = (int)n, k_ = (int)k, nt = (int)n_ntiles, mp = (int)n_mpairs;
  void *args[] = {(void *)&a, &lda_, (void *)&w, &ldw_, (void *)&c, &ldc_, &m_, &n_, &k_, (void *)&bias,
                  (void *)&residual, &ldr_, &act, &nt, &mp};
  MEVI_HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)(n_mpairs * n_ntiles)), dim3(PP_THREADS), args, lds_bytes, stream));
  return MEVI_OK;
}
