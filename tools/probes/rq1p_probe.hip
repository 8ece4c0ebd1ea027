// Round 5 feasibility probe for a ONE-PASS RQ shortlist kernel at (3, 256): every wave holds 32 rows x 768 centroids = 24 MFMA
// tiles of 32 x 32 (384 accumulators: one wave per SIMD), a workgroup = 4 waves = 128 rows; per 32-k unit the whole centroid slab
// (768 rows x 64 B = 48 KiB, L2-resident image) and the rows' f32 slab (128 x 128 B, HBM) arrive by LDS-DMA one unit ahead.
// No epilogue, synthetic data: the question is what a unit costs (the XDEEP kernel of the tree: ~1.25 us per 32 rows x 256
// centroids x 8 waves in its later passes, ~3.2 us in pass 0) and whether 384 accumulators + operands fit 512 registers.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/rq1p_probe.hip -o tools/probes/rq1p_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TA = 24;                 // MFMA tiles per wave (3 levels x 256 centroids / 32)
constexpr int AROWS = TA * 32;         // 768 image rows per unit
constexpr int AFL = AROWS * 16;        // floats of a centroid unit (64 B per row): 48 KiB
constexpr int XFL = 128 * 32;          // floats of an x unit (128 rows x 128 B): 16 KiB
constexpr int U = 24;                  // units per row tile (dim 768)
constexpr int LDS_FLOATS = 2 * AFL + 3 * XFL;   // centroid ring of 2, x ring of 3: 144 KiB

template <int HS, bool CONVERT>
__global__ __launch_bounds__(256, 1) void rq1p_kernel(const _Float16 *__restrict__ img, const float *__restrict__ X, long long n_tiles,
                                                      float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(img), 0, U * AROWS * 64, 0x00020000);
  // centroid piece: 16 rows x 64 B per wave instruction; wave w moves pieces w, w + 4, ... (12 per unit)
  const int voff_a = (lane >> 2) * 64 + (((lane & 3) ^ (((lane >> 2) >> 2) & 3)) << 4);   // slot s of a row holds piece s ^ key(row)
  // x piece: 8 rows x 128 B; wave w its own 32 rows (4 pieces)
  const int voff_x = (lane >> 3) * 3072 + (((lane & 7) ^ (((lane >> 3) >> 1) & 7)) << 4);  // (rows 8 i + r8: key (row >> 1) & 7 = (r8 >> 1) + 4 (i & 1): see below)
  const int voff_x1 = (lane >> 3) * 3072 + (((lane & 7) ^ ((((lane >> 3) >> 1) + 4) & 7)) << 4);
  const int a_sw = (lrow >> 2) & 3, b_sw = (lrow >> 1) & 7;
  f32x16 acc[TA];
  float sink = 0.f;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X) + (size_t)tile * 128 * 768, 0, 128 * 3072, 0x00020000);
    auto dma = [&](int u) {
      float *da = lds + (u & 1) * AFL;
#pragma unroll
      for (int i = 0; i < 12; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void *)(da + (16 * (w + 4 * i)) * 16), 16, voff_a,
                                                 u * AROWS * 64 + (w + 4 * i) * 1024, 0, 0);
      float *dx = lds + 2 * AFL + (u % 3) * XFL;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(dx + (32 * w + 8 * i) * 32), 16,
                                                 (i & 1) ? voff_x1 : voff_x, u * 128 + (32 * w + 8 * i) * 3072, 0, 0);
    };
#pragma unroll
    for (int ti = 0; ti < TA; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;
    dma(0);
    for (int u = 0; u < U; ++u) {
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // unit u landed; everybody is done with unit u - 1
      if (u + 1 < U) dma(u + 1);
      const float *ua = lds + (u & 1) * AFL + lrow * 16;
      const float *ux = lds + 2 * AFL + (u % 3) * XFL + (32 * w + lrow) * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f16x8 b;
        if (CONVERT) {
          const int c0 = 4 * j + 2 * half;
          const float4 x0 = *reinterpret_cast<const float4 *>(ux + ((c0 ^ b_sw) << 2));
          const float4 x1 = *reinterpret_cast<const float4 *>(ux + (((c0 + 1) ^ b_sw) << 2));
          b[0] = (_Float16)(x0.x * 0.5f), b[1] = (_Float16)(x0.y * 0.5f), b[2] = (_Float16)(x0.z * 0.5f), b[3] = (_Float16)(x0.w * 0.5f);
          b[4] = (_Float16)(x1.x * 0.5f), b[5] = (_Float16)(x1.y * 0.5f), b[6] = (_Float16)(x1.z * 0.5f), b[7] = (_Float16)(x1.w * 0.5f);
        } else {
          b = *reinterpret_cast<const f16x8 *>(ux + 8 * j + 4 * half);
        }
        // HS tiles per step: fragments of step s + 1 read under the MFMAs of step s
        f16x8 fa[2][HS];
#pragma unroll
        for (int i = 0; i < HS; ++i) fa[0][i] = *reinterpret_cast<const f16x8 *>(ua + 32 * i * 16 + (((2 * j + half) ^ a_sw) << 2));
#pragma unroll
        for (int s = 0; s < TA / HS; ++s) {
          if (s + 1 < TA / HS) {
#pragma unroll
            for (int i = 0; i < HS; ++i)
              fa[(s + 1) & 1][i] = *reinterpret_cast<const f16x8 *>(ua + 32 * ((s + 1) * HS + i) * 16 + (((2 * j + half) ^ a_sw) << 2));
          }
#pragma unroll
          for (int i = 0; i < HS; ++i) acc[s * HS + i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][i], b, acc[s * HS + i], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int ti = 0; ti < TA; ++ti)
#pragma unroll
      for (int r = 0; r < 16; ++r) sink += acc[ti][r];
  }
  out[blockIdx.x * 256 + t] = sink;
}

template <int HS, bool CONVERT>
static int run(const _Float16 *img, const float *X, long long n_tiles, float *out) {
  const void *fn = reinterpret_cast<const void *>(rq1p_kernel<HS, CONVERT>);
  CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    rq1p_kernel<HS, CONVERT><<<256, 256, LDS_FLOATS * 4>>>(img, X, n_tiles, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  const double units = (double)((n_tiles + 255) / 256) * U;
  printf("steps of %d tiles, %s: %8.2f ms for %lld tiles of 128 rows = %.2f us per unit and CU (%.0f TFLOP/s f16)\n", HS,
         CONVERT ? "f32 rows converted" : "f16 rows as they are", best, n_tiles, best * 1e3 / units,
         2.0 * n_tiles * 128 * 768.0 * 768.0 / best / 1e9);
  return 0;
}

int main(int argc, char **argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 8841823;
  const long long n_tiles = (n + 127) / 128;
  _Float16 *img;
  float *X, *out;
  CK(hipMalloc(&img, (size_t)U * AROWS * 64));
  CK(hipMalloc(&X, (size_t)n_tiles * 128 * 768 * 4));
  CK(hipMalloc(&out, 256 * 256 * 4));
  CK(hipMemset(img, 0x11, (size_t)U * AROWS * 64));
  CK(hipMemset(X, 0, (size_t)n_tiles * 128 * 768 * 4));
  if (run<4, true>(img, X, n_tiles, out) || run<6, true>(img, X, n_tiles, out) || run<4, false>(img, X, n_tiles, out)) return 1;
  return 0;
}
