// Round 6 feasibility probe: a ONE-PASS RQ shortlist loop at (3, 256) in which WAVES OWN LEVELS.
//
// The struck one-pass probe (rq1p_probe.hip) put all 768 centroid columns on one wave (384 accumulators -> spills).  Here a
// workgroup of 12 waves (3 per SIMD) shares one 128-row tile; a wave owns ONE level's columns for a block of the rows, so it
// carries 128 accumulators like the production kernel's waves:
//   SHAPE 0:  wave = 32 rows x 256 centroids of its level   (8 MFMA tiles; per 16-k step 8 A-fragment reads + 1 x fragment)
//   SHAPE 1:  wave = 64 rows x 128 centroids of its level   (2 x 4 tiles;  per 16-k step 4 A-fragment reads + 2 x fragments)
// x crosses HBM ONCE (f32 rows by LDS-DMA, converted f32 -> f16 between the LDS read and the MFMA by every wave that needs the
// fragment: VALU work that hides under the other two waves of the SIMD), no fragment scratch.  The centroid image (768 columns x
// 768 k, f16, 1.2 MB: L2 resident) streams through LDS in units of 16 k (24 KiB) -- every workgroup reads all of it per tile:
// 69 077 tiles x 1.18 MB = 81 GB of L2 -> LDS traffic per corpus.
// Two DMA streams with their own in-order counters: waves 0-7 stage the centroid units (ring of NCB), waves 8-11 the x units of
// 32 k (ring of NXB); one barrier per 16-k step.  No epilogue (the level chain is table look-ups on finished accumulators), a
// token reduction keeps the accumulators alive.  The question: what does the loop cost per corpus (go if <= 10 ms)?
//     hipcc --offload-arch=gfx950 -O3 tools/probes/rq_wol_probe.hip -o tools/probes/rq_wol_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ROWS = 128, DIM = 768, NCENT = 768;
constexpr int CU_PER_TILE = DIM / 16;      // 48 centroid units of 16 k
constexpr int CFL = NCENT * 8;             // floats of a centroid unit: 768 image rows x 32 B = 24 KiB
constexpr int XFL = ROWS * 32;             // floats of an x unit: 128 rows x 128 B = 16 KiB

template <int NCB, int NXB>
constexpr size_t lds_bytes() { return (size_t)(NCB * CFL + NXB * XFL + DIM) * 4; }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ABL (ablations, timing only): 1 no centroid DMA, 2 no x DMA, 4 no A-fragment LDS reads, 8 no MFMA, 16 no barrier,
// 32 swizzled A-fragment rows (slot half ^ ((row >> 4) & 1): a ds_read_b128 lane group {0-3, 12-15, 20-27} then covers all 64 banks)
template <int NCB, int NXB, int SHAPE, bool CONVERT, int ABL = 0>
__global__ __launch_bounds__(768, 1) void rq_wol_kernel(const _Float16 *__restrict__ img, const float *__restrict__ X, long long n_tiles,
                                                        const float *__restrict__ mu, float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *cring = lds, *xring = lds + NCB * CFL, *mus = xring + NXB * XFL;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lrow = lane & 31, half = lane >> 5;
  const int lv = w >> 2, rem = w & 3;
  // SHAPE 0: row group rem (32 rows), tiles 8 lv .. + 7.   SHAPE 1: row group rem >> 1 (64 rows), tiles 8 lv + 4 (rem & 1) .. + 3
  const int row0 = SHAPE == 0 ? 32 * rem : 64 * (rem >> 1);
  const int tile0 = SHAPE == 0 ? 8 * lv : 8 * lv + 4 * (rem & 1);
  for (int k = t; k < DIM; k += 768) mus[k] = mu[k];
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(img), 0, CU_PER_TILE * CFL * 4, 0x00020000);
  const bool cwave = w < 8;
  const int voff_c = (ABL & 32) ? ((lane >> 1) * 32 + ((((lane & 1) ^ ((lane >> 5) & 1))) << 4)) : lane * 16;
  const long long my_tiles = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  const long long G = my_tiles * CU_PER_TILE;        // 16-k steps of this workgroup, streamed ACROSS its tiles
  // x DMA: wave 8 + xw, instruction i -> block b = xw + 4 i (8 rows x 128 B); LDS slot s of row r holds logical piece s ^ key(r),
  // key(r) = (r >> 1) & 7 (conflict-free 16-byte reads of 32 rows 128 B apart): the swizzle goes on the SOURCE address
  const int xw = w - 8;
  int voff_x[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 8 * (xw + 4 * i) + (lane >> 3);
    voff_x[i] = (lane >> 3) * (DIM * 4) + (((lane & 7) ^ ((r >> 1) & 7)) << 4);
  }
  auto dma_c = [&](long long g) {        // centroid unit g % 48 -> ring slot g % NCB      (waves 0-7, three 1-KiB pieces each)
    if (ABL & 1) return;
    const int v = (int)(g % CU_PER_TILE);
    float *dst = cring + (int)(g % NCB) * CFL;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void *)(dst + (w + 8 * i) * 256), 16, voff_c,
                                               v * (CFL * 4) + (w + 8 * i) * 1024, 0, 0);
  };
  auto dma_x = [&](long long gx) {       // x unit gx (32 k of one tile's 128 rows) -> ring slot gx % NXB      (waves 8-11, four pieces each)
    if (ABL & 2) return;
    const long long tile = blockIdx.x + (gx / (DIM / 32)) * gridDim.x;
    const int u = (int)(gx % (DIM / 32));
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X) + (size_t)tile * ROWS * DIM, 0, ROWS * DIM * 4, 0x00020000);
    float *dst = xring + (int)(gx % NXB) * XFL;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (__attribute__((address_space(3))) void *)(dst + (xw + 4 * i) * 256), 16, voff_x[i],
                                               u * 128 + (xw + 4 * i) * 8 * (DIM * 4), 0, 0);
  };
  const long long GX = my_tiles * (DIM / 32);
  if (cwave) {
#pragma unroll
    for (int g = 0; g < NCB - 1; ++g)
      if (g < G) dma_c(g);
  } else {
#pragma unroll
    for (int g = 0; g < NXB - 1; ++g)
      if (g < GX) dma_x(g);
  }
  constexpr int NT = SHAPE == 0 ? 8 : 4, NR = SHAPE == 0 ? 1 : 2;
  f32x16 acc[NR][NT];
  float sink = 0.f;
  const int key = (lrow >> 1) & 7;       // rows row0 + 32 rb + lrow: row0 and 32 rb are multiples of 16, the key is lrow's
  for (long long g = 0; g < G; ++g) {
    const int v = (int)(g % CU_PER_TILE);
    // unit g (and, on even steps, x unit g / 2) landed; everybody is done with step g - 1
    if (cwave) { if (!(ABL & 1)) wait_vm<3 * (NCB - 2)>(); }
    else if ((g & 1) == 0) { if (!(ABL & 2)) wait_vm<4 * (NXB - 2)>(); }
    if (!(ABL & 16)) asm volatile("s_barrier" ::: "memory");
    if (cwave) {
      if (g + NCB - 1 < G) dma_c(g + NCB - 1);
      else if (!(ABL & 1)) {                   // keep the counter's arithmetic: a piece that lands in the slot just freed
        float *dst = cring + (int)((g + NCB - 1) % NCB) * CFL;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rc, (__attribute__((address_space(3))) void *)(dst + (w + 8 * i) * 256), 16, lane * 16, 0, 0, 0);
      }
    } else if ((g & 1) == 0) {
      const long long gx = (g >> 1) + NXB - 1;
      dma_x(gx < GX ? gx : GX - 1);            // (past the end: the last unit again, into the slot just freed)
    }
    if (v == 0) {
#pragma unroll
      for (int rb = 0; rb < NR; ++rb)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
    }
    const float *cb = cring + (int)(g % NCB) * CFL + (32 * tile0 + lrow) * 8 + ((ABL & 32) ? (half ^ ((lrow >> 4) & 1)) : half) * 4;
    const float *xb = xring + (int)((g >> 1) % NXB) * XFL;
    const int c0 = 4 * (v & 1) + 2 * half;    // logical 16-byte pieces c0, c0 + 1 of the row's 128-byte unit = k 16 (v & 1) + 8 half .. + 7
    f16x8 b[NR];
#pragma unroll
    for (int rb = 0; rb < NR; ++rb) {
      const float *xr = xb + (row0 + 32 * rb + lrow) * 32;
      const float4 x0 = *reinterpret_cast<const float4 *>(xr + ((c0 ^ key) << 2));
      if (CONVERT) {
        const float4 x1 = *reinterpret_cast<const float4 *>(xr + (((c0 + 1) ^ key) << 2));
        const float4 m0 = *reinterpret_cast<const float4 *>(mus + 16 * v + 8 * half);
        const float4 m1 = *reinterpret_cast<const float4 *>(mus + 16 * v + 8 * half + 4);
        const float S = 64.f;
        b[rb][0] = (_Float16)fmaf(x0.x, S, -m0.x), b[rb][1] = (_Float16)fmaf(x0.y, S, -m0.y);
        b[rb][2] = (_Float16)fmaf(x0.z, S, -m0.z), b[rb][3] = (_Float16)fmaf(x0.w, S, -m0.w);
        b[rb][4] = (_Float16)fmaf(x1.x, S, -m1.x), b[rb][5] = (_Float16)fmaf(x1.y, S, -m1.y);
        b[rb][6] = (_Float16)fmaf(x1.z, S, -m1.z), b[rb][7] = (_Float16)fmaf(x1.w, S, -m1.w);
      } else {
        b[rb] = *reinterpret_cast<const f16x8 *>(&x0);      // ablation: the loop without the second read and the conversion
      }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      f16x8 a;
      if (ABL & 4) a = b[0];
      else a = *reinterpret_cast<const f16x8 *>(cb + 32 * i * 8);
#pragma unroll
      for (int rb = 0; rb < NR; ++rb) {
        if (ABL & 8) acc[rb][i][0] += (float)a[0] * (float)b[rb][1];
        else acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[rb], acc[rb][i], 0, 0, 0);
      }
    }
    if (v == CU_PER_TILE - 1) {             // token epilogue: one value per accumulator set
#pragma unroll
      for (int rb = 0; rb < NR; ++rb)
#pragma unroll
        for (int i = 0; i < NT; ++i) sink += acc[rb][i][0] + acc[rb][i][7] + acc[rb][i][15];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink == 1234.5f) out[t] = sink;
  else if (blockIdx.x == 0 && t < 768) out[t] = sink;
}

__global__ void fill_f32(float *p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 65536.f - 0.5f) * 0.2f;
  }
}
__global__ void fill_f16(_Float16 *p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (_Float16)(((float)(h & 0xffff) / 65536.f - 0.5f) * 4.f);
  }
}

template <int NCB, int NXB, int SHAPE, bool CONVERT, int ABL = 0>
int run(const char *name, const _Float16 *img, const float *X, long long n_tiles, const float *mu, float *out, int grid) {
  auto fn = rq_wol_kernel<NCB, NXB, SHAPE, CONVERT, ABL>;
  const size_t lds = lds_bytes<NCB, NXB>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(grid), dim3(768), lds, 0, img, X, n_tiles, mu, out);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(768), lds, 0, img, X, n_tiles, mu, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double steps = (double)((n_tiles + grid - 1) / grid) * CU_PER_TILE;
  const double flop = 2.0 * (double)n_tiles * ROWS * NCENT * DIM;
  printf("%-64s %7.2f ms for %lld tiles of 128 rows = %.3f us per 16-k step and CU (%.0f TFLOP/s f16, x at %.2f TB/s, L2->LDS %.1f TB/s), LDS %zu KiB\n",
         name, best, n_tiles, best * 1e3 / steps, flop / best / 1e9, (double)n_tiles * ROWS * DIM * 4 / best / 1e9,
         (double)n_tiles * CU_PER_TILE * CFL * 4 / best / 1e9, lds / 1024);
  return 0;
}

int main(int argc, char **argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 8841823LL;
  const long long n_tiles = rows / ROWS;
  int dev = 0, n_cu = 256;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
  _Float16 *img;
  float *X, *mu, *out;
  CK(hipMalloc(&img, (size_t)CU_PER_TILE * CFL * 4));
  CK(hipMalloc(&X, (size_t)n_tiles * ROWS * DIM * 4));
  CK(hipMalloc(&mu, DIM * 4));
  CK(hipMalloc(&out, 768 * 4));
  hipLaunchKernelGGL(fill_f16, dim3(256), dim3(256), 0, 0, img, (size_t)CU_PER_TILE * CFL * 2, 7u);
  hipLaunchKernelGGL(fill_f32, dim3(4096), dim3(256), 0, 0, X, (size_t)n_tiles * ROWS * DIM, 11u);
  hipLaunchKernelGGL(fill_f32, dim3(4), dim3(256), 0, 0, mu, (size_t)DIM, 13u);
  CK(hipDeviceSynchronize());
  printf("rq_wol_probe: %lld rows, %d CUs, one 768-thread workgroup per CU\n", rows, n_cu);
  if (run<4, 3, 0, true>("32 rows x 256 centroids per wave, rings C4 X3, f32 rows converted", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 0, true>("32 rows x 256 centroids per wave, rings C3 X4, f32 rows converted", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<4, 3, 1, true>("64 rows x 128 centroids per wave, rings C4 X3, f32 rows converted", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true>("64 rows x 128 centroids per wave, rings C3 X4, f32 rows converted", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<4, 3, 1, false>("64 rows x 128 centroids per wave, rings C4 X3, no conversion (ablation)", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<4, 3, 0, false>("32 rows x 256 centroids per wave, rings C4 X3, no conversion (ablation)", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 32>("64 x 128, C3 X4, A rows swizzled", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 0, true, 32>("32 x 256, C3 X4, A rows swizzled", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 1>("64 x 128, C3 X4, ablation: no centroid DMA", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 2>("64 x 128, C3 X4, ablation: no x DMA", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 3>("64 x 128, C3 X4, ablation: no DMA at all", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 4>("64 x 128, C3 X4, ablation: no A-fragment LDS reads", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 8>("64 x 128, C3 X4, ablation: no MFMA", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 16>("64 x 128, C3 X4, ablation: no barrier", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 19>("64 x 128, C3 X4, ablation: no DMA, no barrier", img, X, n_tiles, mu, out, n_cu)) return 1;
  if (run<3, 4, 1, true, 31>("64 x 128, C3 X4, ablation: nothing but x LDS reads + conversion", img, X, n_tiles, mu, out, n_cu)) return 1;
  return 0;
}
