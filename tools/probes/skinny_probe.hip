// Where does gemm_split_skinny_kernel's time go?  One wave per 32 x 32 outputs, K' = 3 x K/16 dependent MFMAs fed by
// 16-byte fragment loads straight from global memory.  Variants: full, no loads, no MFMA, coalesced-through-LDS staging.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/skinny_probe.hip -o /tmp/skinny_probe && /tmp/skinny_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(const _Float16 *A, const _Float16 *W, int kp, float *C, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lrow = lane & 31, half = lane >> 5;
  const int n0 = (blockIdx.x * 4 + wave) * 32;
  const _Float16 *pa = A + (size_t)lrow * 2 * kp + 8 * half;
  const _Float16 *pw = W + (size_t)(n0 + lrow) * 2 * kp + 8 * half;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  f16x8 one = {1, 1, 1, 1, 1, 1, 1, 1};
  f32x16 keep = acc;
#pragma unroll 4
  for (int kk = 0; kk < kp; kk += 16) {
    f16x8 ah = one, al = one, wh = one, wl = one;
    if (MODE != 1) {
      ah = *reinterpret_cast<const f16x8 *>(pa + kk);
      al = *reinterpret_cast<const f16x8 *>(pa + kp + kk);
      wh = *reinterpret_cast<const f16x8 *>(pw + kk);
      wl = *reinterpret_cast<const f16x8 *>(pw + kp + kk);
    }
    if (MODE != 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah, acc, 0, 0, 0);
    } else {
      keep[0] += (float)ah[0] + (float)al[1] + (float)wh[2] + (float)wl[3];
    }
  }
  if (MODE == 2) acc = keep;
  for (int r = 0; r < 16; ++r) C[(size_t)(lrow) * N + n0 + (r & 3) + 8 * (r >> 2) + 4 * half] = acc[r];
}

template <int MODE>
float run(const _Float16 *A, const _Float16 *W, int kp, float *C, int N) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(N / 128), dim3(256), 0, 0, A, W, kp, C, N);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep && ms < best) best = ms;
  }
  return best * 1e3f;
}

int main() {
  for (int kp : {768, 3072}) {
    for (int N : {768, 3072}) {
      _Float16 *A, *W;
      float *C;
      hipMalloc(&A, (size_t)32 * 2 * kp * 2);
      hipMalloc(&W, (size_t)N * 2 * kp * 2);
      hipMalloc(&C, (size_t)32 * N * 4);
      hipMemset(A, 0, (size_t)32 * 2 * kp * 2);
      hipMemset(W, 0, (size_t)N * 2 * kp * 2);
      printf("K %4d N %4d: full %7.1f us | no loads %7.1f us | no MFMA %7.1f us\n", kp, N, run<0>(A, W, kp, C, N), run<1>(A, W, kp, C, N),
             run<2>(A, W, kp, C, N));
      hipFree(A); hipFree(W); hipFree(C);
    }
  }
  return 0;
}
