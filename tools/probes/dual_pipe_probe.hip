// Can the f32 MFMA pipe and the f32 VALU (v_pk_fma_f32) pipe of a SIMD run at the same time?  Register-only loops:
// mode 0: every wave issues v_mfma_f32_32x32x2_f32; mode 1: every wave issues v_pk_fma_f32; mode 2: even waves MFMA,
// odd waves VALU (one of each per SIMD with 8 waves per workgroup).  Reports TFLOP/s of each kind.
//   hipcc --offload-arch=gfx950 -O3 -o dual_pipe_probe dual_pipe_probe.hip && ./dual_pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void probe(float *out, int iters, int mode) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool mfma = mode == 0 || (mode == 2 && (w & 1) == 0);
  float r = 0.f;
  if (mfma) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  } else {
    f32x2 acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = (f32x2){0.f, 0.f};
    f32x2 a = {threadIdx.x * 1e-3f, 0.5f}, b = {1.0f - a.x, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int i = 0; i < 32; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      }
    }
    for (int i = 0; i < 32; ++i) r += acc[i].x + acc[i].y;
  }
  if (r == 123.456f) out[0] = r;
}

int main() {
  float *out; hipMalloc(&out, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int grid = 256 * 4, iters = 20000;
  for (int mode = 0; mode < 3; ++mode) {
    probe<<<grid, 512>>>(out, 10, mode);
    hipEventRecord(a);
    probe<<<grid, 512>>>(out, iters, mode);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double wm = mode == 0 ? 8 : mode == 1 ? 0 : 4, wv = 8 - wm;
    double fm = (double)grid * wm * iters * 16 * (2.0 * 32 * 32 * 2);       // 16 MFMAs per iteration, 4096 FLOP each
    double fv = (double)grid * wv * iters * 128 * (64 * 2 * 2.0);            // 128 pk_fma per iteration, 256 FLOP each
    printf("mode %d: %.2f ms  MFMA %.1f TFLOP/s  VALU %.1f TFLOP/s  sum %.1f\n", mode, ms, fm / ms / 1e9, fv / ms / 1e9,
           (fm + fv) / ms / 1e9);
  }
  return 0;
}
