// Micro-probe: sustained issue rate of v_mfma_f32_32x32x2_f32 in the patterns the
// dense filter kernel uses.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_probe.hip -o tools/build/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float *out, int iters, const float *in) {
  __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 36];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  for (int i = t; i < 2 * 256 * 36; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float av0[4] = {in[t], in[t + 1], in[t + 2], in[t + 3]}, av1[4] = {in[t + 4], in[t + 5], in[t + 6], in[t + 7]};
  float bv0[4] = {in[t + 8], in[t + 9], in[t + 10], in[t + 11]}, bv1[4] = {in[t + 12], in[t + 13], in[t + 14], in[t + 15]};
  const float *pa = lds + ((wave >> 1) * 64 + (lane & 31)) * 36 + 16 * (lane >> 5);
  const float *pb = lds + (128 + (wave & 1) * 64 + (lane & 31)) * 36 + 16 * (lane >> 5);
  for (int it = 0; it < iters; ++it) {
    const int boff = (it & 1) * 256 * 36;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      if (MODE >= 1) {
        float4 a0 = *reinterpret_cast<const float4 *>(pa + boff + 4 * jj);
        float4 a1 = *reinterpret_cast<const float4 *>(pa + boff + 32 * 36 + 4 * jj);
        float4 b0 = *reinterpret_cast<const float4 *>(pb + boff + 4 * jj);
        float4 b1 = *reinterpret_cast<const float4 *>(pb + boff + 32 * 36 + 4 * jj);
        av0[0] = a0.x; av0[1] = a0.y; av0[2] = a0.z; av0[3] = a0.w;
        av1[0] = a1.x; av1[1] = a1.y; av1[2] = a1.z; av1[3] = a1.w;
        bv0[0] = b0.x; bv0[1] = b0.y; bv0[2] = b0.z; bv0[3] = b0.w;
        bv1[0] = b1.x; bv1[1] = b1.y; bv1[2] = b1.z; bv1[3] = b1.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv0[e], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], bv1[e], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv0[e], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], bv1[e], acc[1][1], 0, 0, 0);
      }
    }
    if (MODE >= 2) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + t] = s;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, float *out, const float *in) {
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, 10, in);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, in);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double mf = (double)grid * 4 * iters * 64;           // MFMAs
  double tf = mf * 32 * 32 * 2 * 2 / (ms * 1e-3) / 1e12;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64 * blocks_per_cu); // cycles per MFMA per SIMD at 2.4GHz
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  %.1f cyc/MFMA/SIMD\n", name, blocks_per_cu, ms, tf, cyc);
}

int main() {
  float *out, *in;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&in, 4096 * 4);
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("regs only", 1, out, in);
    run<0>("regs only", 2, out, in);
    run<1>("lds reads", 1, out, in);
    run<1>("lds reads", 2, out, in);
    run<2>("lds reads + barrier/64", 1, out, in);
    run<2>("lds reads + barrier/64", 2, out, in);
  }
  return 0;
}
