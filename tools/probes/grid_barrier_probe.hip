// Round 5: what a grid-wide hand-off costs inside ONE persistent kernel on the 8 XCDs of an MI355X, against the >= 4.7 us a
// dependent launch costs in graph replay (profiles/r05_latency.txt).  One workgroup per CU; every round each workgroup writes a
// word, all meet at a barrier built from one agent-scope counter + agent-scope release / acquire fences (the L2s of the XCDs
// are not coherent with each other without them), then each reads the word of a workgroup on ANOTHER XCD and checks it.
//     hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o tools/probes/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// MODE 0: counter + spin on the counter itself; 1: counter + separate generation flag written by the last arrival;
// 2: like 0 without the fences (what the synchronisation alone costs: NOT a valid hand-off)
template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned int *ctr, unsigned int *flag, unsigned int nwg, unsigned int &gen) {
  __syncthreads();
  ++gen;
  if (threadIdx.x == 0) {
    if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned int prev = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) {
      if (prev == gen * nwg - 1u) __hip_atomic_store(flag, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) __builtin_amdgcn_s_sleep(1);
    } else {
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * nwg) __builtin_amdgcn_s_sleep(1);
    }
    if (MODE != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned int *ctr, unsigned int *flag, unsigned int *words, int rounds,
                                                     int payload, unsigned int *bad) {
  unsigned int gen = 0;
  const unsigned int nwg = gridDim.x;
  unsigned int wrong = 0;
  for (int r = 0; r < rounds; ++r) {
    // the "stage": every workgroup writes `payload` words (plain stores), thread 0 the one that is checked
    for (int i = threadIdx.x; i < payload; i += blockDim.x) words[(size_t)blockIdx.x * payload + i] = (unsigned int)(r * 7919 + blockIdx.x + i);
    grid_barrier<MODE>(ctr, flag, nwg, gen);
    // read a word of a workgroup 3 XCDs further (workgroups go round-robin over the XCDs)
    const unsigned int other = (blockIdx.x + 3u + 8u * (threadIdx.x & 7)) % nwg;
    const int i = payload ? (int)(threadIdx.x % (unsigned)payload) : 0;
    if (payload) {
      const unsigned int v = words[(size_t)other * payload + i];
      if (v != (unsigned int)(r * 7919 + other + i)) ++wrong;
    }
    grid_barrier<MODE>(ctr, flag, nwg, gen);   // (the words are overwritten next round)
  }
  if (wrong) atomicAdd(bad, wrong);
}

__global__ void empty_kernel(unsigned int *p) { if (p == nullptr) __builtin_trap(); }

template <int MODE>
static int run(const char *name, int nwg, int rounds, int payload, unsigned int *ctr, unsigned int *flag, unsigned int *words,
               unsigned int *bad) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  float best = 1e30f;
  unsigned int hbad = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(ctr, 0, 4));
    CK(hipMemset(flag, 0, 4));
    CK(hipMemset(bad, 0, 4));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    rounds_kernel<MODE><<<nwg, 256>>>(ctr, flag, words, rounds, payload, bad);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
    unsigned int h;
    CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
    hbad += h;
  }
  printf("%-44s %3d workgroups, payload %5d words: %6.2f us per barrier (2 per round), stale reads %u\n", name, nwg, payload,
         best * 1e3f / (2.f * rounds), hbad);
  return 0;
}

int main() {
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount;
  unsigned int *ctr, *flag, *words, *bad;
  CK(hipMalloc(&ctr, 256));
  CK(hipMalloc(&flag, 256));
  CK(hipMalloc(&bad, 256));
  CK(hipMalloc(&words, (size_t)ncu * 16384 * 4));
  printf("%s: %d CUs\n", pr.name, ncu);
  const int rounds = 2000;
  for (int nwg : {ncu, ncu / 2, 64, 32, 8}) {
    for (int payload : {1, 1024, 16384}) {
      if (run<0>("counter, spin on the counter", nwg, rounds, payload, ctr, flag, words, bad)) return 1;
      if (run<1>("counter + generation flag", nwg, rounds, payload, ctr, flag, words, bad)) return 1;
    }
    if (run<2>("counter only, NO fences (not a hand-off)", nwg, rounds, 0, ctr, flag, words, bad)) return 1;
  }
  // the other side of the comparison on this box: back-to-back dependent launches of an empty kernel, eager and graph-replayed
  {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a, s));
      for (int i = 0; i < n; ++i) empty_kernel<<<ncu, 256, 0, s>>>(ctr);
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
    }
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("eager: %d dependent empty launches: %.2f us each\n", n, ms * 1e3f / n);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) empty_kernel<<<ncu, 256, 0, s>>>(ctr);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a, s));
      for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(b, s));
      CK(hipEventSynchronize(b));
    }
    CK(hipEventElapsedTime(&ms, a, b));
    printf("graph replay: 200 dependent empty launches per graph: %.2f us each\n", ms * 1e3f / 2000);
  }
  return 0;
}
