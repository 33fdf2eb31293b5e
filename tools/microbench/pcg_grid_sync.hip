// What would ONE persistent launch per damped system pay for synchronisation?    Today a PCG iteration
// is three launches (point pass, camera pass, vector step); a persistent kernel would replace the three kernel boundaries
// by three grid-wide barriers and the table fill of the point pass (144 KB per workgroup) by a refresh of the 48 KB of
// camera vectors.  This probe runs the geometry of the C3 point pass -- G workgroups of 1024 threads, ~144 KB of LDS,
// one per compute unit -- through `iters` rounds of
//   (a) a counter barrier alone (ONE agent-scope release + relaxed polling by thread 0 per workgroup, one acquire),
//   (b) the barrier + every workgroup pulling a fresh 48 KB vector (written by 63 of the workgroups with write-through
//       stores just before the barrier) into LDS with L1-bypassing loads,
//   (c) an XCD-hierarchical barrier (per-XCD counters, leaders meet on a top counter) alone,
// and prints microseconds per round.  hipcc -O3 --offload-arch=gfx950 -o pcg_grid_sync pcg_grid_sync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr long long SPIN_TICKS = 20000000;      // 200 ms of the 100 MHz clock: bounded spins

__device__ inline bool wait_ge(const unsigned long long* w, unsigned long long target) {
  long long t0 = 0; int spins = 0;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023) == 0) { const long long now = (long long)wall_clock64(); if (!t0) t0 = now; else if (now - t0 > SPIN_TICKS) return false; }
  }
  return true;
}
// mode 0: flat counter barrier; 1: barrier + 48 KB refresh; 2: XCD-hierarchical barrier (blockIdx % 8 = XCD, speed only)
__global__ void __launch_bounds__(1024) k_sync(unsigned long long* ctr, unsigned long long* xctr, double* vec, int iters, int mode, int* err, long long* clk, double* sink) {
  extern __shared__ __align__(16) double lds[];
  __shared__ int ok;
  const int G = gridDim.x, g = blockIdx.x, x = g & 7;
  const int per_x = (G - x + 7) / 8;            // workgroups with blockIdx % 8 == x
  double acc = 0.0;
  const long long t0 = (long long)wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (mode == 1 && g < 63) {                  // the "vector step": 63 workgroups rewrite their slice of the 6000 doubles
      for (int i = threadIdx.x; i < 96; i += 1024) { const int e = g * 96 + i; if (e < 6000) __hip_atomic_store(vec + e, (double)(it + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      bool good = true;
      if (mode != 2) {
        __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        good = wait_ge(ctr, (unsigned long long)G * (it + 1));
      } else {
        const unsigned long long old = __hip_atomic_fetch_add(xctr + 16 * x, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == (unsigned long long)per_x * (it + 1)) {           // last of its XCD: meets the other leaders, then releases its XCD
          __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          good = wait_ge(ctr, 8ull * (it + 1));
          __hip_atomic_store(xctr + 16 * x + 8, (unsigned long long)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else good = wait_ge(xctr + 16 * x + 8, (unsigned long long)(it + 1));
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      ok = good ? 1 : 0;
    }
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0) *err = 1; return; }
    if (mode == 1) {                           // every workgroup: the fresh 48 KB into LDS (6 loads of 8 bytes per thread in flight)
      double v[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) { const int e = threadIdx.x + 1024 * q; v[q] = e < 6000 ? __hip_atomic_load(vec + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0; }
#pragma unroll
      for (int q = 0; q < 6; ++q) { const int e = threadIdx.x + 1024 * q; if (e < 6000) lds[e] = v[q]; acc += v[q]; }
    }
  }
  if (threadIdx.x == 0 && g == 0) clk[0] = (long long)wall_clock64() - t0;
  if (acc == -1.0) sink[0] = acc;
}

int main() {
  unsigned long long *ctr, *xctr; double *vec, *sink; int* err; long long* clk;
  hipMalloc(&ctr, 8); hipMalloc(&xctr, 8 * 16 * 8); hipMalloc(&vec, 6000 * 8); hipMalloc(&sink, 8); hipMalloc(&err, 4); hipMalloc(&clk, 8);
  hipFuncSetAttribute((const void*)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  const int iters = 500;
  const char* names[3] = {"flat counter barrier", "flat barrier + 48 KB refresh into LDS", "XCD-hierarchical barrier"};
  for (int G : {196, 256}) for (int mode = 0; mode < 3; ++mode) {
    hipMemset(ctr, 0, 8); hipMemset(xctr, 0, 8 * 16 * 8); hipMemset(err, 0, 4);
    hipLaunchKernelGGL(k_sync, dim3(G), dim3(1024), 144 * 1024, 0, ctr, xctr, vec, iters, mode, err, clk, sink);
    hipDeviceSynchronize();
    int herr; long long hclk;
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(&hclk, clk, 8, hipMemcpyDeviceToHost);
    printf("%3d workgroups x 1024 threads, 144 KB LDS: %-40s %6.2f us per round%s\n", G, names[mode], hclk * 0.01 / iters, herr ? "  (TIMED OUT)" : "");
  }
  return 0;
}
