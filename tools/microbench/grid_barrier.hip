// Cost of a software barrier among a few co-resident workgroups (agent-scope atomics on one counter word), and of handing
// a small array from every workgroup to every other one across it.  hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline bool barrier_wait(unsigned long long* ctr, unsigned long long target, int* err) {
  // thread 0 of the workgroup arrives and spins (bounded); everybody else waits at the workgroup barrier
  __shared__ int ok;
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    ok = 1;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { ok = 0; *err = 1; break; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // workgroup-scope is enough for the data: it is read with agent-scope loads
  }
  __syncthreads();
  return ok != 0;
}

// stride: only blocks with blockIdx % stride == 0 take part (stride 8: all on one XCD if blockIdx % 8 = XCD)
__global__ void __launch_bounds__(512) k_barriers(unsigned long long* ctr, double* slots, int nwg, int stride, int iters, int payload,
                                                  int* err, long long* out_clk, double* out_sum) {
  if (blockIdx.x % stride) return;
  const int g = blockIdx.x / stride;
  if (g >= nwg) return;
  double acc = 0.0;
  const long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    // every workgroup writes `payload` doubles into its slot (agent-scope stores), barrier, then reads all slots
    double* mine = slots + ((size_t)(it & 1) * nwg + g) * payload;
    for (int i = threadIdx.x; i < payload; i += blockDim.x)
      __hip_atomic_store(mine + i, (double)(it + g + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!barrier_wait(ctr, (unsigned long long)nwg * (it + 1), err)) return;
    const double* all = slots + (size_t)(it & 1) * nwg * payload;
    for (int i = threadIdx.x; i < nwg * payload; i += blockDim.x)
      acc += __hip_atomic_load(all + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out_clk[g] = t1 - t0; }
  atomicAdd(out_sum, acc);
}

int main(int argc, char** argv) {
  const int iters = 200;
  unsigned long long* ctr; double* slots; int* err; long long* clk; double* sum;
  hipMalloc(&ctr, 8); hipMalloc(&slots, 2 * 64 * 4096 * 8); hipMalloc(&err, 4); hipMalloc(&clk, 64 * 8); hipMalloc(&sum, 8);
  for (int stride : {1, 8}) for (int nwg : {2, 4, 8, 16}) for (int payload : {0, 64, 1024, 2304}) {
    hipMemset(ctr, 0, 8); hipMemset(err, 0, 4); hipMemset(sum, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_barriers, dim3(nwg * stride), dim3(512), 0, 0, ctr, slots, nwg, stride, iters, payload, err, clk, sum);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int herr; long long hclk[64]; double hs;
    hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost); hipMemcpy(hclk, clk, nwg * 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, sum, 8, hipMemcpyDeviceToHost);
    printf("stride %d  %2d workgroups  payload %4d doubles each: %6.2f us per round (in-kernel clock of workgroup 0: %6.2f us)%s\n", stride, nwg, payload,
           ms * 1e3 / iters, hclk[0] * 0.01 / iters, herr ? "  TIMED OUT" : "");
  }
  return 0;
}
