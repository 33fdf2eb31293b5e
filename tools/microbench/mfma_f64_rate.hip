// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: one wave, N instructions over 1 or 3 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void rate(double a, double b, long long* out, double* sink, int iters) {
  d4 acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = (d4){0, 0, 0, 0};
  a += threadIdx.x; b -= threadIdx.x;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
  long long* d; double* sink; long long h[8];
  hipMalloc(&d, 64); hipMalloc(&sink, 8 * 1024 * 8);
  for (int waves = 1; waves <= 8; waves *= 2) {
    rate<1><<<1, 64 * waves>>>(1.0, 2.0, d, sink, 256); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s), 1 accumulator : %.1f shader-clock ticks per MFMA (per wave)\n", waves, h[0] / 1024.0);
    rate<3><<<1, 64 * waves>>>(1.0, 2.0, d, sink, 256); hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("%d wave(s), 3 accumulators: %.1f shader-clock ticks per MFMA (per wave)\n", waves, h[0] / 3072.0);
  }
  return 0;
}
