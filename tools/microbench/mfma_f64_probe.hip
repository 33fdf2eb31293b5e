// Probe of v_mfma_f64_16x16x4_f64's operand / result layout on gfx950 (for csrc/ba_small.hpp's dense Schur product).
// D (16x16) = A (16x4) * B (4x16): which lane holds which element?  A[i][k] = 100 i + k, B[k][j] = (k == kk && j == jj).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(const double* A /*16x4*/, const double* B /*4x16*/, double* D /*64 lanes x 4*/) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];        // hypothesis: lane l -> A[i = l % 16][k = l / 16]
  const double b = B[(l / 16) * 16 + (l % 16)];       // hypothesis: lane l -> B[k = l / 16][j = l % 16]
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) hA[i * 4 + k] = 1 + i + 0.1 * k;
  for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) hB[k * 16 + j] = 1 + 0.01 * j + 10 * k;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  // hypothesis for D: lane l, register r -> D[i = 4 * r + l / 16][j = l % 16]
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + (l / 16), j = l % 16;
    if (fabs(hD[l * 4 + r] - ref[i * 16 + j]) > 1e-9) ++bad;
  }
  printf("layout hypothesis (A: i=l%%16,k=l/16; B: k=l/16,j=l%%16; D: i=4*r+l/16, j=l%%16): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  if (bad) {   // brute force: find for lane 0..3 / reg where the values are
    for (int l = 0; l < 64; l += 7) for (int r = 0; r < 4; ++r) {
      for (int q = 0; q < 256; ++q) if (fabs(hD[l * 4 + r] - ref[q]) < 1e-9) printf("lane %d reg %d = D[%d][%d]\n", l, r, q / 16, q % 16);
    }
  }
  return bad != 0;
}
