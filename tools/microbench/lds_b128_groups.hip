// Which lanes of a wave share an LDS issue group for ds_read_b128?  The point passes read a camera's 144-byte row
// (36-dword stride) with nine ds_read_b128; the host orders each point's observations so that the lanes ASSUMED to be
// served together ({0-3,12-15,20-27} / {4-11,16-19,28-31} of each 32-lane half) fall into distinct bank classes
// (row mod 16).  This probe measures that assumption: a wave reads rows row[lane] (nine b128 reads per repetition) and
//   (1) times whole patterns: rows = lane (distinct classes everywhere), random rows, rows with distinct classes
//       inside each assumed group;
//   (2) starts from rows = lane and lets lane b read a DIFFERENT row of lane a's class, for every pair (a, b): a pair that
//       got slower shares an issue group.
// hipcc -O3 --offload-arch=gfx950 -o lds_b128_groups lds_b128_groups.hip && ./lds_b128_groups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int ROWS = 1024, STRIDE = 18;      // doubles per row (144 bytes)
constexpr int REPS = 256;

__global__ void __launch_bounds__(64) k_probe(const int* __restrict__ rows, int n_patterns, long long* __restrict__ cycles, double* __restrict__ sink) {
  __shared__ __align__(16) double tab[ROWS * STRIDE];
  for (int i = threadIdx.x; i < ROWS * STRIDE; i += 64) tab[i] = (double)i;
  __syncthreads();
  double acc9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int pat = 0; pat < n_patterns; ++pat) {
    const int r = rows[pat * 64 + threadIdx.x];
    const double2* src = (const double2*)(tab + STRIDE * r);
    __syncthreads();
    const long long t0 = clock64();
    for (int rep = 0; rep < REPS; ++rep) {
      double2 v[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) v[q] = src[q];
#pragma unroll
      for (int q = 0; q < 9; ++q) acc9[q] += v[q].x;          // nine independent chains: the reads, not the adds, set the pace
      asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[pat] = t1 - t0;
  }
  double acc = 0.0;
  for (int q = 0; q < 9; ++q) acc += acc9[q];
  sink[threadIdx.x] = acc;
}

int main() {
  std::vector<int> pats;
  auto add = [&](const std::vector<int>& p) { pats.insert(pats.end(), p.begin(), p.end()); };
  std::vector<int> ident(64);
  for (int l = 0; l < 64; ++l) ident[l] = l;
  add(ident);                                                    // 0: rows = lane
  srand(7);
  for (int k = 0; k < 8; ++k) { std::vector<int> p(64); for (int l = 0; l < 64; ++l) p[l] = rand() % ROWS; add(p); }     // 1..8 random
  // 9..16: distinct classes inside each ASSUMED group, random otherwise
  auto group_of = [](int lane) { const int l = lane & 31; return (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1; };
  for (int k = 0; k < 8; ++k) {
    std::vector<int> p(64);
    for (int half = 0; half < 2; ++half)
      for (int g = 0; g < 2; ++g) {
        int cls[16]; for (int i = 0; i < 16; ++i) cls[i] = i;
        for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = cls[i]; cls[i] = cls[j]; cls[j] = t; }
        int n = 0;
        for (int l = 0; l < 32; ++l) if (group_of(l) == g) p[32 * half + l] = cls[n++] + 16 * (rand() % (ROWS / 16));
      }
    add(p);
  }
  // 17..: 16 consecutive lanes distinct (groups = lanes 0-15, 16-31, ...)
  for (int k = 0; k < 4; ++k) {
    std::vector<int> p(64);
    for (int blk = 0; blk < 4; ++blk) {
      int cls[16]; for (int i = 0; i < 16; ++i) cls[i] = i;
      for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = cls[i]; cls[i] = cls[j]; cls[j] = t; }
      for (int l = 0; l < 16; ++l) p[16 * blk + l] = cls[l] + 16 * (rand() % (ROWS / 16));
    }
    add(p);
  }
  // pair probes: every lane reads row 5 (one address: a broadcast), lane a row 16 and lane b row 32 -- two different rows of
  // bank class 0: exactly ONE pair of lanes can conflict
  std::vector<int> bcast(64, 5);
  // distinct classes inside every 8 consecutive lanes only (the two halves of a 16-lane block drawn independently)
  const int n_before8 = (int)pats.size() / 64;
  for (int k = 0; k < 8; ++k) {
    std::vector<int> p(64);
    for (int blk = 0; blk < 8; ++blk) {
      int cls[16]; for (int i = 0; i < 16; ++i) cls[i] = i;
      for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = cls[i]; cls[i] = cls[j]; cls[j] = t; }
      for (int l = 0; l < 8; ++l) p[8 * blk + l] = cls[l] + 16 * (rand() % (ROWS / 16));
    }
    add(p);
  }
  // lanes l and l + 8 of every 16-lane block share a class, the eight pairs distinct
  for (int k = 0; k < 4; ++k) {
    std::vector<int> p(64);
    for (int blk = 0; blk < 4; ++blk) {
      int cls[16]; for (int i = 0; i < 16; ++i) cls[i] = i;
      for (int i = 15; i > 0; --i) { int j = rand() % (i + 1); int t = cls[i]; cls[i] = cls[j]; cls[j] = t; }
      for (int l = 0; l < 8; ++l) { p[16 * blk + l] = cls[l] + 16 * (rand() % (ROWS / 16)); p[16 * blk + 8 + l] = cls[l] + 16 * (rand() % (ROWS / 16)); }
    }
    add(p);
  }
  add(bcast);
  const int n_fixed = (int)pats.size() / 64;
  for (int a = 0; a < 64; ++a)
    for (int b = 0; b < 64; ++b) { std::vector<int> p = bcast; p[a] = 16; if (a != b) p[b] = 32; add(p); }
  const int n_pat = (int)pats.size() / 64;
  int* d_rows; long long* d_cyc; double* d_sink;
  hipMalloc(&d_rows, pats.size() * sizeof(int)); hipMalloc(&d_cyc, n_pat * sizeof(long long)); hipMalloc(&d_sink, 64 * sizeof(double));
  hipMemcpy(d_rows, pats.data(), pats.size() * sizeof(int), hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d_rows, n_pat, d_cyc, d_sink);
  std::vector<long long> cyc(n_pat);
  hipMemcpy(cyc.data(), d_cyc, n_pat * sizeof(long long), hipMemcpyDeviceToHost);
  const double per = 1.0 / (REPS * 9);
  printf("cycles per ds_read_b128 (one wave, 64 lanes x 16 bytes):\n  rows = lane                      %6.2f\n", cyc[0] * per);
  double s = 0; for (int k = 1; k <= 8; ++k) s += cyc[k]; printf("  random rows (8 patterns)         %6.2f\n", s / 8 * per);
  s = 0; for (int k = 9; k <= 16; ++k) s += cyc[k]; printf("  distinct classes per ASSUMED group %6.2f\n", s / 8 * per);
  s = 0; for (int k = 17; k < 21; ++k) s += cyc[k]; printf("  distinct classes per 16 consecutive lanes %6.2f\n", s / 4 * per);
  s = 0; for (int k = n_before8; k < n_before8 + 8; ++k) s += cyc[k]; printf("  distinct classes per 8 consecutive lanes only %6.2f\n", s / 8 * per);
  s = 0; for (int k = n_before8 + 8; k < n_before8 + 12; ++k) s += cyc[k]; printf("  lanes l and l + 8 of a 16-lane block share a class %6.2f\n", s / 4 * per);
  printf("  all lanes one row (broadcast)    %6.2f\n", cyc[n_fixed - 1] * per);
  return 0;
}
