// The solve step for the BAL 9-parameter camera [rvec | t | f k1 k2] (BASELINE config 5 is stated
// on a BAL problem).  The reference has no counterpart: its only camera is cv2.projectPoints(..., distCoeffs=None)
// (src/bundle_adjuster.py:67) with one shared K.  Same algorithm as ba_solve -- LM, Schur complement on the cameras,
// matrix-free PCG, IRLS Huber, every sum in a fixed order -- with 9x9 camera blocks:
//
//   camera model   P = R(rvec) X + t,  p = -P.xy / P.z,  proj = f (1 + k1 |p|^2 + k2 |p|^4) p,  residual = uv - proj
//   blocks         A = d proj / d P (2x3, full: the radial term couples x and y),  B = A R
//                  Jp = -B;  Jc = [ (B_row x X) M | -A | -rad p | -f n2 p | -f n2^2 p ]     (M: right Jacobian of SO(3))
//
// These are first-version kernels: correct, deterministic, resident on the device, but written for clarity, not tuned
// like the 6-parameter family in ba_kernels.hpp (no LDS camera table: camera rows come through L1 / L2; 8 lanes per
// point; ONE workgroup for the camera-vector algebra, a thread per vector entry; the host waits for a verdict word after
// every PCG iteration).  Camera-major passes reuse that family's (camera, partition) wave mapping and its wave sums;
// the host loop (ba_solve_bal in ba_hip.hip) mirrors oracle.lm_solve(model='bal', precond='jacobi') line by line.
#pragma once
#include "ba_kernels.hpp"

namespace ba {

constexpr int BC = 9;                 // parameters per BAL camera
constexpr int BH = 45;                // packed upper triangle of a 9x9 block
constexpr int BLIN = BH + BC + 2;     // running sums of the camera half: Hcc | bc | sum r^2 | sum rho-term
constexpr int BAL_VEC_THREADS = 1024; // the camera-vector kernels are ONE workgroup
constexpr int BAL_PT_THREADS = 256;
constexpr int BAL_LANES = 8;          // lanes per point in the point passes (tracks are short on average, a few are long;
                                      // measured on the 1723-camera chain: 4 lanes 77 us per PCG iteration, 8: 70, 16: 76)
constexpr int BAL_PTS_PER_BLOCK = BAL_PT_THREADS / BAL_LANES;
constexpr int BF = BC * BC;           // a full 9x9 block, row-major (the vector kernels read rows)

__host__ __device__ constexpr int U9(int a, int b) { return a * 9 - a * (a - 1) / 2 + (b - a); }   // a <= b

struct BalPcg {            // device-resident PCG state; iteration k reads st[k & 1] and leaves st[(k + 1) & 1]
  double gamma_prev, alpha_prev, gamma0;
  int iters, done;
};
constexpr int BAL_CAMS_PER_WG = 28;                         // camera-vector kernels: whole cameras per workgroup,
constexpr int BAL_VEC_WG = 256;                             // 252 of 256 threads hold one vector entry each

struct BalObs {
  double r0, r1;          // residual
  double A[6];            // d proj / d P, rows (u, v)
  double B[6];            // A R
  double p0, p1, n2, rad, f;
};

__device__ inline void bal_obs(const double* __restrict__ cs, double f, double k1, double k2, double X0, double X1, double X2,
                               double u, double v, BalObs& g) {
  const double Px = cs[0] * X0 + cs[1] * X1 + cs[2] * X2 + cs[9];
  const double Py = cs[3] * X0 + cs[4] * X1 + cs[5] * X2 + cs[10];
  const double Pz = cs[6] * X0 + cs[7] * X1 + cs[8] * X2 + cs[11];
  const double iz = (Pz != 0.0) ? 1.0 / Pz : 1.0;             // guarded like K1 (obs_project)
  const double p0 = -Px * iz, p1 = -Py * iz;
  const double n2 = p0 * p0 + p1 * p1;
  const double rad = 1.0 + n2 * (k1 + k2 * n2), drad = k1 + 2.0 * k2 * n2;
  g.p0 = p0; g.p1 = p1; g.n2 = n2; g.rad = rad; g.f = f;
  g.r0 = u - f * rad * p0;
  g.r1 = v - f * rad * p1;
  const double d00 = f * (rad + 2.0 * drad * p0 * p0), d01 = f * 2.0 * drad * p0 * p1, d11 = f * (rad + 2.0 * drad * p1 * p1);
  // d p / d P = [-iz 0 Px iz^2; 0 -iz Py iz^2] = -iz [1 0 p0; 0 1 p1]
  g.A[0] = -iz * d00; g.A[1] = -iz * d01; g.A[2] = -iz * (d00 * p0 + d01 * p1);
  g.A[3] = -iz * d01; g.A[4] = -iz * d11; g.A[5] = -iz * (d01 * p0 + d11 * p1);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) g.B[3 * r + k] = g.A[3 * r] * cs[k] + g.A[3 * r + 1] * cs[3 + k] + g.A[3 * r + 2] * cs[6 + k];
}

// the two rows of Jc (9 each) at one observation
__device__ inline void bal_cam_rows(const double* __restrict__ cs, const BalObs& g, double X0, double X1, double X2,
                                    double (&J0)[BC], double (&J1)[BC]) {
  const double* M = cs + 12;
  const double a0 = g.B[1] * X2 - g.B[2] * X1, a1 = g.B[2] * X0 - g.B[0] * X2, a2 = g.B[0] * X1 - g.B[1] * X0;   // B_0 x X
  const double b0 = g.B[4] * X2 - g.B[5] * X1, b1 = g.B[5] * X0 - g.B[3] * X2, b2 = g.B[3] * X1 - g.B[4] * X0;   // B_1 x X
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    J0[q] = a0 * M[q] + a1 * M[3 + q] + a2 * M[6 + q];
    J1[q] = b0 * M[q] + b1 * M[3 + q] + b2 * M[6 + q];
    J0[3 + q] = -g.A[q];
    J1[3 + q] = -g.A[3 + q];
  }
  J0[6] = -g.rad * g.p0;             J1[6] = -g.rad * g.p1;
  J0[7] = -g.f * g.n2 * g.p0;        J1[7] = -g.f * g.n2 * g.p1;
  J0[8] = -g.f * g.n2 * g.n2 * g.p0; J1[8] = -g.f * g.n2 * g.n2 * g.p1;
}

__device__ inline void bal_weights(bool robust, const BalObs& g, double hub_c, double& w0, double& w1, double& rho) {
  w0 = 1.0; w1 = 1.0;
  rho = g.r0 * g.r0 + g.r1 * g.r1;
  if (robust) { double t0, t1; huber(g.r0, hub_c, t0, w0); huber(g.r1, hub_c, t1, w1); rho = t0 + t1; }
}

// ---- K2, camera half: partL[(k Nc + c) BLIN + q] = sums over partition k of camera c
template <bool ROBUST>
__global__ void __launch_bounds__(64 * WPB)
k_bal_lin_cam(const double* __restrict__ cs, const double* __restrict__ intr, const double* __restrict__ ptab,
              const int* __restrict__ offk, const int* __restrict__ c_pt, const double2* __restrict__ c_uv, double hub_c,
              int n_cams, int band, int fixed_cam, double* __restrict__ partL) {
  Seg s;
  if (!cam_segment(offk, n_cams, band, s)) return;
  const double* cam = cs + CS * s.c;
  const double f = intr[3 * s.c], k1 = intr[3 * s.c + 1], k2 = intr[3 * s.c + 2];
  double acc[BLIN];
#pragma unroll
  for (int q = 0; q < BLIN; ++q) acc[q] = 0.0;
  for (int i = s.beg + s.lane; i < s.end; i += 64) {
    const double4 X = *(const double4*)(ptab + PT * (size_t)c_pt[i]);
    const double2 uv = c_uv[i];
    BalObs g;
    bal_obs(cam, f, k1, k2, X.x, X.y, X.z, uv.x, uv.y, g);
    double w0, w1, rho;
    bal_weights(ROBUST, g, hub_c, w0, w1, rho);
    acc[BH + BC] += g.r0 * g.r0 + g.r1 * g.r1;
    acc[BH + BC + 1] += rho;
    if (s.c != fixed_cam) {
      double J0[BC], J1[BC];
      bal_cam_rows(cam, g, X.x, X.y, X.z, J0, J1);
#pragma unroll
      for (int a = 0; a < BC; ++a) {
        const double wa0 = w0 * J0[a], wa1 = w1 * J1[a];
#pragma unroll
        for (int b = a; b < BC; ++b) acc[U9(a, b)] += wa0 * J0[b] + wa1 * J1[b];
        acc[BH + a] += wa0 * g.r0 + wa1 * g.r1;
      }
    }
  }
  wave_store_sums<BLIN>(acc, s.lane, partL + ((size_t)s.k * n_cams + s.c) * BLIN);
}

// ---- K2, point half (BAL_LANES lanes per point): Hpp (6 packed), bp (3), block maximum of |bp|
template <bool ROBUST>
__global__ void __launch_bounds__(BAL_PT_THREADS)
k_bal_lin_pt(const double* __restrict__ cs, const double* __restrict__ intr, const double* __restrict__ ptab,
             const int* __restrict__ pt_off, const int* __restrict__ p_cam, const double2* __restrict__ p_uv, double hub_c,
             int n_pts, double* __restrict__ Hpp, double* __restrict__ bp, double* __restrict__ partG) {
  __shared__ double sm[BAL_PT_THREADS / 64];
  const int p = blockIdx.x * BAL_PTS_PER_BLOCK + threadIdx.x / BAL_LANES, sub = threadIdx.x % BAL_LANES;
  double gm = 0.0;
  double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (p < n_pts) {
    const double4 X = *(const double4*)(ptab + PT * (size_t)p);
    for (int j = pt_off[p] + sub; j < pt_off[p + 1]; j += BAL_LANES) {
      const int c = p_cam[j];
      const double2 uv = p_uv[j];
      BalObs g;
      bal_obs(cs + CS * (size_t)c, intr[3 * c], intr[3 * c + 1], intr[3 * c + 2], X.x, X.y, X.z, uv.x, uv.y, g);
      double w0, w1, rho;
      bal_weights(ROBUST, g, hub_c, w0, w1, rho);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const double wa0 = w0 * g.B[q], wa1 = w1 * g.B[3 + q];          // Jp = -B: Jp^T w Jp = B^T w B
#pragma unroll
        for (int r = q; r < 3; ++r) a[U3(q, r)] += wa0 * g.B[r] + wa1 * g.B[3 + r];
        a[6 + q] -= wa0 * g.r0 + wa1 * g.r1;                              // Jp^T w r
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 9; ++q) a[q] = lanes_sum<BAL_LANES>(a[q]);
  if (p < n_pts && sub == BAL_LANES - 1) {
#pragma unroll
    for (int q = 0; q < 6; ++q) Hpp[6 * (size_t)p + q] = a[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) { bp[3 * (size_t)p + q] = a[6 + q]; gm = nanmax(gm, fabs(a[6 + q])); }
  }
  gm = wave_nanmax(gm);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int w = 0; w < BAL_PT_THREADS / 64; ++w) m = nanmax(m, sm[w]);
    partG[blockIdx.x] = m;
  }
}

// ---- point pass of the Schur product (BAL_LANES lanes per point).  u = sum_o Jp^T w (Jc v_c);
//   MODE 0 (PCG): y = Hppinv u into the point record's y slot; partB[4 block] = sum u . y.
//   MODE 1 (back substitution, v = dc): dp = -(y0 + Hppinv u), trial point, partB[block][4] = bp.dp, sum Dp dp^2, |dp|^2, |X|^2
template <bool ROBUST, int MODE>
__global__ void __launch_bounds__(BAL_PT_THREADS)
k_bal_pt_schur(const double* __restrict__ cs, const double* __restrict__ intr, double* __restrict__ ptab,
               const int* __restrict__ pt_off, const int* __restrict__ p_cam, const double2* __restrict__ p_uv, double hub_c,
               int n_pts, int fixed_cam, const double* __restrict__ vec, const double* __restrict__ Hppinv,
               const double* __restrict__ y0, const double* __restrict__ Hpp, const double* __restrict__ bp,
               double* __restrict__ ptab_trial, double* __restrict__ partB, const BalPcg* __restrict__ st_k) {
  __shared__ double sm[4 * (BAL_PT_THREADS / 64)];
  if (st_k && st_k->done) return;                      // queued past the end of PCG
  const int p = blockIdx.x * BAL_PTS_PER_BLOCK + threadIdx.x / BAL_LANES, sub = threadIdx.x % BAL_LANES;
  double acc[4] = {0, 0, 0, 0};
  double u[3] = {0, 0, 0};
  double4 X = make_double4(0, 0, 0, 0);
  if (p < n_pts) {
    X = *(const double4*)(ptab + PT * (size_t)p);
    for (int j = pt_off[p] + sub; j < pt_off[p + 1]; j += BAL_LANES) {
      const int c = p_cam[j];
      if (c == fixed_cam) continue;
      const double2 uv = p_uv[j];
      const double* cam = cs + CS * (size_t)c;
      BalObs g;
      bal_obs(cam, intr[3 * c], intr[3 * c + 1], intr[3 * c + 2], X.x, X.y, X.z, uv.x, uv.y, g);
      double w0, w1, rho;
      bal_weights(ROBUST, g, hub_c, w0, w1, rho);
      // Jc v without forming Jc: (B_row x X) . (M v_r) - A_row . v_t - (rad v_f + f n2 v_k1 + f n2^2 v_k2) p_row; vec = vt
      const double* v = vec + BC * (size_t)c;
      const double a0 = g.B[1] * X.z - g.B[2] * X.y, a1 = g.B[2] * X.x - g.B[0] * X.z, a2 = g.B[0] * X.y - g.B[1] * X.x;
      const double b0 = g.B[4] * X.z - g.B[5] * X.y, b1 = g.B[5] * X.x - g.B[3] * X.z, b2 = g.B[3] * X.y - g.B[4] * X.x;
      const double ki = g.rad * v[6] + g.f * g.n2 * (v[7] + g.n2 * v[8]);
      double s0 = a0 * v[0] + a1 * v[1] + a2 * v[2] - (g.A[0] * v[3] + g.A[1] * v[4] + g.A[2] * v[5]) - ki * g.p0;
      double s1 = b0 * v[0] + b1 * v[1] + b2 * v[2] - (g.A[3] * v[3] + g.A[4] * v[4] + g.A[5] * v[5]) - ki * g.p1;
      s0 *= w0; s1 *= w1;
      u[0] -= g.B[0] * s0 + g.B[3] * s1;                      // Jp^T (.), Jp = -B
      u[1] -= g.B[1] * s0 + g.B[4] * s1;
      u[2] -= g.B[2] * s0 + g.B[5] * s1;
    }
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) u[q] = lanes_sum<BAL_LANES>(u[q]);
  if (p < n_pts && sub == BAL_LANES - 1) {
    double hi[6], yy[3];
#pragma unroll
    for (int q = 0; q < 6; ++q) hi[q] = Hppinv[6 * (size_t)p + q];
    sym3_mul(hi, u, yy);
    if (MODE == 0) {
      double* o = ptab + PT * (size_t)p + 4;
      o[0] = yy[0]; o[1] = yy[1]; o[2] = yy[2];
      acc[0] = u[0] * yy[0] + u[1] * yy[1] + u[2] * yy[2];      // u . y: the point half of z . S z (summed per block below)
    } else {
      const double d0 = -(y0[3 * (size_t)p] + yy[0]), d1 = -(y0[3 * (size_t)p + 1] + yy[1]), d2 = -(y0[3 * (size_t)p + 2] + yy[2]);
      double* o = ptab_trial + PT * (size_t)p;
      o[0] = X.x + d0; o[1] = X.y + d1; o[2] = X.z + d2;
      const double D0 = fmax(Hpp[6 * (size_t)p], DIAG_FLOOR), D1 = fmax(Hpp[6 * (size_t)p + 3], DIAG_FLOOR),
                   D2 = fmax(Hpp[6 * (size_t)p + 5], DIAG_FLOOR);
      acc[0] = bp[3 * (size_t)p] * d0 + bp[3 * (size_t)p + 1] * d1 + bp[3 * (size_t)p + 2] * d2;
      acc[1] = D0 * d0 * d0 + D1 * d1 * d1 + D2 * d2 * d2;
      acc[2] = d0 * d0 + d1 * d1 + d2 * d2;
      acc[3] = X.x * X.x + X.y * X.y + X.z * X.z;
    }
  }
  if (partB) {
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = wave_total_dpp(acc[q]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sm[(threadIdx.x >> 6) * 4 + q] = acc[q];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      double a = 0.0;
      for (int w = 0; w < BAL_PT_THREADS / 64; ++w) a += sm[w * 4 + threadIdx.x];
      partB[4 * (size_t)blockIdx.x + threadIdx.x] = a;
    }
  }
}

// ---- camera pass of the Schur product: part9[(k Nc + c) 9 + q] = sum over partition k of Jc^T w (Jp y_p), y in the point record
template <bool ROBUST>
__global__ void __launch_bounds__(64 * WPB)
k_bal_cam_schur(const double* __restrict__ cs, const double* __restrict__ intr, const double* __restrict__ ptab,
                const int* __restrict__ offk, const int* __restrict__ c_pt, const double2* __restrict__ c_uv, double hub_c,
                int n_cams, int band, int fixed_cam, double* __restrict__ part9, const BalPcg* __restrict__ st_k,
                const double* __restrict__ partB, int nB, double* __restrict__ uy_out) {
  if (st_k && st_k->done) return;                      // queued past the end of PCG
  if (uy_out && blockIdx.x == gridDim.x - 1) {         // one extra workgroup: u . y of the point pass, blocks in a fixed order
    __shared__ double smu[WPB];
    double a = 0.0;
    for (int b = threadIdx.x; b < nB; b += 64 * WPB) a += partB[4 * (size_t)b];
    a = wave_total_dpp(a);
    if ((threadIdx.x & 63) == 0) smu[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0.0; for (int w = 0; w < WPB; ++w) t += smu[w]; uy_out[0] = t; }
    return;
  }
  Seg s;
  if (!cam_segment(offk, n_cams, band, s)) return;
  const double* cam = cs + CS * s.c;
  const double f = intr[3 * s.c], k1 = intr[3 * s.c + 1], k2 = intr[3 * s.c + 2];
  double acc[BC];
#pragma unroll
  for (int q = 0; q < BC; ++q) acc[q] = 0.0;
  if (s.c != fixed_cam) {
    for (int i = s.beg + s.lane; i < s.end; i += 64) {
      const double* rec = ptab + PT * (size_t)c_pt[i];
      const double4 X = *(const double4*)rec;
      const double y0 = rec[4], y1 = rec[5], y2 = rec[6];
      const double2 uv = c_uv[i];
      BalObs g;
      bal_obs(cam, f, k1, k2, X.x, X.y, X.z, uv.x, uv.y, g);
      double w0, w1, rho;
      bal_weights(ROBUST, g, hub_c, w0, w1, rho);
      double J0[BC], J1[BC];
      bal_cam_rows(cam, g, X.x, X.y, X.z, J0, J1);
      const double t0 = -w0 * (g.B[0] * y0 + g.B[1] * y1 + g.B[2] * y2);       // w (Jp y), Jp = -B
      const double t1 = -w1 * (g.B[3] * y0 + g.B[4] * y1 + g.B[5] * y2);
#pragma unroll
      for (int q = 0; q < BC; ++q) acc[q] += J0[q] * t0 + J1[q] * t1;
    }
  }
  wave_store_sums<BC>(acc, s.lane, part9 + ((size_t)s.k * n_cams + s.c) * BC);
}

// ---- camera-vector algebra, one workgroup.  Deterministic workgroup sum: wave sums by DPP, the waves in order.
template <int N>
__device__ inline void bal_block_sum(double (&v)[N], double* __restrict__ sm /* [16][N] */) {
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_total_dpp(v[q]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();                               // sm may still be read from the previous sum
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) sm[wv * N + q] = v[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < N; ++q) {
    double a = 0.0;
    for (int w = 0; w < BAL_VEC_THREADS / 64; ++w) a += sm[w * N + q];
    v[q] = a;
  }
}

// Linearisation -> damped camera blocks, full row-major 9x9 so that the vector kernels read rows.  relin: fold the camera
// half's partition sums into HccBc first (a rejected step only re-damps).  Also the cost of this linearisation (the
// camera half carried sum r^2 and the rho terms).
constexpr int BAL_PREP_WG = 64;     // cameras per workgroup of k_bal_prep (one wave: the 54-word fold lives in registers)
__global__ void __launch_bounds__(BAL_PREP_WG)
k_bal_prep(const double* __restrict__ partL, int relin, double lambda, int n_cams, int fixed_cam, double* __restrict__ HccBc,
           double* __restrict__ Hd, double* __restrict__ out /* [2 wg], [2 wg + 1]: this workgroup's sse, rho-sum (relin only) */) {
  const int c = blockIdx.x * BAL_PREP_WG + threadIdx.x;
  double cs2[2] = {0.0, 0.0};
  if (c < n_cams) {
    double h[BH + BC];
    if (relin) {
#pragma unroll
      for (int q = 0; q < BH + BC; ++q) h[q] = 0.0;
      for (int k = 0; k < NPART; ++k) {
        const double* src = partL + ((size_t)k * n_cams + c) * BLIN;
#pragma unroll
        for (int q = 0; q < BH + BC; ++q) h[q] += src[q];
        cs2[0] += src[BH + BC]; cs2[1] += src[BH + BC + 1];
      }
#pragma unroll
      for (int q = 0; q < BH + BC; ++q) HccBc[(size_t)c * (BH + BC) + q] = h[q];
    } else {
#pragma unroll
      for (int q = 0; q < BH; ++q) h[q] = HccBc[(size_t)c * (BH + BC) + q];
    }
    // damped block (what the operator multiplies by); the fixed camera's block is the identity
#pragma unroll
    for (int a = 0; a < BC; ++a)
#pragma unroll
      for (int b = 0; b < BC; ++b) {
        double v = h[a <= b ? U9(a, b) : U9(b, a)];
        if (a == b) v += lambda * fmax(v, DIAG_FLOOR);
        if (c == fixed_cam) v = (a == b) ? 1.0 : 0.0;
        Hd[(size_t)c * BF + a * BC + b] = v;
      }
  }
  if (relin) {                                      // per-workgroup partial of the cost; the host adds the few of them in order
    cs2[0] = wave_total_dpp(cs2[0]); cs2[1] = wave_total_dpp(cs2[1]);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = cs2[0]; out[2 * blockIdx.x + 1] = cs2[1]; }
  }
}

// Schur-Jacobi: partS[(k Nc + c) BH + q] = sum over partition k of W Hpp^-1 W^T (9x9, packed), W = Jc^T w Jp per observation
template <bool ROBUST>
__global__ void __launch_bounds__(64 * WPB)
k_bal_cam_sdiag(const double* __restrict__ cs, const double* __restrict__ intr, const double* __restrict__ ptab,
                const int* __restrict__ offk, const int* __restrict__ c_pt, const double2* __restrict__ c_uv, double hub_c,
                int n_cams, int band, int fixed_cam, const double* __restrict__ Hppinv, double* __restrict__ partS) {
  Seg s;
  if (!cam_segment(offk, n_cams, band, s)) return;
  const double* cam = cs + CS * s.c;
  const double f = intr[3 * s.c], k1 = intr[3 * s.c + 1], k2 = intr[3 * s.c + 2];
  double acc[BH];
#pragma unroll
  for (int q = 0; q < BH; ++q) acc[q] = 0.0;
  if (s.c != fixed_cam) {
    for (int i = s.beg + s.lane; i < s.end; i += 64) {
      const int p = c_pt[i];
      const double4 X = *(const double4*)(ptab + PT * (size_t)p);
      const double2 uv = c_uv[i];
      double hi[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) hi[q] = Hppinv[6 * (size_t)p + q];
      BalObs g;
      bal_obs(cam, f, k1, k2, X.x, X.y, X.z, uv.x, uv.y, g);
      double w0, w1, rho;
      bal_weights(ROBUST, g, hub_c, w0, w1, rho);
      double J0[BC], J1[BC];
      bal_cam_rows(cam, g, X.x, X.y, X.z, J0, J1);
      // W = -(w0 J0 B0^T + w1 J1 B1^T); with G = B Hinv B^T-like 2x2 weights: W Hinv W^T = [J0 J1] C [J0 J1]^T,
      // C = diag(w) (B Hinv B^T) diag(w)  (2x2, symmetric)
      double t0[3], t1[3];
      sym3_mul(hi, g.B, t0);
      sym3_mul(hi, g.B + 3, t1);
      const double c00 = w0 * w0 * (g.B[0] * t0[0] + g.B[1] * t0[1] + g.B[2] * t0[2]);
      const double c01 = w0 * w1 * (g.B[0] * t1[0] + g.B[1] * t1[1] + g.B[2] * t1[2]);
      const double c11 = w1 * w1 * (g.B[3] * t1[0] + g.B[4] * t1[1] + g.B[5] * t1[2]);
#pragma unroll
      for (int a = 0; a < BC; ++a) {
        const double ua = c00 * J0[a] + c01 * J1[a], va = c01 * J0[a] + c11 * J1[a];
#pragma unroll
        for (int b = a; b < BC; ++b) acc[U9(a, b)] += ua * J0[b] + va * J1[b];
      }
    }
  }
  wave_store_sums<BH>(acc, s.lane, partS + ((size_t)s.k * n_cams + s.c) * BH);
}

// Preconditioner blocks: Minv = (Hd - sum_k partS)^-1 (Schur-Jacobi; partS == nullptr: block-Jacobi), full row-major 9x9.
// Dense 9x9 Cholesky + inverse in private memory, once per damping change.
__global__ void __launch_bounds__(256)
k_bal_minv(const double* __restrict__ Hd, const double* __restrict__ partS, int n_cams, int fixed_cam, double* __restrict__ Minv) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n_cams) return;
  double Lm[BC][BC], Li[BC][BC];
  for (int a = 0; a < BC; ++a)
    for (int b = 0; b < BC; ++b) Lm[a][b] = Hd[(size_t)c * BF + a * BC + b];
  if (partS && c != fixed_cam) {
    for (int a = 0; a < BC; ++a)
      for (int b = a; b < BC; ++b) {
        double t = 0.0;
        for (int k = 0; k < NPART; ++k) t += partS[((size_t)k * n_cams + c) * BH + U9(a, b)];
        Lm[a][b] -= t;
        if (b != a) Lm[b][a] -= t;
      }
  }
  for (int j = 0; j < BC; ++j) {                       // Cholesky, lower triangle in place
    double d = Lm[j][j];
    for (int k = 0; k < j; ++k) d -= Lm[j][k] * Lm[j][k];
    d = sqrt(fmax(d, DIAG_FLOOR));
    Lm[j][j] = d;
    for (int i = j + 1; i < BC; ++i) {
      double sacc = Lm[i][j];
      for (int k = 0; k < j; ++k) sacc -= Lm[i][k] * Lm[j][k];
      Lm[i][j] = sacc / d;
    }
  }
  for (int j = 0; j < BC; ++j) {                       // Li = L^-1 (lower), column by column
    for (int i = 0; i < BC; ++i) Li[i][j] = 0.0;
    Li[j][j] = 1.0 / Lm[j][j];
    for (int i = j + 1; i < BC; ++i) {
      double sacc = 0.0;
      for (int k = j; k < i; ++k) sacc -= Lm[i][k] * Li[k][j];
      Li[i][j] = sacc / Lm[i][i];
    }
  }
  for (int a = 0; a < BC; ++a)                         // (L L^T)^-1 = Li^T Li
    for (int b = a; b < BC; ++b) {
      double sacc = 0.0;
      for (int k = b; k < BC; ++k) sacc += Li[k][a] * Li[k][b];
      Minv[(size_t)c * BF + a * BC + b] = sacc;
      Minv[(size_t)c * BF + b * BC + a] = sacc;
    }
}

// row e = 9 c + a of a block-diagonal product: sum_b M[c][a][b] v[9 c + b]
__device__ inline double bal_row_dot(const double* __restrict__ M, const double* __restrict__ v, int e) {
  const int c = e / BC;
  const double* row = M + (size_t)e * BC;
  const double* vc = v + (size_t)c * BC;
  double s = 0.0;
#pragma unroll
  for (int b = 0; b < BC; ++b) s += row[b] * vc[b];
  return s;
}

// The point passes multiply Jc by a camera vector v through its rotation part as (B_row x X) . (M v_r): M v_r is computed
// HERE, once per camera and vector, instead of once per observation -- entry e = 9 c + a of vt = (M v_r | v_t | v_f v_k1 v_k2).
// cs: camera states (M at [12..20], row-major); v complete for camera c (same workgroup, barrier before the call).
// (Packing pose, intrinsics and vt into one record per camera for the point passes was measured: no gain at a 208-byte
// stride, 13 % slower at 256 bytes -- the dense small arrays cache better.  So was an LDS window per 32-point workgroup
// (min .. max camera of its observations, 208-byte rows): 79 instead of 60 us per PCG iteration on the 1723-camera chain --
// with tracks up to 120 cameras long a 32-point window holds about as many rows as the workgroup has observations.)
__device__ inline double bal_vt_entry(const double* __restrict__ cs, const double* __restrict__ v, int e) {
  const int c = e / BC, a = e % BC;
  if (a >= 3) return v[e];
  const double* M = cs + CS * (size_t)c + 12 + 3 * a;
  const double* vc = v + (size_t)c * BC;
  return M[0] * vc[0] + M[1] * vc[1] + M[2] * vc[2];
}

// PCG start.  g = -(bc - W y0) from the camera pass on y0; x = 0, r = g, z = Minv r, p = s = 0; the two dot products the
// first step needs (gamma = r . z, zeta = z . Hd z) into partV[0], partV[1] (the other workgroups' slots zero); vt = the
// point passes' form of z; max |gradient|.  One workgroup, a thread per vector entry.
__global__ void __launch_bounds__(BAL_VEC_THREADS)
k_bal_pcg_init(const double* __restrict__ HccBc, const double* __restrict__ part9, const double* __restrict__ Hd,
               const double* __restrict__ Minv, const double* __restrict__ partG, int nG, int n_cams, int fixed_cam,
               double* __restrict__ x, double* __restrict__ r, double* __restrict__ z, double* __restrict__ p, double* __restrict__ sv,
               BalPcg* __restrict__ st, double* __restrict__ partV, int n_wg, double* __restrict__ host_out /* [0] rz0, [1] gmax */,
               long long* __restrict__ host_flag, long long seq, const double* __restrict__ cs, double* __restrict__ vt) {
  __shared__ double sm[(BAL_VEC_THREADS / 64) * 2];
  __shared__ double smax[BAL_VEC_THREADS / 64];
  const int n = n_cams * BC;
  double gm = 0.0;
  for (int e = threadIdx.x; e < n; e += BAL_VEC_THREADS) {
    const int c = e / BC, a = e % BC;
    const double bca = HccBc[(size_t)c * (BH + BC) + BH + a];
    double wy = 0.0;
    for (int k = 0; k < NPART; ++k) wy += part9[(size_t)k * n + e];
    r[e] = (c == fixed_cam) ? 0.0 : -(bca - wy);
    x[e] = 0.0; p[e] = 0.0; sv[e] = 0.0;
    gm = nanmax(gm, fabs(bca));
  }
  for (int b = threadIdx.x; b < nG; b += BAL_VEC_THREADS) gm = nanmax(gm, partG[b]);
  __syncthreads();                                      // r complete (this workgroup wrote all of it)
  for (int e = threadIdx.x; e < n; e += BAL_VEC_THREADS) z[e] = bal_row_dot(Minv, r, e);
  __syncthreads();                                      // z complete
  double acc[2] = {0.0, 0.0};
  for (int e = threadIdx.x; e < n; e += BAL_VEC_THREADS) {
    const double ze = z[e];
    acc[0] += r[e] * ze;
    acc[1] += ze * ((e / BC == fixed_cam) ? ze : bal_row_dot(Hd, z, e));
    vt[e] = bal_vt_entry(cs, z, e);
  }
  bal_block_sum<2>(acc, sm);
  for (int w = threadIdx.x; w < 2 * n_wg; w += BAL_VEC_THREADS) partV[w] = w < 2 ? acc[w] : 0.0;
  gm = wave_nanmax(gm);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int w = 0; w < BAL_VEC_THREADS / 64; ++w) m = nanmax(m, smax[w]);
    st[0].gamma_prev = 1.0; st[0].alpha_prev = 1.0; st[0].gamma0 = acc[0]; st[0].iters = 0; st[0].done = !(acc[0] > 0.0);
    host_out[0] = acc[0]; host_out[1] = m;
    publish_flag(host_flag, seq, 1);
  }
}

// One PCG iteration on the camera vectors in ONE launch of ceil(Nc / 28) workgroups (whole cameras per workgroup),
// Chronopoulos-Gear form as in k_pcg_step (ba_kernels.hpp): the two passes have applied S to z (not to the direction),
//     w = Hd z - W Hpp^-1 W^T z,   delta = z . w = zeta - u . y,   beta = gamma / gamma_prev,
//     alpha = gamma / (delta - beta gamma / alpha_prev),   p = z + beta p,   s = w + beta s,   x += alpha p,   r -= alpha s,
// and every scalar on the right is a sum that already exists when the kernel starts -- gamma = r . z and zeta = z . Hd z
// as per-workgroup partials of the PREVIOUS step (partV), u . y folded by the camera pass's extra workgroup -- so every
// workgroup re-sums them in the same order and all agree without a grid-wide barrier.  Same iterates as oracle.pcg in
// exact arithmetic.  The head of step k is also the convergence test after k iterations (gamma <= tol^2 gamma0): the
// update is then NOT applied and workgroup 0 tells the host.  Verdict word = flag_base + 4 (k + 1) + {1 applied, go on;
// 2 converged after k iterations; 3 broke down (denominator <= 0), k iterations stand}.
__global__ void __launch_bounds__(BAL_VEC_WG)
k_bal_cg_step(int k, BalPcg* __restrict__ st, const double* __restrict__ Hd, const double* __restrict__ Minv,
              const double* __restrict__ part9, const double* __restrict__ uy_src, const double* __restrict__ partV_in, int n_wg,
              int n_cams, int fixed_cam, double tol2, int min_iters, double* __restrict__ x, double* __restrict__ r,
              double* __restrict__ p, double* __restrict__ sv, double* __restrict__ z, double* __restrict__ partV_out,
              long long* __restrict__ host_flag, long long flag_base, const double* __restrict__ cs, double* __restrict__ vt) {
  __shared__ double sm[2][4];
  const BalPcg sin = st[k & 1];
  if (sin.done) return;
  double gamma = 0.0, zeta = 0.0;
  for (int w = 0; w < n_wg; ++w) { gamma += partV_in[2 * w]; zeta += partV_in[2 * w + 1]; }
  const double delta = zeta - uy_src[0];
  const double beta = (k == 0) ? 0.0 : gamma / sin.gamma_prev;
  const double denom = (k == 0) ? delta : delta - beta * gamma / sin.alpha_prev;
  int verdict = 1;
  if (k >= 1 && k >= min_iters && gamma <= tol2 * sin.gamma0) verdict = 2;
  else if (!(denom > 0.0) || !isfinite(denom)) verdict = 3;
  if (verdict != 1) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      BalPcg o = sin; o.done = 1; o.iters = k;
      st[(k + 1) & 1] = o;
      __hip_atomic_store(host_flag, flag_base + 4 * (long long)(k + 1) + verdict, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const double alpha = gamma / denom;
  const int n = n_cams * BC;
  const int e = blockIdx.x * (BAL_CAMS_PER_WG * BC) + threadIdx.x;
  const bool live = threadIdx.x < BAL_CAMS_PER_WG * BC && e < n;
  if (live) {
    const int c = e / BC;
    double w;
    if (c == fixed_cam) w = z[e];
    else {
      double t = 0.0;
      for (int kk = 0; kk < NPART; ++kk) t += part9[(size_t)kk * n + e];
      w = bal_row_dot(Hd, z, e) - t;
    }
    const double pe = z[e] + beta * p[e], se = w + beta * sv[e];
    p[e] = pe; sv[e] = se;
    x[e] += alpha * pe;
    r[e] -= alpha * se;
  }
  __syncthreads();                                       // the camera's nine r entries, all written by this workgroup
  double zn = 0.0;
  if (live) { zn = bal_row_dot(Minv, r, e); }
  __syncthreads();                                       // every read of the old z above is done
  if (live) z[e] = zn;
  __syncthreads();                                       // the camera's nine new z entries
  double g2 = 0.0, z2 = 0.0;
  if (live) {
    g2 = r[e] * zn;
    z2 = zn * ((e / BC == fixed_cam) ? zn : bal_row_dot(Hd, z, e));
    vt[e] = bal_vt_entry(cs, z, e);
  }
  g2 = wave_total_dpp(g2); z2 = wave_total_dpp(z2);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = g2; sm[1][threadIdx.x >> 6] = z2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partV_out[2 * blockIdx.x] = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
    partV_out[2 * blockIdx.x + 1] = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
    if (blockIdx.x == 0) {
      BalPcg o = sin; o.gamma_prev = gamma; o.alpha_prev = alpha; o.iters = k + 1;
      st[(k + 1) & 1] = o;
      __hip_atomic_store(host_flag, flag_base + 4 * (long long)(k + 1) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// trial cameras = cameras + dc (rotation vector, translation, f, k1, k2 all additive); camera-side sums of the step:
// out[0..4] = bc.dc, sum Dc dc^2, |dc|^2, |cams|^2, dc.r (the PCG residual left over: model term of an inexact step)
__global__ void __launch_bounds__(BAL_VEC_THREADS)
k_bal_update(const double* __restrict__ cams, const double* __restrict__ intr, const double* __restrict__ x,
             const double* __restrict__ r, const double* __restrict__ HccBc, int n_cams, int fixed_cam,
             double* __restrict__ cams_t, double* __restrict__ intr_t, double* __restrict__ cs_t, double* __restrict__ out,
             const double* __restrict__ cs_cur, double* __restrict__ vt) {
  __shared__ double sm[(BAL_VEC_THREADS / 64) * 5];
  for (int e = threadIdx.x; e < n_cams * BC; e += BAL_VEC_THREADS)      // the step in the point passes' form (back substitution)
    vt[e] = (e / BC == fixed_cam) ? 0.0 : bal_vt_entry(cs_cur, x, e);
  double acc[5] = {0, 0, 0, 0, 0};
  for (int c = threadIdx.x; c < n_cams; c += BAL_VEC_THREADS) {
    const double* h = HccBc + (size_t)c * (BH + BC);
    double cam[6];
#pragma unroll
    for (int a = 0; a < BC; ++a) {
      const double d = (c == fixed_cam) ? 0.0 : x[(size_t)c * BC + a];
      const double xv = a < 6 ? cams[6 * (size_t)c + a] : intr[3 * (size_t)c + a - 6];
      if (a < 6) { cam[a] = xv + d; cams_t[6 * (size_t)c + a] = cam[a]; }
      else intr_t[3 * (size_t)c + a - 6] = xv + d;
      acc[0] += h[BH + a] * d;
      if (c != fixed_cam) acc[1] += fmax(h[U9(a, a)], DIAG_FLOOR) * d * d;
      acc[2] += d * d;
      acc[3] += xv * xv;
      acc[4] += d * r[(size_t)c * BC + a];
    }
    camera_state(cam, cs_t + CS * (size_t)c);
  }
  bal_block_sum<5>(acc, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) out[q] = acc[q];
  }
}

// step sums to the host: out_host[0..8] = camera side (5) | point side (4, partB blocks) ; [9], [10] = sse, rho-sum of the trial
__global__ void __launch_bounds__(BAL_VEC_THREADS)
k_bal_step_sums(const double* __restrict__ cam5, const double* __restrict__ partB, int nB, const double* __restrict__ partR,
                int n_cams, double* __restrict__ host_out, long long* __restrict__ host_flag, long long seq) {
  __shared__ double sm[(BAL_VEC_THREADS / 64) * 6];
  double v[6] = {0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < nB; b += BAL_VEC_THREADS) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += partB[4 * (size_t)b + q];
  }
  for (int i = threadIdx.x; i < NPART * n_cams; i += BAL_VEC_THREADS) { v[4] += partR[2 * (size_t)i]; v[5] += partR[2 * (size_t)i + 1]; }
  bal_block_sum<6>(v, sm);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 0; q < 5; ++q) host_out[q] = cam5[q];
#pragma unroll
    for (int q = 0; q < 4; ++q) host_out[5 + q] = v[q];
    host_out[9] = v[4]; host_out[10] = v[5];
    publish_flag(host_flag, seq, 1);
  }
}

}  // namespace ba
