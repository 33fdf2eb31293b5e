// Two-level preconditioner for band-structured ("chain") problems such as BASELINE config 5: sequential captures, every
// landmark seen from a run of consecutive cameras.  The reduced camera matrix S is then block-banded and its slow modes
// are drifts along the chain, which the per-camera Schur-Jacobi blocks cannot see (~136 PCG iterations per LM
// iteration on C5).  Additive coarse correction (Nicolaides coarse space):
//
//        M^-1 r  =  M_J^-1 r  +  P E^-1 P^T r ,        E = P^T S P ,
//
// P = piecewise-constant prolongation, one coarse unknown per parameter (6) and per AGGREGATE of VEC_CAMS = 16
// consecutive cameras -- exactly the cameras of one camera-vector workgroup, so the restriction P^T r is a sum inside
// a workgroup.  The fixed camera is left out of P.  Everything else (matrix-free S, the PCG recurrences) is unchanged.
//
// E is assembled once per damped system, without forming S:
//   E = sum_{c in a} Hccd_c (diagonal aggregates)  -  sum_p  U_{p,a}^T Hpp_p^-1 U_{p,b} ,    U_{p,a} = sum_{o of p, cam(o) in a} W_o^T
// (W_o = Jc^T w Jp, 6x3).  Observations inside a point are sorted by camera for such problems, so a point's
// observations of one aggregate are a contiguous RUN; k_coarse_runs computes U per run, k_coarse_pairs adds, for every
// pair of runs of one point, the 6x6 product into E.  The adds are atomic in 64-bit FIXED POINT (integer addition is
// associative): the result does not depend on the order in which the hardware performs them, so the solve stays
// bit-reproducible like every other sum of the library.  The scale is a power of two taken from E's largest diagonal
// entry.  E is block-banded (a point spans few aggregates); it is factored by a banded Cholesky in one workgroup and
// its explicit inverse (dense, (6 Na)^2 doubles) is formed column by column; per PCG iteration every aggregate's
// workgroup multiplies its six rows of E^-1 with the restricted residual.
#pragma once
#include "ba_kernels.hpp"

namespace ba {

// ---- per-run blocks U (3x6, row-major) -----------------------------------------------------------------
// run r = observations [run_beg[r], run_beg[r+1]) of the point-ordered list, all of point run_pt[r] and of cameras
// in one aggregate.  Thread per run.
template <bool ROBUST>
__global__ void __launch_bounds__(256)
k_coarse_runs(const double* __restrict__ cs, const double* __restrict__ ptab, const int* __restrict__ run_beg,
              const int* __restrict__ run_pt, const int* __restrict__ p_cam, const double2* __restrict__ p_w,
              double fx, double fy, int fixed_cam, int n_runs, double* __restrict__ U) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_runs) return;
  const int p = run_pt[r];
  const double4 X = *(const double4*)(ptab + PT * (size_t)p);
  double u[18];
#pragma unroll
  for (int q = 0; q < 18; ++q) u[q] = 0.0;
  for (int j = run_beg[r]; j < run_beg[r + 1]; ++j) {
    int c = p_cam[j];                              // ROBUST: the flagged copy
    double2 w = make_double2(1.0, 1.0);
    if (ROBUST && c < 0) w = p_w[j];
    if (ROBUST) c &= IDX_MASK;
    if (c == fixed_cam) continue;
    const double* cam = cs + CS * (size_t)c;
    double row[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) row[q] = cam[q];
    const double* M = cam + 12;
    Geom g;
    obs_geom_fast(row, X.x, X.y, X.z, fx, fy, g);
    double J0[6], J1[6];
    cam_jac_rows(g, X.x, X.y, X.z, J0, J1);
    // post-M camera rows: [J[0:3] M | J[3:6]]
    double c0[6], c1[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      c0[q] = J0[0] * M[q] + J0[1] * M[3 + q] + J0[2] * M[6 + q];
      c1[q] = J1[0] * M[q] + J1[1] * M[3 + q] + J1[2] * M[6 + q];
      c0[3 + q] = J0[3 + q];
      c1[3 + q] = J1[3 + q];
    }
    // W_o^T = Jp^T w Jc,  Jp = -P
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double a0 = -g.P[i] * w.x, a1 = -g.P[3 + i] * w.y;
#pragma unroll
      for (int q = 0; q < 6; ++q) u[6 * i + q] += a0 * c0[q] + a1 * c1[q];
    }
  }
  double* o = U + 18 * (size_t)r;
#pragma unroll
  for (int q = 0; q < 18; ++q) o[q] = u[q];
}

// ---- diagonal aggregates + fixed-point scale (one workgroup) ---------------------------------------------
// Eint[(6a+i) * n + (6a+j)] = fixed(sum_{c in a, c != fixed} Hccd_c[i][j]); info[0] = scale (a power of two)
__global__ void __launch_bounds__(1024)
k_coarse_diag(const double* __restrict__ Hccd, int n_cams, int fixed_cam, int n_agg, long long* __restrict__ Eint,
              double* __restrict__ info) {
  __shared__ double smax[16];
  __shared__ double s_scale;
  const int n = 6 * n_agg;
  double mx = 0.0;
  // pass 1: largest diagonal entry of any aggregate sum
  for (int t = threadIdx.x; t < n; t += 1024) {
    const int a = t / 6, i = t % 6;
    double d = 0.0;
    for (int c = a * VEC_CAMS; c < min(n_cams, (a + 1) * VEC_CAMS); ++c)
      if (c != fixed_cam) d += Hccd[21 * (size_t)c + U6(i, i)];
    mx = fmax(mx, d);
  }
  mx = wave_nanmax(mx);
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = smax[0];
    for (int w = 1; w < 16; ++w) m = nanmax(m, smax[w]);
    int e = 0;
    (void)frexp((m > 0.0 && isfinite(m)) ? m : 1.0, &e);       // m < 2^e
    s_scale = ldexp(1.0, 58 - e);                                // |entries| * scale < 2^58: room for the sums
    info[0] = s_scale;
    info[1] = m;
  }
  __syncthreads();
  const double scale = s_scale;
  for (int t = threadIdx.x; t < n_agg * 36; t += 1024) {
    const int a = t / 36, i = (t % 36) / 6, j = t % 6;
    double d = 0.0;
    for (int c = a * VEC_CAMS; c < min(n_cams, (a + 1) * VEC_CAMS); ++c)
      if (c != fixed_cam) d += Hccd[21 * (size_t)c + S6(i, j)];
    Eint[(size_t)(6 * a + i) * n + (6 * a + j)] = llrint(d * scale);
  }
}

// ---- pairs of runs of one point: E[a1][a2] -= U1^T Hinv U2 (and the transpose block) ------------------------
__global__ void __launch_bounds__(256)
k_coarse_pairs(const int2* __restrict__ pairs, int n_pairs, const int* __restrict__ run_pt, const int* __restrict__ run_agg,
               const double* __restrict__ U, const double* __restrict__ Hppinv, int n_agg, const double* __restrict__ info,
               unsigned long long* __restrict__ Eint) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_pairs) return;
  const int2 pr = pairs[t];
  const int p = run_pt[pr.x];
  const int a1 = run_agg[pr.x], a2 = run_agg[pr.y];
  const double scale = info[0];
  const int n = 6 * n_agg;
  double u1[18], u2[18], hi[6];
#pragma unroll
  for (int q = 0; q < 18; ++q) { u1[q] = U[18 * (size_t)pr.x + q]; u2[q] = U[18 * (size_t)pr.y + q]; }
#pragma unroll
  for (int q = 0; q < 6; ++q) hi[q] = Hppinv[6 * (size_t)p + q];
  double T[18];                                   // Hinv U2 (3x6)
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    T[j] = hi[0] * u2[j] + hi[1] * u2[6 + j] + hi[2] * u2[12 + j];
    T[6 + j] = hi[1] * u2[j] + hi[3] * u2[6 + j] + hi[4] * u2[12 + j];
    T[12 + j] = hi[2] * u2[j] + hi[4] * u2[6 + j] + hi[5] * u2[12 + j];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const double b = u1[i] * T[j] + u1[6 + i] * T[6 + j] + u1[12 + i] * T[12 + j];
      const long long v = -llrint(b * scale);
      atomicAdd(Eint + (size_t)(6 * a1 + i) * n + (6 * a2 + j), (unsigned long long)v);
      if (pr.x != pr.y) atomicAdd(Eint + (size_t)(6 * a2 + j) * n + (6 * a1 + i), (unsigned long long)v);
    }
  }
}

// fixed point -> double, the fixed camera's aggregate keeps a positive diagonal even if it has no free camera
__global__ void __launch_bounds__(256)
k_coarse_to_double(const long long* __restrict__ Eint, int n, const double* __restrict__ info, double* __restrict__ E) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= (size_t)n * n) return;
  double v = (double)Eint[t] / info[0];
  if (t / n == t % n && !(v > 0.0)) v = 1.0;
  E[t] = v;
}

// ---- banded Cholesky, one workgroup: E = L L^T in place (lower triangle; bw = half bandwidth in scalars) ----
__global__ void __launch_bounds__(1024)
k_coarse_cholesky(double* __restrict__ E, int n, int bw) {
  __shared__ double col[512];                       // L[j+1 .. j+bw][j]
  for (int j = 0; j < n; ++j) {
    const int m = min(bw, n - 1 - j);
    const double d = sqrt(fmax(E[(size_t)j * n + j], DIAG_FLOOR));
    if ((int)threadIdx.x < m) {
      const double l = E[(size_t)(j + 1 + threadIdx.x) * n + j] / d;
      col[threadIdx.x] = l;
      E[(size_t)(j + 1 + threadIdx.x) * n + j] = l;
    }
    if (threadIdx.x == 0) E[(size_t)j * n + j] = d;
    __syncthreads();
    // trailing update of the band: rows i in (j, j+m], columns k in (j, i]
    for (int t = threadIdx.x; t < m * m; t += 1024) {
      const int ii = t / m, kk = t % m;
      if (kk <= ii) E[(size_t)(j + 1 + ii) * n + (j + 1 + kk)] -= col[ii] * col[kk];
    }
    __syncthreads();
  }
}

// ---- explicit inverse from the banded factor: thread t solves L L^T x = e_t; Y (n x n) is the scratch AND the result
// (column t of E^-1 = row t, symmetric), stored so that threads access consecutive words: Y[i * n + t] ---------------
__global__ void __launch_bounds__(64)
k_coarse_inverse(const double* __restrict__ L, int n, int bw, double* __restrict__ Y) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  if (t >= n) return;
  // forward: L y = e_t (y[i] = 0 for i < t)
  for (int i = 0; i < n; ++i) {
    double s = (i == t) ? 1.0 : 0.0;
    if (i >= t) {
      const int k0 = max(t, i - bw);
      for (int k = k0; k < i; ++k) s -= L[(size_t)i * n + k] * Y[(size_t)k * n + t];
      Y[(size_t)i * n + t] = s / L[(size_t)i * n + i];
    } else {
      Y[(size_t)i * n + t] = 0.0;
    }
  }
  // backward: L^T x = y
  for (int i = n - 1; i >= 0; --i) {
    double s = Y[(size_t)i * n + t];
    const int k1 = min(n - 1, i + bw);
    for (int k = i + 1; k <= k1; ++k) s -= L[(size_t)k * n + i] * Y[(size_t)k * n + t];
    Y[(size_t)i * n + t] = s / L[(size_t)i * n + i];
  }
}

// ---- PCG with the coarse term --------------------------------------------------------------------------
// Vector kernel A (k_pcg_step<COARSE = true>, ba_kernels.hpp) leaves r, zJ = M_J^-1 r and the aggregate's restricted
// residual rc[6a..6a+5] = sum_{c in a, c != fixed} r_c.  Kernel C below (one wave per aggregate) adds the coarse
// correction and produces what the single-level vector kernel produced at its end: z, the partial dot products
// (gamma = r.z, zeta = z.Hccd z), vtil = (M z_r, z_t) into the point passes' camera table.
__global__ void __launch_bounds__(VEC_BLOCK)
k_pcg_coarse(int k, const double* __restrict__ Einv, const double* __restrict__ rc, int n_agg,
             const double* __restrict__ Hccd, const double* __restrict__ cs, int n_cams, int fixed_cam,
             const double* __restrict__ r, double* __restrict__ z, double* __restrict__ vtil,
             double* __restrict__ partV, int nblkV, const double* __restrict__ verdict, int check_verdict) {
  __shared__ double l_zc[6];
  const int a = blockIdx.x, lane = threadIdx.x;
  const int n = 6 * n_agg;
  if (check_verdict) {                            // iteration k already over (converged / breakdown): nothing to do
    double g, zt;
    if (pcg_verdict(verdict, k, g, zt)) return;
  }
  // zc = rows [6a, 6a+6) of E^-1 times rc
  double acc6[6] = {0, 0, 0, 0, 0, 0};
  for (int j = lane; j < n; j += 64) {
    const double v = rc[j];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc6[i] += Einv[(size_t)(6 * a + i) * n + j] * v;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc6[i] = wave_total_dpp(acc6[i]);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) l_zc[i] = acc6[i];
  }
  __syncthreads();
  const int c = vec_camera(n_cams);
  double acc[2] = {0, 0};
  if (c < n_cams && c != fixed_cam) {
    double zz[6], rr[6], h[21], hz[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { zz[q] = z[6 * (size_t)c + q] + l_zc[q]; rr[q] = r[6 * (size_t)c + q]; }
#pragma unroll
    for (int q = 0; q < 21; ++q) h[q] = Hccd[21 * (size_t)c + q];
    sym6_mul(h, zz, hz);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      z[6 * (size_t)c + q] = zz[q];
      acc[0] += rr[q] * zz[q];
      acc[1] += zz[q] * hz[q];
    }
    write_vtil(cs + CS * (size_t)c + 12, zz, vtil + TA * c + 12);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) acc[q] = wave_total_dpp(acc[q]);
  if (lane == 0) {
    double* pv = partV + (size_t)((k + 1) & 1) * 2 * nblkV;      // the slot the NEXT iteration's probe sums
    pv[2 * blockIdx.x] = acc[0];
    pv[2 * blockIdx.x + 1] = acc[1];
  }
}

}  // namespace ba
