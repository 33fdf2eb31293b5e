// Direct solver for sliding-window-sized problems: the reference's default use (BundleAdjuster(window_size=5),
// src/pipeline.py:39,99: a handful of keyframes, a few hundred landmarks).  At that size the multi-kernel LM / Schur /
// PCG loop is nothing but launch floors (174 launches at ~6 us for a 1.1 ms solve of a 2 k-observation window), while
// the reduced camera system is at most 48 x 48.  Here the WHOLE Levenberg-Marquardt loop runs in ONE workgroup of one
// kernel launch: same residual, same analytic blocks, same damping / gain-ratio / stopping rules as ba_solve, but the
// reduced system is formed explicitly and solved by dense Cholesky in LDS instead of PCG, and nothing returns to the
// host until the solve is over.
//
//   per LM iteration (all 1024 threads, __syncthreads between phases):
//     C1  camera-major, wave = camera: Hcc (21) | bc (6) | cost (2) by DPP wave sums              (linearise, camera half)
//     P1  point-major, thread = point: Hpp, bp, damped Hpp^-1, y0; every pair of the point's observations adds its
//         6x6 block -W_a Hpp^-1 W_b^T to S with 64-bit FIXED-POINT LDS atomics (integer addition is associative: the
//         result does not depend on the order of the adds; scale = power of two from the largest diagonal of Hcc + lam D)
//     C2  camera-major: W y0 -> reduced right-hand side g = -(bc - W y0)                            (exact fp64 sums)
//     S = blockdiag(Hcc + lam D) + fixed(S); Cholesky (6 Nc columns, all threads); two triangular solves (one wave)
//     camera update (thread = camera), P2 back substitution + model terms, C3 cost at the trial point
//     thread 0: gain ratio, accept / reject, Nielsen's damping update, ftol / xtol / gtol / max_iters (as ba_solve)
//
// Limits (checked by the host): Nc <= SMALL_MAX_CAMS, single rank.  Any number of points / observations works (thread-
// and lane-strided loops); the host only dispatches problems small enough that one workgroup beats the launch-bound path.
#pragma once
#include "ba_kernels.hpp"

namespace ba {

constexpr int SMALL_MAX_CAMS = 8;
constexpr int SMALL_N = 6 * SMALL_MAX_CAMS;       // reduced system dimension bound
constexpr int SMALL_THREADS = 1024;

struct SmallArgs {
  double* cams[2]; double* cs[2]; double* ptab[2];
  const int* offk; const int* c_pt; const double2* c_uv;
  const int* pt_off; const int* p_cam; const double2* p_uv;
  double* Hpp; double* bp; double* Hppinv; double* y0;     // per point: damped inverse, y0, diagonal D (in Hpp), bp
  int n_cams, n_pts, fixed_cam, robust;
  double fx, fy, cx, cy, hub_c;
  int max_iters; double ftol, xtol, gtol, lambda0;
  int cur;                                                   // which parameter set holds the start point
  ba_summary* summary; ba_iter_record* trace; int* cur_out;
};

// post-M camera rows of one observation: c0 / c1 = d res_u / d cam, d res_v / d cam (6 each)
__device__ inline void small_cam_rows(const double* __restrict__ cs, const Geom& g, double X0, double X1, double X2,
                                      double (&c0)[6], double (&c1)[6]) {
  double J0[6], J1[6];
  cam_jac_rows(g, X0, X1, X2, J0, J1);
  const double* M = cs + 12;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    c0[q] = J0[0] * M[q] + J0[1] * M[3 + q] + J0[2] * M[6 + q];
    c1[q] = J1[0] * M[q] + J1[1] * M[3 + q] + J1[2] * M[6 + q];
    c0[3 + q] = J0[3 + q];
    c1[3 + q] = J1[3 + q];
  }
}

// deterministic workgroup sum of N values held by every thread -> all threads get the totals (fixed order)
template <int N>
__device__ inline void small_block_sum(double (&v)[N], double* __restrict__ sm /* [16][N] */, double* __restrict__ out /* [N] */) {
#pragma unroll
  for (int q = 0; q < N; ++q) v[q] = wave_total_dpp(v[q]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) sm[wv * N + q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x < N) {
    double a = 0.0;
    for (int w = 0; w < SMALL_THREADS / 64; ++w) a += sm[w * N + threadIdx.x];
    out[threadIdx.x] = a;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(SMALL_THREADS)
k_small_lm(SmallArgs A) {
  __shared__ double l_cs[2][SMALL_MAX_CAMS][CS];
  __shared__ double l_cam[2][SMALL_MAX_CAMS][6];
  __shared__ double l_Hcc[SMALL_MAX_CAMS][21], l_bc[SMALL_MAX_CAMS][6];
  __shared__ double l_S[SMALL_N][SMALL_N + 1];
  __shared__ long long l_Sint[SMALL_N][SMALL_N];
  __shared__ double l_g[SMALL_N], l_dc[SMALL_N], l_y[SMALL_N];
  __shared__ double l_red[16 * 8], l_tot[8], l_camred[SMALL_MAX_CAMS][4], l_costred[SMALL_MAX_CAMS][2];
  __shared__ double s_scale, s_lambda, s_cost, s_sse, s_cost_new, s_sse_new, s_gmax;
  __shared__ int s_cur, s_stop, s_it, s_acc, s_status;
  __shared__ double s_nu;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int Nc = A.n_cams, Np = A.n_pts, n = 6 * Nc;
  const bool robust = A.robust != 0;

  // ---- start point into LDS
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) l_cam[A.cur][tid][q] = A.cams[A.cur][6 * tid + q];
    camera_state(&l_cam[A.cur][tid][0], &l_cs[A.cur][tid][0]);
  }
  if (tid == 0) { s_cur = A.cur; s_lambda = A.lambda0; s_nu = 2.0; s_stop = 0; s_it = 0; s_acc = 0; s_status = 0; }
  __syncthreads();

  // camera-major cost at parameter set w: sse, rho-sum -> s_sse_new, s_cost_new (thread 0 combines cameras in order)
  auto cam_cost = [&](int w) {
    if (wv < Nc) {
      const int c = wv;
      const double* cs = &l_cs[w][c][0];
      double acc[2] = {0.0, 0.0};
      for (int i = A.offk[c * (NPART + 1)] + lane; i < A.offk[c * (NPART + 1) + NPART]; i += 64) {
        const double4 X = *(const double4*)(A.ptab[w] + PT * (size_t)A.c_pt[i]);
        const double2 uv = A.c_uv[i];
        double xh, yh;
        obs_project(cs, X.x, X.y, X.z, xh, yh);
        const double ru = uv.x - (xh * A.fx + A.cx), rv = uv.y - (yh * A.fy + A.cy);
        acc[0] += ru * ru + rv * rv;
        if (robust) { double t0, t1, ww; huber(ru, A.hub_c, t0, ww); huber(rv, A.hub_c, t1, ww); acc[1] += t0 + t1; }
      }
      if (!robust) acc[1] = acc[0];
      acc[0] = wave_total_dpp(acc[0]); acc[1] = wave_total_dpp(acc[1]);
      if (lane == 0) { l_costred[c][0] = acc[0]; l_costred[c][1] = acc[1]; }
    }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int c = 0; c < Nc; ++c) { a += l_costred[c][0]; b += l_costred[c][1]; }
      s_sse_new = a; s_cost_new = 0.5 * b;
    }
    __syncthreads();
  };

  cam_cost(A.cur);
  if (tid == 0) {
    s_sse = s_sse_new; s_cost = s_cost_new;
    A.summary->initial_sse = s_sse; A.summary->initial_cost = s_cost;
    if (!isfinite(s_cost)) { s_stop = 1; s_status = -4; }          // BA_ERR_NUMERIC
    if (Np == 0 || A.max_iters <= 0) s_stop = 1;
  }
  __syncthreads();

  bool need_lin = true;
  while (!s_stop) {
    const int cur = s_cur, tr = 1 - cur;
    const double lambda = s_lambda;
    if (need_lin) {
      // ---- C1: camera half of the normal equations (post-M rows: no congruence afterwards)
      if (wv < Nc) {
        const int c = wv;
        const double* cs = &l_cs[cur][c][0];
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        if (c != A.fixed_cam) {
          for (int i = A.offk[c * (NPART + 1)] + lane; i < A.offk[c * (NPART + 1) + NPART]; i += 64) {
            const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)A.c_pt[i]);
            const double2 uv = A.c_uv[i];
            Geom g;
            obs_geom(cs, X.x, X.y, X.z, A.fx, A.fy, g);
            const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
            double w0 = 1.0, w1 = 1.0;
            if (robust) { double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1); }
            double c0[6], c1[6];
            small_cam_rows(cs, g, X.x, X.y, X.z, c0, c1);
#pragma unroll
            for (int a = 0; a < 6; ++a) {
              const double wa0 = w0 * c0[a], wa1 = w1 * c1[a];
#pragma unroll
              for (int b = a; b < 6; ++b) acc[U6(a, b)] += wa0 * c0[b] + wa1 * c1[b];
              acc[21 + a] += wa0 * ru + wa1 * rv;
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = wave_total_dpp(acc[q]);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < 21; ++q) l_Hcc[c][q] = acc[q];
#pragma unroll
          for (int q = 0; q < 6; ++q) l_bc[c][q] = acc[21 + q];
        }
      }
      __syncthreads();
    }
    // ---- fixed-point scale from the largest damped diagonal entry; S <- 0
    if (tid == 0) {
      double m = 0.0, gm = 0.0;
      for (int c = 0; c < Nc; ++c)
        for (int i = 0; i < 6; ++i) {
          const double d = l_Hcc[c][U6(i, i)];
          m = fmax(m, d + lambda * fmax(d, DIAG_FLOOR));
          gm = nanmax(gm, fabs(l_bc[c][i]));
        }
      int e = 0;
      (void)frexp((m > 0.0 && isfinite(m)) ? m : 1.0, &e);
      s_scale = ldexp(1.0, 58 - e);
      s_gmax = gm;
    }
    for (int t = tid; t < SMALL_N * SMALL_N; t += SMALL_THREADS) l_Sint[t / SMALL_N][t % SMALL_N] = 0;
    __syncthreads();
    const double scale = s_scale;
    // ---- P1: point half, damped inverse, y0, Schur contributions
    double gmp = 0.0;
    for (int p = tid; p < Np; p += SMALL_THREADS) {
      const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
      const int beg = A.pt_off[p], end = A.pt_off[p + 1];
      double hinv[6];
      if (need_lin) {
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = beg; j < end; ++j) {
          const int c = A.p_cam[j];
          const double2 uv = A.p_uv[j];
          Geom g;
          obs_geom(&l_cs[cur][c][0], X.x, X.y, X.z, A.fx, A.fy, g);
          const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
          double w0 = 1.0, w1 = 1.0;
          if (robust) { double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1); }
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const double wa0 = w0 * g.P[q], wa1 = w1 * g.P[3 + q];
#pragma unroll
            for (int r = q; r < 3; ++r) a[U3(q, r)] += wa0 * g.P[r] + wa1 * g.P[3 + r];
            a[6 + q] -= wa0 * ru + wa1 * rv;
          }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) A.Hpp[6 * (size_t)p + q] = a[q];
#pragma unroll
        for (int q = 0; q < 3; ++q) { A.bp[3 * (size_t)p + q] = a[6 + q]; gmp = nanmax(gmp, fabs(a[6 + q])); }
      }
      double h[6], b3[3], y[3];
#pragma unroll
      for (int q = 0; q < 6; ++q) h[q] = A.Hpp[6 * (size_t)p + q];
#pragma unroll
      for (int q = 0; q < 3; ++q) b3[q] = A.bp[3 * (size_t)p + q];
      h[0] += lambda * fmax(h[0], DIAG_FLOOR);
      h[3] += lambda * fmax(h[3], DIAG_FLOOR);
      h[5] += lambda * fmax(h[5], DIAG_FLOOR);
      sym3_inverse(h, hinv);
      sym3_mul(hinv, b3, y);
#pragma unroll
      for (int q = 0; q < 6; ++q) A.Hppinv[6 * (size_t)p + q] = hinv[q];
#pragma unroll
      for (int q = 0; q < 3; ++q) A.y0[3 * (size_t)p + q] = y[q];
      // pairs of observations of this point: S[ca][cb] -= W_a Hinv W_b^T
      for (int ja = beg; ja < end; ++ja) {
        const int ca = A.p_cam[ja];
        if (ca == A.fixed_cam) continue;
        double Ta[18];                                 // W_a Hinv, 6x3
        {
          const double2 uv = A.p_uv[ja];
          Geom g;
          obs_geom(&l_cs[cur][ca][0], X.x, X.y, X.z, A.fx, A.fy, g);
          double w0 = 1.0, w1 = 1.0;
          if (robust) {
            const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
            double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1);
          }
          double c0[6], c1[6];
          small_cam_rows(&l_cs[cur][ca][0], g, X.x, X.y, X.z, c0, c1);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const double wa0 = -w0 * c0[i], wa1 = -w1 * c1[i];          // W_a[i][k] = wa0 P[k] + wa1 P[3+k]
            const double W0 = wa0 * g.P[0] + wa1 * g.P[3], W1 = wa0 * g.P[1] + wa1 * g.P[4], W2 = wa0 * g.P[2] + wa1 * g.P[5];
            Ta[3 * i] = W0 * hinv[0] + W1 * hinv[1] + W2 * hinv[2];
            Ta[3 * i + 1] = W0 * hinv[1] + W1 * hinv[3] + W2 * hinv[4];
            Ta[3 * i + 2] = W0 * hinv[2] + W1 * hinv[4] + W2 * hinv[5];
          }
        }
        for (int jb = ja; jb < end; ++jb) {
          const int cb = A.p_cam[jb];
          if (cb == A.fixed_cam) continue;
          const double2 uv = A.p_uv[jb];
          Geom g;
          obs_geom(&l_cs[cur][cb][0], X.x, X.y, X.z, A.fx, A.fy, g);
          double w0 = 1.0, w1 = 1.0;
          if (robust) {
            const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
            double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1);
          }
          double c0[6], c1[6];
          small_cam_rows(&l_cs[cur][cb][0], g, X.x, X.y, X.z, c0, c1);
#pragma unroll
          for (int jj = 0; jj < 6; ++jj) {
            const double wb0 = -w0 * c0[jj], wb1 = -w1 * c1[jj];
            const double W0 = wb0 * g.P[0] + wb1 * g.P[3], W1 = wb0 * g.P[1] + wb1 * g.P[4], W2 = wb0 * g.P[2] + wb1 * g.P[5];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const double bij = Ta[3 * i] * W0 + Ta[3 * i + 1] * W1 + Ta[3 * i + 2] * W2;
              const long long v = -llrint(bij * scale);
              atomicAdd((unsigned long long*)&l_Sint[6 * ca + i][6 * cb + jj], (unsigned long long)v);
              if (jb != ja) atomicAdd((unsigned long long*)&l_Sint[6 * cb + jj][6 * ca + i], (unsigned long long)v);
            }
          }
        }
      }
    }
    if (need_lin) {                                     // max |bp| for the gtol test
      gmp = wave_nanmax(gmp);
      if (lane == 0) l_red[wv] = gmp;
    }
    __syncthreads();
    if (need_lin && tid == 0) {
      double m = s_gmax;
      for (int w = 0; w < SMALL_THREADS / 64; ++w) m = nanmax(m, l_red[w]);
      s_gmax = m;
      if (!isfinite(m)) { s_stop = 1; s_status = -4; }
      else if (A.gtol > 0 && m <= A.gtol) { s_stop = 1; s_status = 3; }
    }
    __syncthreads();
    if (s_stop) break;
    // ---- C2: W y0 per camera -> g = -(bc - W y0)
    if (wv < Nc) {
      const int c = wv;
      const double* cs = &l_cs[cur][c][0];
      double acc[6] = {0, 0, 0, 0, 0, 0};
      if (c != A.fixed_cam) {
        for (int i = A.offk[c * (NPART + 1)] + lane; i < A.offk[c * (NPART + 1) + NPART]; i += 64) {
          const int p = A.c_pt[i];
          const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
          const double2 uv = A.c_uv[i];
          Geom g;
          obs_geom(cs, X.x, X.y, X.z, A.fx, A.fy, g);
          double w0 = 1.0, w1 = 1.0;
          if (robust) {
            const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
            double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1);
          }
          double c0[6], c1[6];
          small_cam_rows(cs, g, X.x, X.y, X.z, c0, c1);
          const double y0 = A.y0[3 * (size_t)p], y1 = A.y0[3 * (size_t)p + 1], y2 = A.y0[3 * (size_t)p + 2];
          const double s0 = -(g.P[0] * y0 + g.P[1] * y1 + g.P[2] * y2) * w0;      // (Jp y) weighted
          const double s1 = -(g.P[3] * y0 + g.P[4] * y1 + g.P[5] * y2) * w1;
#pragma unroll
          for (int q = 0; q < 6; ++q) acc[q] += c0[q] * s0 + c1[q] * s1;
        }
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) acc[q] = wave_total_dpp(acc[q]);
      if (lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) l_g[6 * c + q] = (c == A.fixed_cam) ? 0.0 : -(l_bc[c][q] - acc[q]);
      }
    }
    __syncthreads();
    // ---- S = blockdiag(Hcc + lam D) + fixed(S); the fixed camera's rows / columns are the identity
    for (int t = tid; t < n * n; t += SMALL_THREADS) {
      const int i = t / n, j = t % n, ci = i / 6, cj = j / 6;
      double v = (double)l_Sint[i][j] / scale;
      if (ci == cj) {
        const int a = i % 6, b = j % 6;
        double d = l_Hcc[ci][S6(a, b)];
        if (a == b) d += lambda * fmax(d, DIAG_FLOOR);
        v += d;
      }
      if (ci == A.fixed_cam || cj == A.fixed_cam) v = (i == j) ? 1.0 : 0.0;
      l_S[i][j] = v;
    }
    __syncthreads();
    // ---- Cholesky (lower triangle in place), all threads
    for (int j = 0; j < n; ++j) {
      if (tid == 0) l_S[j][j] = sqrt(fmax(l_S[j][j], DIAG_FLOOR));
      __syncthreads();
      const double d = l_S[j][j];
      if (tid > j && tid < n) l_S[tid][j] /= d;
      __syncthreads();
      const int m = n - 1 - j;
      for (int t = tid; t < m * m; t += SMALL_THREADS) {
        const int ii = j + 1 + t / m, kk = j + 1 + t % m;
        if (kk <= ii) l_S[ii][kk] -= l_S[ii][j] * l_S[kk][j];
      }
      __syncthreads();
    }
    // ---- L y = g, L^T dc = y (wave 0; lane-parallel dot products)
    if (wv == 0) {
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = lane; k < i; k += 64) s += l_S[i][k] * l_y[k];
        s = wave_total_dpp(s);
        if (lane == 0) l_y[i] = (l_g[i] - s) / l_S[i][i];
      }
      for (int i = n - 1; i >= 0; --i) {
        double s = 0.0;
        for (int k = i + 1 + lane; k < n; k += 64) s += l_S[k][i] * l_dc[k];
        s = wave_total_dpp(s);
        if (lane == 0) l_dc[i] = (l_y[i] - s) / l_S[i][i];
      }
    }
    __syncthreads();
    // ---- camera update + camera-side scalars: g.d, sum D d^2, |d|^2, |x|^2
    if (tid < Nc) {
      const int c = tid;
      double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
      for (int q = 0; q < 6; ++q) {
        const double d = (c == A.fixed_cam) ? 0.0 : l_dc[6 * c + q];
        const double xq = l_cam[cur][c][q];
        l_cam[tr][c][q] = xq + d;
        a0 += l_bc[c][q] * d;
        a1 += fmax(l_Hcc[c][U6(q, q)], DIAG_FLOOR) * d * d;
        a2 += d * d;
        a3 += xq * xq;
      }
      l_camred[c][0] = a0; l_camred[c][1] = a1; l_camred[c][2] = a2; l_camred[c][3] = a3;
      camera_state(&l_cam[tr][c][0], &l_cs[tr][c][0]);
    }
    __syncthreads();
    // ---- P2: back substitution dp = -(y0 + Hinv W^T dc), trial points, point-side scalars
    double ps[4] = {0, 0, 0, 0};
    for (int p = tid; p < Np; p += SMALL_THREADS) {
      const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
      double u[3] = {0, 0, 0};
      for (int j = A.pt_off[p]; j < A.pt_off[p + 1]; ++j) {
        const int c = A.p_cam[j];
        if (c == A.fixed_cam) continue;
        const double2 uv = A.p_uv[j];
        Geom g;
        obs_geom(&l_cs[cur][c][0], X.x, X.y, X.z, A.fx, A.fy, g);
        double w0 = 1.0, w1 = 1.0;
        if (robust) {
          const double ru = uv.x - (g.xh * A.fx + A.cx), rv = uv.y - (g.yh * A.fy + A.cy);
          double t; huber(ru, A.hub_c, t, w0); huber(rv, A.hub_c, t, w1);
        }
        double c0[6], c1[6];
        small_cam_rows(&l_cs[cur][c][0], g, X.x, X.y, X.z, c0, c1);
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) { s0 += c0[q] * l_dc[6 * c + q]; s1 += c1[q] * l_dc[6 * c + q]; }
        s0 *= w0; s1 *= w1;
        u[0] -= g.P[0] * s0 + g.P[3] * s1;              // Jp^T w (Jc dc), Jp = -P
        u[1] -= g.P[1] * s0 + g.P[4] * s1;
        u[2] -= g.P[2] * s0 + g.P[5] * s1;
      }
      double hinv[6], yy[3];
#pragma unroll
      for (int q = 0; q < 6; ++q) hinv[q] = A.Hppinv[6 * (size_t)p + q];
      sym3_mul(hinv, u, yy);
      const double d0 = -(A.y0[3 * (size_t)p] + yy[0]), d1 = -(A.y0[3 * (size_t)p + 1] + yy[1]), d2 = -(A.y0[3 * (size_t)p + 2] + yy[2]);
      double* o = A.ptab[tr] + PT * (size_t)p;
      o[0] = X.x + d0; o[1] = X.y + d1; o[2] = X.z + d2;
      const double D0 = fmax(A.Hpp[6 * (size_t)p], DIAG_FLOOR), D1 = fmax(A.Hpp[6 * (size_t)p + 3], DIAG_FLOOR),
                   D2 = fmax(A.Hpp[6 * (size_t)p + 5], DIAG_FLOOR);
      ps[0] += A.bp[3 * (size_t)p] * d0 + A.bp[3 * (size_t)p + 1] * d1 + A.bp[3 * (size_t)p + 2] * d2;
      ps[1] += D0 * d0 * d0 + D1 * d1 * d1 + D2 * d2 * d2;
      ps[2] += d0 * d0 + d1 * d1 + d2 * d2;
      ps[3] += X.x * X.x + X.y * X.y + X.z * X.z;
    }
    small_block_sum<4>(ps, l_red, l_tot);
    __threadfence_block();
    __syncthreads();                                     // trial points visible to the camera-major cost pass
    cam_cost(tr);
    // ---- verdict (thread 0), the rules of ba_solve
    if (tid == 0) {
      double gTd = l_tot[0], dDd = l_tot[1], step2 = l_tot[2], x2 = l_tot[3];
      for (int c = 0; c < Nc; ++c) { gTd += l_camred[c][0]; dDd += l_camred[c][1]; step2 += l_camred[c][2]; x2 += l_camred[c][3]; }
      const double model = 0.5 * (lambda * dDd - gTd);
      const double cost_new = s_cost_new;
      const double rho = (model > 0.0 && isfinite(cost_new)) ? (s_cost - cost_new) / model : -1.0;
      const int it = ++s_it;
      ba_iter_record rec;
      rec.iteration = it; rec.accepted = (rho > 0.0 && isfinite(cost_new)) ? 1 : 0; rec.pcg_iterations = 0; rec.reserved = 0;
      rec.cost = s_cost; rec.cost_trial = cost_new; rec.sse_trial = s_sse_new; rec.lambda = lambda; rec.gain_ratio = rho;
      rec.step_norm = sqrt(step2); rec.seconds = 0.0;
      A.trace[it - 1] = rec;
      int stop = 0;
      if (rec.accepted) {
        const double dcost = s_cost - cost_new;
        s_cur = tr;
        s_cost = cost_new; s_sse = s_sse_new;
        ++s_acc;
        const double t = 2.0 * rho - 1.0;
        s_lambda = fmax(lambda * fmax(1.0 / 3.0, 1.0 - t * t * t), 1e-12);
        s_nu = 2.0;
        if (dcost <= A.ftol * cost_new) { s_status = 1; stop = 1; }
      } else {
        if (!isfinite(cost_new) && lambda >= 1e12) { s_status = -4; stop = 1; }
        s_lambda = fmin(lambda * s_nu, 1e12);
        s_nu *= 2.0;
      }
      if (!stop && sqrt(step2) <= A.xtol * (A.xtol + sqrt(x2))) { s_status = 2; stop = 1; }
      if (!stop && it >= A.max_iters) { s_status = 0; stop = 1; }
      s_stop = stop;
      l_tot[7] = rec.accepted ? 1.0 : 0.0;
    }
    __syncthreads();
    need_lin = l_tot[7] != 0.0;                          // a rejected step keeps the linearisation, only re-damps
    __syncthreads();
  }
  // ---- results: cameras of the accepted set back to global, summary
  __syncthreads();
  const int fin = s_cur;
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) A.cams[fin][6 * tid + q] = l_cam[fin][tid][q];
    for (int q = 0; q < CS; ++q) A.cs[fin][CS * tid + q] = l_cs[fin][tid][q];
  }
  if (tid == 0) {
    A.summary->iterations = s_it; A.summary->accepted = s_acc; A.summary->pcg_iterations = 0; A.summary->status = s_status;
    A.summary->final_sse = s_sse; A.summary->final_cost = s_cost; A.summary->final_lambda = s_lambda;
    *A.cur_out = fin;
  }
}

}  // namespace ba
