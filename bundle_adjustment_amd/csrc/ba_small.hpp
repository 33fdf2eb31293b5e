// Direct solver for sliding-window-sized problems: the reference's default use (BundleAdjuster(window_size=5),
// src/pipeline.py:39,99: a handful of keyframes, a few hundred landmarks).  At that size the multi-kernel LM / Schur /
// PCG loop is nothing but launch floors (174 launches at ~6 us for a 1.1 ms solve of a 2 k-observation window), while
// the reduced camera system is at most 48 x 48.  Here the WHOLE Levenberg-Marquardt loop runs in ONE workgroup of one
// kernel launch: same residual, same analytic blocks, same damping / gain-ratio / stopping rules as ba_solve, but the
// reduced system is formed explicitly and solved by dense Cholesky in LDS instead of PCG, and nothing returns to the
// host until the solve is over.
//
// The Schur complement as a dense product.  With Hpp^-1 = L L^T per point (3x3 Cholesky) and V = W L (W = Jc^T w Jp, the
// 6x3 camera / point coupling of one observation), every term the elimination needs is a product with the one matrix
//     V  (6 Nc rows) x (3 Np columns),  V[6c + i][k * Np_pad + p] = (W_{c,p} L_p)[i][k]   (zero where c does not see p),
//     stored in slabs of 16 columns x 64 rows (8 KB each, row-major inside): one MFMA operand load is 2 KB contiguous:
//     S   = blockdiag(Hcc + lam D) - V V^T                    v_mfma_f64_16x16x4_f64 tiles, K split over the 8 waves
//     g   = -(bc - V z),  z_p = L_p^T bp_p                     (W Hpp^-1 bp; z is row 6 Nc of V, so V z is a column of V V^T)
//     dp  = -(y0 + L_p (V^T dc)_p)                             (back substitution)
// This is the one place in the repository where the matrix cores fit: a genuinely dense symmetric rank-k update.  The
// fp64 MFMA rate equals the fp64 VALU rate on gfx950; what it removes is everything around the multiply (no atomics, no
// per-pair geometry, no cross-lane reductions), and its sums have a fixed order.
//
//   per LM iteration (all 512 threads, __syncthreads between phases):
//     C1  camera-major, 8 / Nc waves per camera: Hcc (21) | bc (6) by DPP wave sums              (linearise, camera half)
//     P1  point-major, thread = point: Hpp, bp, damped inverse -> L, y0, z; the point's columns of V
//     G   V V^T: wave w takes the 16-column slabs w, w + 8, ...; partial tiles to global; summed in wave order
//     S and g from the tiles; Cholesky of S and the two triangular solves in ONE wave (lane = row, no workgroup barriers)
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
constexpr int SMALL_THREADS = 512;
constexpr int SMALL_DEFAULT_MAX_OBS = 6144;      // ba_solve dispatches here up to this many observations (BA_SMALL_MAX_OBS);
                                                  // measured crossover with the multi-kernel path: ~6 k (tools/small_crossover.py)
constexpr int SMALL_WAVES = SMALL_THREADS / 64;
constexpr int SMALL_VROWS = 64;                   // rows of V: 6 Nc camera rows, then z = L^T bp, zero up to the tile edge
constexpr int SMALL_TILES = 9;                    // 16x16 tiles (ti <= tj) of the 49 x 49 product that are needed

typedef double small_d4 __attribute__((ext_vector_type(4)));

// All 18 values in registers HERE: the loads that produce them are issued back to back and waited for once.  (Left to
// itself the scheduler of this very large kernel serialises load -> wait -> use, one L2 round trip per value.)
__device__ inline void small_pin18(double (&v)[18]) {
  asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                    "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]), "+v"(v[16]), "+v"(v[17]));
}

// offset of element (row 0, column k) of V; row r of the same column is 16 r further on
__device__ inline size_t small_vcol(int k) { return ((size_t)(k >> 4) * SMALL_VROWS + 0) * 16 + (k & 15); }


struct SmallArgs {
  double* cams[2]; double* cs[2]; double* ptab[2]; double* camA[2];
  const int* offk; const int* c_pt; const double2* c_uv;
  const int* pt_off; const int* p_cam; const double2* p_uv;
  double* Hpp; double* bp; double* Lf; double* y0;         // per point: undamped Hpp (6), bp (3), L of the damped inverse (6), y0 (3)
  double* V;                                                 // Kp / 16 slabs of [SMALL_VROWS][16], zero outside what P1 writes
  double* gS;                                                // [waves][SMALL_TILES][64 lanes x 4]: partial V V^T
  int Kp, Np_pad;                                            // Kp = 3 Np_pad, Np_pad a multiple of 16
  int n_cams, n_pts, fixed_cam, robust;
  double fx, fy, cx, cy, hub_c;
  int max_iters; double ftol, xtol, gtol, lambda0;
  int cur;                                                   // which parameter set holds the start point
  ba_summary* summary; ba_iter_record* trace; int* cur_out;      // host-mapped
  long long* stamps;                                               // diagnostic (BA_SMALL_STAMPS): 100 MHz clock at the phase boundaries of LM iteration 2
  long long* host_flag; long long seq;                             // published (system scope) when the results are written
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
    for (int w = 0; w < SMALL_WAVES; ++w) a += sm[w * N + threadIdx.x];
    out[threadIdx.x] = a;
  }
  __syncthreads();
}

// [V; z] [V; z]^T on the matrix cores, the upper 16x16 tiles (ti <= tj < NT) of it.  Lane l of a wave feeds row
// 16 t + l % 16 and, from one 32-byte load, the four columns 16 s + 4 (l / 16) + e of slab s; MFMA step e multiplies
// column e of every lane group -- the same permutation of the sum over columns on both operands, so the product is
// unchanged.  Wave w takes slabs w, w + W, ...; two slabs' operands are loaded ahead of their 2 x 4 steps.  The partial
// tiles go to gS[w][tile tj (tj + 1) / 2 + ti][lane][4]; with NT == 4 only the z row lives in tile row 3 and its own
// square is skipped.
template <int NT>
__device__ inline void small_syrk(const double* __restrict__ V, int Kp, double* __restrict__ gS, int wv, int lane) {
  constexpr int NTILE = NT > 3 ? 9 : NT * (NT + 1) / 2;
  small_d4 acc[NTILE];
#pragma unroll
  for (int q = 0; q < NTILE; ++q) acc[q] = (small_d4){0.0, 0.0, 0.0, 0.0};
  const double* vrow = V + (lane & 15) * 16 + 4 * (lane >> 4);
  const int nslab = Kp >> 4;
  constexpr size_t SLAB = (size_t)SMALL_VROWS * 16;
  const small_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  auto load_pair = [&](int s0, small_d4 (&dst)[2][NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dst[0][t] = s0 < nslab ? *(const small_d4*)(vrow + SLAB * s0 + 256 * t) : zero4;
      dst[1][t] = s0 + SMALL_WAVES < nslab ? *(const small_d4*)(vrow + SLAB * (s0 + SMALL_WAVES) + 256 * t) : zero4;
    }
  };
  auto multiply = [&](const small_d4 (&a)[2][NT]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int tj = 0; tj < NT; ++tj) {
#pragma unroll
          for (int ti = 0; ti <= tj; ++ti) {
            if (NT > 3 && ti == 3) continue;
            acc[tj * (tj + 1) / 2 + ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][ti][e], a[u][tj][e], acc[tj * (tj + 1) / 2 + ti], 0, 0, 0);
          }
        }
      }
    }
  };
  small_d4 a[2][NT];                                       // (keeping the next pair in flight as well measured slower)
  for (int sl = wv; sl < nslab; sl += 2 * SMALL_WAVES) {
    load_pair(sl, a);
    multiply(a);
  }
  small_d4* out = (small_d4*)gS + (size_t)(wv * SMALL_TILES) * 64 + lane;
#pragma unroll
  for (int q = 0; q < NTILE; ++q) out[q * 64] = acc[q];
}

#define SMALL_STAMP(k) do { if (A.stamps && tid == 0 && s_it == 1) { A.stamps[k] = (long long)wall_clock64(); if ((k) == 0 || (k) == 8) A.stamps[9 + (k) / 8] = (long long)clock64(); } } while (0)

// One workgroup of 8 waves is all that ever runs: tell the compiler that 2 waves per SIMD is the occupancy to schedule for,
// or it trades the instruction-level parallelism of every phase for registers nobody else will use.
__global__ void __launch_bounds__(SMALL_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_small_lm(SmallArgs A) {
  __shared__ double l_cs[2][SMALL_MAX_CAMS][CS];
  __shared__ double l_cam[2][SMALL_MAX_CAMS][6];
  __shared__ double l_Hcc[SMALL_MAX_CAMS][21], l_bc[SMALL_MAX_CAMS][6];
  __shared__ double l_S[SMALL_N][SMALL_N + 1];
  __shared__ double l_g[SMALL_N], l_dc[SMALL_N];
  __shared__ double l_wpart[SMALL_WAVES][27];               // per-wave partial sums of the camera-major passes
  __shared__ double l_red[SMALL_WAVES * 8], l_tot[8], l_camred[SMALL_MAX_CAMS][4];
  __shared__ double s_lambda, s_cost, s_sse, s_cost_new, s_sse_new, s_gmax;
  __shared__ int s_cur, s_stop, s_it, s_acc, s_status;
  __shared__ double s_nu;

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in a scalar register
  const int Nc = A.n_cams, Np = A.n_pts, n = 6 * Nc;
  const int Kp = A.Kp, Npp = A.Np_pad;
  const bool robust = A.robust != 0;
  // camera-major passes: wpc waves share one camera (lanes stride 64 wpc), partial sums combined in wave order
  const int wpc = Nc <= 1 ? SMALL_WAVES : Nc <= 2 ? SMALL_WAVES / 2 : Nc <= 4 ? SMALL_WAVES / 4 : SMALL_WAVES / 8;
  const int my_cam = wv / wpc, my_sub = wv % wpc;

  // ---- start point into LDS
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) l_cam[A.cur][tid][q] = A.cams[A.cur][6 * tid + q];
    camera_state(&l_cam[A.cur][tid][0], &l_cs[A.cur][tid][0]);
  }
  if (tid == 0) { s_cur = A.cur; s_lambda = A.lambda0; s_nu = 2.0; s_stop = 0; s_it = 0; s_acc = 0; s_status = 0; }
  __syncthreads();

  // camera-major cost at parameter set w: sse, rho-sum -> s_sse_new, s_cost_new (thread 0 combines the waves in order)
  auto cam_cost = [&](int w) {
    if (my_cam < Nc) {
      const int c = my_cam;
      const double* cs = &l_cs[w][c][0];
      double acc[2] = {0.0, 0.0};
      for (int i = A.offk[c * (NPART + 1)] + my_sub * 64 + lane; i < A.offk[c * (NPART + 1) + NPART]; i += 64 * wpc) {
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
      if (lane == 0) { l_wpart[wv][0] = acc[0]; l_wpart[wv][1] = acc[1]; }
    }
    __syncthreads();
    if (tid == 0) {
      double a = 0.0, b = 0.0;
      for (int ww = 0; ww < Nc * wpc; ++ww) { a += l_wpart[ww][0]; b += l_wpart[ww][1]; }
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
    SMALL_STAMP(0);
    if (need_lin) {
      // ---- C1: camera half of the normal equations (post-M rows: no congruence afterwards)
      if (my_cam < Nc) {
        const int c = my_cam;
        const double* cs = &l_cs[cur][c][0];
        double acc[27];
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        if (c != A.fixed_cam) {
          for (int i = A.offk[c * (NPART + 1)] + my_sub * 64 + lane; i < A.offk[c * (NPART + 1) + NPART]; i += 64 * wpc) {
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
          for (int q = 0; q < 27; ++q) l_wpart[wv][q] = acc[q];
        }
      }
      __syncthreads();
      if (tid < Nc * 27) {
        const int c = tid / 27, q = tid % 27;
        double a = 0.0;
        for (int sb = 0; sb < wpc; ++sb) a += l_wpart[c * wpc + sb][q];
        if (q < 21) l_Hcc[c][q] = a; else l_bc[c][q - 21] = a;
      }
      __syncthreads();
      if (tid == 0) {
        double gm = 0.0;
        for (int c = 0; c < Nc; ++c)
          for (int i = 0; i < 6; ++i) gm = nanmax(gm, fabs(l_bc[c][i]));
        s_gmax = gm;
      }
    }
    SMALL_STAMP(1);
    // ---- P1: point half, damped inverse = L L^T, y0, z = L^T bp, the point's columns of V = W L
    double gmp = 0.0;
    for (int p = tid; p < Np; p += SMALL_THREADS) {
      const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
      const int beg = A.pt_off[p], end = A.pt_off[p + 1];
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
      double h[6], hinv[6], b3[3], y[3];
#pragma unroll
      for (int q = 0; q < 6; ++q) h[q] = A.Hpp[6 * (size_t)p + q];
#pragma unroll
      for (int q = 0; q < 3; ++q) b3[q] = A.bp[3 * (size_t)p + q];
      h[0] += lambda * fmax(h[0], DIAG_FLOOR);
      h[3] += lambda * fmax(h[3], DIAG_FLOOR);
      h[5] += lambda * fmax(h[5], DIAG_FLOOR);
      sym3_inverse(h, hinv);
      sym3_mul(hinv, b3, y);
      // hinv = L L^T, L lower triangular: l00 | l10 l11 | l20 l21 l22
      const double l00 = sqrt(fmax(hinv[0], 1e-300));
      const double l10 = hinv[1] / l00, l20 = hinv[2] / l00;
      const double l11 = sqrt(fmax(hinv[3] - l10 * l10, 1e-300));
      const double l21 = (hinv[4] - l20 * l10) / l11;
      const double l22 = sqrt(fmax(hinv[5] - l20 * l20 - l21 * l21, 1e-300));
      double* Lf = A.Lf + 6 * (size_t)p;
      Lf[0] = l00; Lf[1] = l10; Lf[2] = l11; Lf[3] = l20; Lf[4] = l21; Lf[5] = l22;
#pragma unroll
      for (int q = 0; q < 3; ++q) A.y0[3 * (size_t)p + q] = y[q];
      double* V0 = A.V + small_vcol(p), *V1 = A.V + small_vcol(Npp + p), *V2 = A.V + small_vcol(2 * Npp + p);   // the point's 3 columns
      V0[16 * n] = l00 * b3[0] + l10 * b3[1] + l20 * b3[2];
      V1[16 * n] = l11 * b3[1] + l21 * b3[2];
      V2[16 * n] = l22 * b3[2];
      unsigned seen = 0;
      for (int j = beg; j < end; ++j) {
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
        const bool again = (seen >> c) & 1u;             // a second observation of this point by the same camera adds up
        seen |= 1u << c;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double wa0 = -w0 * c0[i], wa1 = -w1 * c1[i];            // W[i][k] = wa0 P[k] + wa1 P[3 + k]
          const double W0 = wa0 * g.P[0] + wa1 * g.P[3], W1 = wa0 * g.P[1] + wa1 * g.P[4], W2 = wa0 * g.P[2] + wa1 * g.P[5];
          double v0 = W0 * l00 + W1 * l10 + W2 * l20, v1 = W1 * l11 + W2 * l21, v2 = W2 * l22;
          const int r = 16 * (6 * c + i);
          if (again) { v0 += V0[r]; v1 += V1[r]; v2 += V2[r]; }
          V0[r] = v0; V1[r] = v1; V2[r] = v2;
        }
      }
      for (int c = 0; c < Nc; ++c) {
        if ((seen >> c) & 1u) continue;
#pragma unroll
        for (int i = 0; i < 6; ++i) { const int r = 16 * (6 * c + i); V0[r] = 0.0; V1[r] = 0.0; V2[r] = 0.0; }
      }
    }
    if (need_lin) {                                     // max |bp| for the gtol test
      gmp = wave_nanmax(gmp);
      if (lane == 0) l_red[wv] = gmp;
    }
    __syncthreads();
    if (need_lin && tid == 0) {
      double m = s_gmax;
      for (int w = 0; w < SMALL_WAVES; ++w) m = nanmax(m, l_red[w]);
      s_gmax = m;
      if (!isfinite(m)) { s_stop = 1; s_status = -4; }
      else if (A.gtol > 0 && m <= A.gtol) { s_stop = 1; s_status = 3; }
    }
    __syncthreads();
    if (s_stop) break;
    SMALL_STAMP(2);
    // ---- G: [V; z] [V; z]^T on the matrix cores (small_syrk)
    switch ((n + 16) >> 4) {                               // tile rows incl. the z row
      case 1: small_syrk<1>(A.V, Kp, A.gS, wv, lane); break;
      case 2: small_syrk<2>(A.V, Kp, A.gS, wv, lane); break;
      case 3: small_syrk<3>(A.V, Kp, A.gS, wv, lane); break;
      default: small_syrk<4>(A.V, Kp, A.gS, wv, lane); break;
    }
    __syncthreads();
    SMALL_STAMP(3);
    // ---- S = blockdiag(Hcc + lam D) - sum of the waves' partial V V^T (wave order); fixed camera: identity rows / columns;
    //      g = -(bc - V z) from column 6 Nc of the product.
    //      tile (ti <= tj) number tj (tj + 1) / 2 + ti; element (i, j) of a tile sits in lane j + 16 (i % 4), register i / 4
    for (int t = tid; t < n * (n + 1); t += SMALL_THREADS) {
      const int i = t / (n + 1), j = t % (n + 1), ci = i / 6, cj = j / 6;
      const int ii = i < j ? i : j, jj = i < j ? j : i;
      const int ti = ii >> 4, tj = jj >> 4, ri = ii & 15, rj = jj & 15;
      const double* ps = A.gS + (size_t)(tj * (tj + 1) / 2 + ti) * 256 + 4 * (rj + 16 * (ri & 3)) + (ri >> 2);
      double vv = 0.0;
#pragma unroll
      for (int w = 0; w < SMALL_WAVES; ++w) vv += ps[(size_t)w * SMALL_TILES * 256];
      if (j == n) {
        l_g[i] = (ci == A.fixed_cam) ? 0.0 : -(l_bc[ci][i % 6] - vv);
        continue;
      }
      double v = -vv;
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
    SMALL_STAMP(4);
    // ---- Cholesky and both triangular solves in wave 0, lane = row, the row of L in REGISTERS: column j is formed from
    //      the finished columns k < j (L[j][k] comes from lane j by v_readlane), so the factorisation touches neither LDS
    //      nor a barrier; the loops are fully unrolled so that every register index is static.  The diagonal holds 1 / L[j][j].
    if (wv == 0) {
      double row[SMALL_N];
      int ln = lane;                                       // opaque per LM iteration: keeps the compiler from hoisting 2 x 48
      asm volatile("" : "+v"(ln));                        // lane-compare masks out of the loop into (spilled) scalar registers
      const int li = ln < n ? ln : 0;                  // lanes past the matrix shadow row 0; their values are never used
#pragma unroll
      for (int k = 0; k < SMALL_N; ++k) row[k] = k < n ? l_S[li][k] : 0.0;
#pragma unroll
      for (int j = 0; j < SMALL_N; ++j) {
        if (j < n) {
          double sa[4] = {row[j], 0.0, 0.0, 0.0};            // four partial sums: the chain of dependent FMAs is j / 4 long
#pragma unroll
          for (int k = 0; k < j; ++k) sa[k & 3] -= row[k] * readlane_f64(row[k], j);
          const double sacc = (sa[0] + sa[1]) + (sa[2] + sa[3]);
          const double d = fmax(readlane_f64(sacc, j), DIAG_FLOOR);
          double inv = __builtin_amdgcn_rsq(d);              // 1 / sqrt(d): hardware estimate + two Newton steps
          inv = inv * (1.5 - 0.5 * d * inv * inv);
          inv = inv * (1.5 - 0.5 * d * inv * inv);
          row[j] = ln == j ? inv : sacc * inv;
        }
      }
      SMALL_STAMP(13);
      // L to LDS (lower triangle, diagonal = 1 / L[j][j]): the backward substitution reads COLUMNS of it, one lane per row
#pragma unroll
      for (int k = 0; k < SMALL_N; ++k)
        if (k < n && ln < n) l_S[ln][k] = row[k];
      double b = ln < n ? l_g[ln] : 0.0, y = 0.0, x = 0.0;
#pragma unroll
      for (int j = 0; j < SMALL_N; ++j) {                  // L y = g, column-oriented
        if (j < n) {
          const double yj = readlane_f64(b, j) * readlane_f64(row[j], j);
          if (ln == j) y = yj;
          else if (ln > j) b -= row[j] * yj;
        }
      }
      SMALL_STAMP(14);
      for (int j = n - 1; j >= 0; --j) {                   // L^T dc = y, column-oriented over the rows of L in LDS
        const double lji = ln < j ? l_S[j][ln] : 0.0;
        const double xj = readlane_f64(y, j) * l_S[j][j];
        if (ln == j) x = xj;
        y -= lji * xj;
      }
      if (lane < n) l_dc[lane] = x;
    }
    __syncthreads();
    SMALL_STAMP(5);
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
    SMALL_STAMP(6);
    // ---- P2: back substitution dp = -(y0 + L (V^T dc)), trial points, point-side scalars
    double ps[4] = {0, 0, 0, 0};
    for (int p = tid; p < Np; p += SMALL_THREADS) {
      const double4 X = *(const double4*)(A.ptab[cur] + PT * (size_t)p);
      const double* V0 = A.V + small_vcol(p), *V1 = A.V + small_vcol(Npp + p), *V2 = A.V + small_vcol(2 * Npp + p);
      double t0 = 0.0, t1 = 0.0, t2 = 0.0;
      for (int c = 0; c < Nc; ++c) {                       // one camera's 6 rows x 3 columns per batch of loads
        double v[18];
#pragma unroll
        for (int i = 0; i < 6; ++i) { v[3 * i] = V0[16 * (6 * c + i)]; v[3 * i + 1] = V1[16 * (6 * c + i)]; v[3 * i + 2] = V2[16 * (6 * c + i)]; }
        small_pin18(v);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const double d = l_dc[6 * c + i];
          t0 += v[3 * i] * d; t1 += v[3 * i + 1] * d; t2 += v[3 * i + 2] * d;
        }
      }
      const double* Lf = A.Lf + 6 * (size_t)p;
      const double d0 = -(A.y0[3 * (size_t)p] + Lf[0] * t0);
      const double d1 = -(A.y0[3 * (size_t)p + 1] + Lf[1] * t0 + Lf[2] * t1);
      const double d2 = -(A.y0[3 * (size_t)p + 2] + Lf[3] * t0 + Lf[4] * t1 + Lf[5] * t2);
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
    SMALL_STAMP(7);
    cam_cost(tr);
    SMALL_STAMP(8);
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
  // ---- results: cameras of the accepted set back to global (state + packed table, as k_cam_prepare leaves them), summary
  __syncthreads();
  const int fin = s_cur;
  if (tid < Nc) {
    for (int q = 0; q < 6; ++q) A.cams[fin][6 * tid + q] = l_cam[fin][tid][q];
    for (int q = 0; q < CS; ++q) A.cs[fin][CS * tid + q] = l_cs[fin][tid][q];
    for (int q = 0; q < 12; ++q) A.camA[fin][TA * tid + q] = l_cs[fin][tid][q];
  }
  if (tid == 0) {
    A.summary->iterations = s_it; A.summary->accepted = s_acc; A.summary->pcg_iterations = 0; A.summary->status = s_status;
    A.summary->final_sse = s_sse; A.summary->final_cost = s_cost; A.summary->final_lambda = s_lambda;
    *A.cur_out = fin;
    publish_flag(A.host_flag, A.seq, 1);
  }
}

}  // namespace ba
