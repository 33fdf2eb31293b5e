// HIP kernels of the bundle-adjustment solve step (gfx950).  Two orderings of the
// observation list exist on the device (built once in ba_set_problem):
//   camera order: observations of camera c are [cam_off[c], cam_off[c+1])  -> c_pt, c_uv
//   point order:  observations of point  p are [pt_off[p],  pt_off[p+1])   -> p_cam, p_uv
// "cam" kernels run one workgroup per camera (pose state uniform -> scalar registers,
// block reduction of the per-camera sums), "pt" kernels one thread per point (point
// state in registers, serial sum over its few observations).  All sums have a fixed
// order, so results are bitwise reproducible run to run.
#pragma once
#include "ba_device.hpp"

namespace ba {

constexpr int CAM_BLOCK = 256;   // threads per camera workgroup
constexpr int PT_BLOCK = 128;    // threads (= points) per point workgroup
constexpr int VEC_BLOCK = 64;    // threads (= cameras) per workgroup in camera-vector kernels

// PCG device state, two copies indexed by iteration parity (see k_pcg_step)
struct PcgState {
  double gamma_prev, alpha_prev, gamma0, pad0;
  int done, iters, flag, pad1;     // done: 1 converged, 2 breakdown
};

// -------------------------------------------------------------------------------------
__global__ void k_cam_prepare(const double* __restrict__ cams, double* __restrict__ cs, int n_cams) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n_cams) camera_state(cams + 6 * c, cs + CS * c);
}

// K1: residuals / cost at (cs, pts).  One workgroup per camera.
//   r_out (nullable): residual pairs scattered to the caller's observation order.
//   part[c][0] = sum r^2, part[c][1] = sum rho-term over camera c's observations.
template <bool ROBUST>
__global__ void __launch_bounds__(CAM_BLOCK)
k_residual_cam(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ cam_off,
               const int* __restrict__ c_pt, const double2* __restrict__ c_uv, const int* __restrict__ c_orig,
               double fx, double fy, double cx, double cy, double hub_c,
               double* __restrict__ r_out, double* __restrict__ part) {
  __shared__ double sm[2 * (CAM_BLOCK / 64)];
  const int c = blockIdx.x;
  const double* cam = cs + CS * c;
  const int beg = cam_off[c], end = cam_off[c + 1];
  double acc[2] = {0.0, 0.0};
  for (int i = beg + threadIdx.x; i < end; i += CAM_BLOCK) {
    const int p = c_pt[i];
    const double2 uv = c_uv[i];
    double xh, yh;
    obs_project(cam, pts[3 * p], pts[3 * p + 1], pts[3 * p + 2], xh, yh);
    const double ru = uv.x - (xh * fx + cx);
    const double rv = uv.y - (yh * fy + cy);
    acc[0] += ru * ru + rv * rv;
    if (ROBUST) {
      double t0, t1, w;
      huber(ru, hub_c, t0, w);
      huber(rv, hub_c, t1, w);
      acc[1] += t0 + t1;
    }
    if (r_out) {
      const int o = c_orig[i];
      r_out[2 * (size_t)o] = ru;
      r_out[2 * (size_t)o + 1] = rv;
    }
  }
  block_sum<2>(acc, sm);
  if (threadIdx.x == 0) {
    part[2 * c] = acc[0];
    part[2 * c + 1] = ROBUST ? acc[1] : acc[0];
  }
}

// K2a: camera half of the normal equations.  One workgroup per camera:
//   Hcc[c] (21, packed upper) = sum Jc^T w Jc,  bc[c] (6) = sum Jc^T w r
// with Jc = [P [X]x M | -dpi].  The sums are taken over A = P [X]x (pre-M) and the
// M^T (.) M congruence is applied once per camera after the block reduction.
// Also stores the IRLS weights of camera-ordered observations (c_w) when ROBUST.
template <bool ROBUST>
__global__ void __launch_bounds__(CAM_BLOCK)
k_linearize_cam(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ cam_off,
                const int* __restrict__ c_pt, const double2* __restrict__ c_uv,
                double fx, double fy, double cx, double cy, double hub_c, int fixed_cam,
                double* __restrict__ Hcc, double* __restrict__ bc, double2* __restrict__ c_w) {
  __shared__ double sm[27 * (CAM_BLOCK / 64)];
  const int c = blockIdx.x;
  const double* cam = cs + CS * c;
  const int beg = cam_off[c], end = cam_off[c + 1];
  double acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  for (int i = beg + threadIdx.x; i < end; i += CAM_BLOCK) {
    const int p = c_pt[i];
    const double2 uv = c_uv[i];
    const double X0 = pts[3 * p], X1 = pts[3 * p + 1], X2 = pts[3 * p + 2];
    Geom g;
    obs_geom(cam, X0, X1, X2, fx, fy, g);
    const double ru = uv.x - (g.xh * fx + cx);
    const double rv = uv.y - (g.yh * fy + cy);
    double w0 = 1.0, w1 = 1.0;
    if (ROBUST) {
      double t;
      huber(ru, hub_c, t, w0);
      huber(rv, hub_c, t, w1);
      c_w[i] = make_double2(w0, w1);
    }
    // rows of the pre-M camera Jacobian: J0 = [P0 x X | -d00 0 -d02], J1 = [P1 x X | 0 -d11 -d12]
    double J0[6], J1[6];
    J0[0] = g.P[1] * X2 - g.P[2] * X1; J0[1] = g.P[2] * X0 - g.P[0] * X2; J0[2] = g.P[0] * X1 - g.P[1] * X0;
    J1[0] = g.P[4] * X2 - g.P[5] * X1; J1[1] = g.P[5] * X0 - g.P[3] * X2; J1[2] = g.P[3] * X1 - g.P[4] * X0;
    J0[3] = -g.d00; J0[4] = 0.0;    J0[5] = -g.d02;
    J1[3] = 0.0;    J1[4] = -g.d11; J1[5] = -g.d12;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const double wa0 = w0 * J0[a], wa1 = w1 * J1[a];
#pragma unroll
      for (int b = a; b < 6; ++b) acc[U6(a, b)] += wa0 * J0[b] + wa1 * J1[b];
      acc[21 + a] += wa0 * ru + wa1 * rv;
    }
  }
  block_sum<27>(acc, sm);
  if (threadIdx.x == 0) {
    double* H = Hcc + 21 * c;
    double* b = bc + 6 * c;
    if (c == fixed_cam) {
      for (int k = 0; k < 21; ++k) H[k] = 0.0;
      for (int k = 0; k < 6; ++k) b[k] = 0.0;
      return;
    }
    const double* M = cam + 12;
    // full symmetric A (pre-M) then H = T^T A T with T = diag(M, I)
    double A[6][6], B[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) A[i][j] = acc[S6(i, j)];
    // B = A T : columns 0..2 mixed by M, 3..5 unchanged
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 3; ++j) B[i][j] = A[i][0] * M[j] + A[i][1] * M[3 + j] + A[i][2] * M[6 + j];
      for (int j = 3; j < 6; ++j) B[i][j] = A[i][j];
    }
    // H = T^T B : rows 0..2 mixed by M^T
    for (int j = 0; j < 6; ++j) {
      double h0 = M[0] * B[0][j] + M[3] * B[1][j] + M[6] * B[2][j];
      double h1 = M[1] * B[0][j] + M[4] * B[1][j] + M[7] * B[2][j];
      double h2 = M[2] * B[0][j] + M[5] * B[1][j] + M[8] * B[2][j];
      A[0][j] = h0; A[1][j] = h1; A[2][j] = h2;
      A[3][j] = B[3][j]; A[4][j] = B[4][j]; A[5][j] = B[5][j];
    }
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) H[U6(i, j)] = A[i][j];
    const double g0 = acc[21], g1 = acc[22], g2 = acc[23];
    b[0] = M[0] * g0 + M[3] * g1 + M[6] * g2;
    b[1] = M[1] * g0 + M[4] * g1 + M[7] * g2;
    b[2] = M[2] * g0 + M[5] * g1 + M[8] * g2;
    b[3] = acc[24]; b[4] = acc[25]; b[5] = acc[26];
  }
}

// K2b: point half.  One thread per point: Hpp[p] (6) = sum P^T w P, bp[p] (3) = -sum P^T w r,
// and the IRLS weights of point-ordered observations (p_w) when ROBUST.
template <bool ROBUST>
__global__ void __launch_bounds__(PT_BLOCK)
k_linearize_pt(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ pt_off,
               const int* __restrict__ p_cam, const double2* __restrict__ p_uv,
               double fx, double fy, double cx, double cy, double hub_c, int n_pts,
               double* __restrict__ Hpp, double* __restrict__ bp, double2* __restrict__ p_w) {
  const int p = blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= n_pts) return;
  const double X0 = pts[3 * p], X1 = pts[3 * p + 1], X2 = pts[3 * p + 2];
  double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  const int beg = pt_off[p], end = pt_off[p + 1];
  for (int j = beg; j < end; ++j) {
    const int c = p_cam[j];
    const double2 uv = p_uv[j];
    Geom g;
    obs_geom(cs + CS * c, X0, X1, X2, fx, fy, g);
    const double ru = uv.x - (g.xh * fx + cx);
    const double rv = uv.y - (g.yh * fy + cy);
    double w0 = 1.0, w1 = 1.0;
    if (ROBUST) {
      double t;
      huber(ru, hub_c, t, w0);
      huber(rv, hub_c, t, w1);
      p_w[j] = make_double2(w0, w1);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double wa0 = w0 * g.P[a], wa1 = w1 * g.P[3 + a];
#pragma unroll
      for (int bb = a; bb < 3; ++bb) h[U3(a, bb)] += wa0 * g.P[bb] + wa1 * g.P[3 + bb];
      b[a] -= wa0 * ru + wa1 * rv;       // Jp = -P
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) Hpp[6 * (size_t)p + k] = h[k];
#pragma unroll
  for (int k = 0; k < 3; ++k) bp[3 * (size_t)p + k] = b[k];
}

// K3: damped 3x3 inverse per point and y0 = (Hpp + lam Dp)^-1 bp.
__global__ void __launch_bounds__(PT_BLOCK)
k_point_invert(const double* __restrict__ Hpp, const double* __restrict__ bp, double lambda, int n_pts,
               double* __restrict__ Hppinv, double* __restrict__ y0) {
  const int p = blockIdx.x * PT_BLOCK + threadIdx.x;
  if (p >= n_pts) return;
  double h[6], inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) h[k] = Hpp[6 * (size_t)p + k];
  h[0] += lambda * fmax(h[0], DIAG_FLOOR);
  h[3] += lambda * fmax(h[3], DIAG_FLOOR);
  h[5] += lambda * fmax(h[5], DIAG_FLOOR);
  sym3_inverse(h, inv);
#pragma unroll
  for (int k = 0; k < 6; ++k) Hppinv[6 * (size_t)p + k] = inv[k];
  const double b[3] = {bp[3 * (size_t)p], bp[3 * (size_t)p + 1], bp[3 * (size_t)p + 2]};
  double y[3];
  sym3_mul(inv, b, y);
  y0[3 * (size_t)p] = y[0]; y0[3 * (size_t)p + 1] = y[1]; y0[3 * (size_t)p + 2] = y[2];
}

// Damped camera blocks Hccd = Hcc + lam * max(diag, floor); fixed camera -> identity.
__global__ void k_damp_cameras(const double* __restrict__ Hcc, double lambda, int n_cams, int fixed_cam,
                               double* __restrict__ Hccd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  double h[21];
  for (int k = 0; k < 21; ++k) h[k] = Hcc[21 * c + k];
  if (c == fixed_cam) {
    for (int k = 0; k < 21; ++k) h[k] = 0.0;
    for (int i = 0; i < 6; ++i) h[U6(i, i)] = 1.0;
  } else {
    for (int i = 0; i < 6; ++i) h[U6(i, i)] += lambda * fmax(h[U6(i, i)], DIAG_FLOOR);
  }
  for (int k = 0; k < 21; ++k) Hccd[21 * c + k] = h[k];
}

// PCG convergence test shared by the three kernels of an iteration: every workgroup
// evaluates it from the same device words, so all of them take the same branch.
__device__ inline bool pcg_finished(int k, const PcgState* __restrict__ st, const double* __restrict__ partV,
                                    int nblkV, double tol2, int min_iters, double& gamma, double& zeta) {
  const PcgState& s = st[k & 1];
  const double* pv = partV + (size_t)(k & 1) * 2 * nblkV;
  double g = 0, z = 0;
  for (int b = 0; b < nblkV; ++b) { g += pv[2 * b]; z += pv[2 * b + 1]; }
  gamma = g; zeta = z;
  if (s.done) return true;
  const double g0 = (k == 0) ? g : s.gamma0;
  if (!(g > 0.0)) return true;
  return (k >= min_iters && g <= tol2 * g0);
}

// K4a / K6: point pass.  One thread per point:
//   u = sum_o Jp^T w (Jc v_c),  Jc v = P (X x vt_r) - dpi vt_t,  vt = (M v_r, v_t)
// MODE 0 (PCG):  y[p] = Hppinv u, partA[block] = sum u.y          (early exit when PCG is done)
// MODE 1 (back substitution): dp = -(y0 + Hppinv u), pts_trial = pts + dp and the
//         point-side scalars of the gain ratio / stopping tests -> partB[block][4]
template <bool ROBUST, int MODE>
__global__ void __launch_bounds__(PT_BLOCK)
k_schur_pt(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ pt_off,
           const int* __restrict__ p_cam, const double2* __restrict__ p_w, const double* __restrict__ vtil,
           const double* __restrict__ Hppinv, double fx, double fy, int n_pts, int fixed_cam,
           double* __restrict__ y, double* __restrict__ partA,
           // MODE 0
           int k, const PcgState* __restrict__ st, const double* __restrict__ partV, int nblkV, double tol2,
           int min_iters,
           // MODE 1
           const double* __restrict__ y0, const double* __restrict__ Hpp, const double* __restrict__ bp,
           double* __restrict__ pts_trial, double* __restrict__ partB) {
  __shared__ double sm[4 * (PT_BLOCK / 64)];
  if (MODE == 0) {
    double g, z;
    if (pcg_finished(k, st, partV, nblkV, tol2, min_iters, g, z)) return;
  }
  const int p = blockIdx.x * PT_BLOCK + threadIdx.x;
  double acc[4] = {0, 0, 0, 0};
  if (p < n_pts) {
    const double X0 = pts[3 * (size_t)p], X1 = pts[3 * (size_t)p + 1], X2 = pts[3 * (size_t)p + 2];
    double u[3] = {0, 0, 0};
    const int beg = pt_off[p], end = pt_off[p + 1];
    for (int j = beg; j < end; ++j) {
      const int c = p_cam[j];
      if (c == fixed_cam) continue;
      const double* v = vtil + 6 * c;
      Geom g;
      obs_geom(cs + CS * c, X0, X1, X2, fx, fy, g);
      const double q0 = X1 * v[2] - X2 * v[1], q1 = X2 * v[0] - X0 * v[2], q2 = X0 * v[1] - X1 * v[0];
      double s0 = g.P[0] * q0 + g.P[1] * q1 + g.P[2] * q2 - (g.d00 * v[3] + g.d02 * v[5]);
      double s1 = g.P[3] * q0 + g.P[4] * q1 + g.P[5] * q2 - (g.d11 * v[4] + g.d12 * v[5]);
      if (ROBUST) { const double2 w = p_w[j]; s0 *= w.x; s1 *= w.y; }
      u[0] -= g.P[0] * s0 + g.P[3] * s1;
      u[1] -= g.P[1] * s0 + g.P[4] * s1;
      u[2] -= g.P[2] * s0 + g.P[5] * s1;
    }
    double hi[6], yy[3];
#pragma unroll
    for (int q = 0; q < 6; ++q) hi[q] = Hppinv[6 * (size_t)p + q];
    sym3_mul(hi, u, yy);
    if (MODE == 0) {
      y[3 * (size_t)p] = yy[0]; y[3 * (size_t)p + 1] = yy[1]; y[3 * (size_t)p + 2] = yy[2];
      acc[0] = u[0] * yy[0] + u[1] * yy[1] + u[2] * yy[2];
    } else {
      const double d0 = -(y0[3 * (size_t)p] + yy[0]);
      const double d1 = -(y0[3 * (size_t)p + 1] + yy[1]);
      const double d2 = -(y0[3 * (size_t)p + 2] + yy[2]);
      pts_trial[3 * (size_t)p] = X0 + d0;
      pts_trial[3 * (size_t)p + 1] = X1 + d1;
      pts_trial[3 * (size_t)p + 2] = X2 + d2;
      const double D0 = fmax(Hpp[6 * (size_t)p], DIAG_FLOOR), D1 = fmax(Hpp[6 * (size_t)p + 3], DIAG_FLOOR),
                   D2 = fmax(Hpp[6 * (size_t)p + 5], DIAG_FLOOR);
      acc[0] = bp[3 * (size_t)p] * d0 + bp[3 * (size_t)p + 1] * d1 + bp[3 * (size_t)p + 2] * d2;   // g^T d
      acc[1] = D0 * d0 * d0 + D1 * d1 * d1 + D2 * d2 * d2;                                           // d^T D d
      acc[2] = d0 * d0 + d1 * d1 + d2 * d2;                                                          // |d|^2
      acc[3] = X0 * X0 + X1 * X1 + X2 * X2;                                                          // |x|^2
    }
  }
  block_sum<4>(acc, sm);
  if (threadIdx.x == 0) {
    if (MODE == 0) partA[blockIdx.x] = acc[0];
    else { for (int q = 0; q < 4; ++q) partB[4 * blockIdx.x + q] = acc[q]; }
  }
}

// K4b: camera pass.  One workgroup per camera:  Wy[c] = sum_o Jc^T w (Jp y_p),  Jp y = -P y,
//   Jc^T s = [ M^T ((P^T s) x X) ; -dpi^T s ].
// The extra workgroup blockIdx.x == n_cams folds partA into comm[6 n_cams] (the u.y sum)
// so that one all-reduce of `comm` carries everything PCG needs from the shards.
// MODE 0 = PCG iteration (early exit when done), MODE 1 = right-hand side (y = y0).
template <bool ROBUST, int MODE>
__global__ void __launch_bounds__(CAM_BLOCK)
k_schur_cam(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ cam_off,
            const int* __restrict__ c_pt, const double2* __restrict__ c_w, const double* __restrict__ y,
            double fx, double fy, int n_cams, int fixed_cam, double* __restrict__ comm,
            const double* __restrict__ partA, int nblkA,
            int k, const PcgState* __restrict__ st, const double* __restrict__ partV, int nblkV, double tol2,
            int min_iters) {
  __shared__ double sm[6 * (CAM_BLOCK / 64)];
  if (MODE == 0) {
    double g, z;
    if (pcg_finished(k, st, partV, nblkV, tol2, min_iters, g, z)) return;
  }
  const int c = blockIdx.x;
  if (c == n_cams) {                       // fold the point-pass partial sums
    double a[1] = {0.0};
    if (MODE == 0) for (int b = threadIdx.x; b < nblkA; b += CAM_BLOCK) a[0] += partA[b];
    block_sum<1>(a, sm);
    if (threadIdx.x == 0) comm[6 * n_cams] = a[0];
    return;
  }
  const double* cam = cs + CS * c;
  const int beg = cam_off[c], end = cam_off[c + 1];
  double acc[6] = {0, 0, 0, 0, 0, 0};
  if (c != fixed_cam) {
    for (int i = beg + threadIdx.x; i < end; i += CAM_BLOCK) {
      const int p = c_pt[i];
      const double X0 = pts[3 * (size_t)p], X1 = pts[3 * (size_t)p + 1], X2 = pts[3 * (size_t)p + 2];
      const double y0 = y[3 * (size_t)p], y1 = y[3 * (size_t)p + 1], y2 = y[3 * (size_t)p + 2];
      Geom g;
      obs_geom(cam, X0, X1, X2, fx, fy, g);
      double s0 = -(g.P[0] * y0 + g.P[1] * y1 + g.P[2] * y2);
      double s1 = -(g.P[3] * y0 + g.P[4] * y1 + g.P[5] * y2);
      if (ROBUST) { const double2 w = c_w[i]; s0 *= w.x; s1 *= w.y; }
      const double e0 = g.P[0] * s0 + g.P[3] * s1, e1 = g.P[1] * s0 + g.P[4] * s1, e2 = g.P[2] * s0 + g.P[5] * s1;
      acc[0] += e1 * X2 - e2 * X1;
      acc[1] += e2 * X0 - e0 * X2;
      acc[2] += e0 * X1 - e1 * X0;
      acc[3] -= g.d00 * s0;
      acc[4] -= g.d11 * s1;
      acc[5] -= g.d02 * s0 + g.d12 * s1;
    }
  }
  block_sum<6>(acc, sm);
  if (threadIdx.x == 0) {
    const double* M = cam + 12;
    double* o = comm + 6 * c;
    o[0] = M[0] * acc[0] + M[3] * acc[1] + M[6] * acc[2];
    o[1] = M[1] * acc[0] + M[4] * acc[1] + M[7] * acc[2];
    o[2] = M[2] * acc[0] + M[5] * acc[1] + M[8] * acc[2];
    o[3] = acc[3]; o[4] = acc[4]; o[5] = acc[5];
  }
}

// Schur-Jacobi preconditioner blocks: E[c] (21) = sum_o W_o Hppinv_p W_o^T, W_o = Jc^T w Jp.
// One workgroup per camera; written to `out` (all-reduced across shards by the host
// side, then subtracted from Hccd and inverted in k_precond_invert).
template <bool ROBUST>
__global__ void __launch_bounds__(CAM_BLOCK)
k_schur_diag(const double* __restrict__ cs, const double* __restrict__ pts, const int* __restrict__ cam_off,
             const int* __restrict__ c_pt, const double2* __restrict__ c_w, const double* __restrict__ Hppinv,
             double fx, double fy, int fixed_cam, double* __restrict__ out) {
  __shared__ double sm[21 * (CAM_BLOCK / 64)];
  const int c = blockIdx.x;
  const double* cam = cs + CS * c;
  const int beg = cam_off[c], end = cam_off[c + 1];
  double acc[21];
#pragma unroll
  for (int q = 0; q < 21; ++q) acc[q] = 0.0;
  if (c != fixed_cam) {
    for (int i = beg + threadIdx.x; i < end; i += CAM_BLOCK) {
      const int p = c_pt[i];
      const double X0 = pts[3 * (size_t)p], X1 = pts[3 * (size_t)p + 1], X2 = pts[3 * (size_t)p + 2];
      Geom g;
      obs_geom(cam, X0, X1, X2, fx, fy, g);
      double w0 = 1.0, w1 = 1.0;
      if (ROBUST) { const double2 w = c_w[i]; w0 = w.x; w1 = w.y; }
      double hi[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) hi[q] = Hppinv[6 * (size_t)p + q];
      // G (2x2) = w Jp Hppinv Jp^T w = (w P) Hppinv (w P)^T
      double t0[3], t1[3];
      sym3_mul(hi, g.P, t0);
      sym3_mul(hi, g.P + 3, t1);
      const double G00 = w0 * w0 * (g.P[0] * t0[0] + g.P[1] * t0[1] + g.P[2] * t0[2]);
      const double G01 = w0 * w1 * (g.P[0] * t1[0] + g.P[1] * t1[1] + g.P[2] * t1[2]);
      const double G11 = w1 * w1 * (g.P[3] * t1[0] + g.P[4] * t1[1] + g.P[5] * t1[2]);
      double J0[6], J1[6];   // pre-M camera Jacobian rows
      J0[0] = g.P[1] * X2 - g.P[2] * X1; J0[1] = g.P[2] * X0 - g.P[0] * X2; J0[2] = g.P[0] * X1 - g.P[1] * X0;
      J1[0] = g.P[4] * X2 - g.P[5] * X1; J1[1] = g.P[5] * X0 - g.P[3] * X2; J1[2] = g.P[3] * X1 - g.P[4] * X0;
      J0[3] = -g.d00; J0[4] = 0.0;    J0[5] = -g.d02;
      J1[3] = 0.0;    J1[4] = -g.d11; J1[5] = -g.d12;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double l0 = J0[a] * G00 + J1[a] * G01, l1 = J0[a] * G01 + J1[a] * G11;
#pragma unroll
        for (int b = a; b < 6; ++b) acc[U6(a, b)] += l0 * J0[b] + l1 * J1[b];
      }
    }
  }
  block_sum<21>(acc, sm);
  if (threadIdx.x == 0) {
    const double* M = cam + 12;
    double A[6][6], B[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) A[i][j] = acc[S6(i, j)];
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 3; ++j) B[i][j] = A[i][0] * M[j] + A[i][1] * M[3 + j] + A[i][2] * M[6 + j];
      for (int j = 3; j < 6; ++j) B[i][j] = A[i][j];
    }
    for (int j = 0; j < 6; ++j) {
      double h0 = M[0] * B[0][j] + M[3] * B[1][j] + M[6] * B[2][j];
      double h1 = M[1] * B[0][j] + M[4] * B[1][j] + M[7] * B[2][j];
      double h2 = M[2] * B[0][j] + M[5] * B[1][j] + M[8] * B[2][j];
      A[0][j] = h0; A[1][j] = h1; A[2][j] = h2;
      A[3][j] = B[3][j]; A[4][j] = B[4][j]; A[5][j] = B[5][j];
    }
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) out[21 * c + U6(i, j)] = A[i][j];
  }
}

// Minv[c] = (Hccd[c] - E[c])^-1 (E may be null: plain block-Jacobi).
__global__ void k_precond_invert(const double* __restrict__ Hccd, const double* __restrict__ E, int n_cams,
                                 double* __restrict__ Minv) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  double h[21], inv[21];
  for (int q = 0; q < 21; ++q) h[q] = Hccd[21 * c + q] - (E ? E[21 * c + q] : 0.0);
  spd6_inverse(h, inv);
  for (int q = 0; q < 21; ++q) Minv[21 * c + q] = inv[q];
}

// PCG start: g = -(bc - Wy0) (fixed camera: 0), x = 0, r = g, z = Minv r, p = s = 0,
// vtil = (M z_r, z_t), first partial sums gamma = r.z, zeta = z.Hccd z, state reset.
__global__ void __launch_bounds__(VEC_BLOCK)
k_pcg_init(const double* __restrict__ bc, const double* __restrict__ comm, const double* __restrict__ Hccd,
           const double* __restrict__ Minv, const double* __restrict__ cs, int n_cams, int fixed_cam,
           double* __restrict__ gvec, double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
           double* __restrict__ s, double* __restrict__ z, double* __restrict__ vtil,
           double* __restrict__ partV, PcgState* __restrict__ st) {
  __shared__ double sm[2];
  const int c = blockIdx.x * VEC_BLOCK + threadIdx.x;
  double acc[2] = {0, 0};
  if (c < n_cams) {
    double g[6], zz[6], hz[6];
    for (int q = 0; q < 6; ++q) g[q] = (c == fixed_cam) ? 0.0 : -(bc[6 * c + q] - comm[6 * c + q]);
    sym6_mul(Minv + 21 * c, g, zz);
    sym6_mul(Hccd + 21 * c, zz, hz);
    const double* M = cs + CS * c + 12;
    for (int q = 0; q < 6; ++q) {
      gvec[6 * c + q] = g[q]; r[6 * c + q] = g[q]; x[6 * c + q] = 0.0; p[6 * c + q] = 0.0; s[6 * c + q] = 0.0;
      z[6 * c + q] = zz[q];
      acc[0] += g[q] * zz[q];
      acc[1] += zz[q] * hz[q];
    }
    vtil[6 * c + 0] = M[0] * zz[0] + M[1] * zz[1] + M[2] * zz[2];
    vtil[6 * c + 1] = M[3] * zz[0] + M[4] * zz[1] + M[5] * zz[2];
    vtil[6 * c + 2] = M[6] * zz[0] + M[7] * zz[1] + M[8] * zz[2];
    vtil[6 * c + 3] = zz[3]; vtil[6 * c + 4] = zz[4]; vtil[6 * c + 5] = zz[5];
  }
  block_sum<2>(acc, sm);
  if (threadIdx.x == 0) {
    partV[2 * blockIdx.x] = acc[0];
    partV[2 * blockIdx.x + 1] = acc[1];
    if (blockIdx.x == 0) {
      PcgState s0 = {0.0, 0.0, 0.0, 0.0, 0, 0, 0, 0};
      st[0] = s0;
      st[1] = s0;
    }
  }
}

// K5: one PCG iteration's vector work (Chronopoulos-Gear single-reduction CG), one
// thread per camera.  With z the preconditioned residual and w = S z:
//   gamma = r.z (partials from the previous step), delta = z.Hccd z - u.y,
//   beta = gamma/gamma_prev, alpha = gamma / (delta - beta gamma / alpha_prev),
//   p = z + beta p, s = w + beta s, x += alpha p, r -= alpha s, z = Minv r.
// Every workgroup recomputes the scalars from the same words; workgroup 0 publishes
// the next state into the other parity slot.
__global__ void __launch_bounds__(VEC_BLOCK)
k_pcg_step(int k, const double* __restrict__ comm, const double* __restrict__ Hccd, const double* __restrict__ Minv,
           const double* __restrict__ cs, int n_cams, int fixed_cam, double tol2, int min_iters,
           double* __restrict__ x, double* __restrict__ r, double* __restrict__ p, double* __restrict__ s,
           double* __restrict__ z, double* __restrict__ vtil, double* __restrict__ partV, int nblkV,
           PcgState* __restrict__ st) {
  __shared__ double sm[2];
  double gamma, zeta;
  const bool fin = pcg_finished(k, st, partV, nblkV, tol2, min_iters, gamma, zeta);
  const PcgState sin = st[k & 1];
  PcgState* sout = st + ((k + 1) & 1);
  if (fin) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      PcgState o = sin;
      if (!o.done) { o.done = 1; o.iters = k; if (k == 0) o.gamma0 = gamma; }
      *sout = o;
    }
    return;
  }
  const double uy = comm[6 * n_cams];
  const double delta = zeta - uy;
  const double beta = (k == 0) ? 0.0 : gamma / sin.gamma_prev;
  const double denom = (k == 0) ? delta : delta - beta * gamma / sin.alpha_prev;
  if (!(denom > 0.0) || !isfinite(denom)) {           // breakdown: keep x, stop
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      PcgState o = sin;
      o.done = 2; o.iters = k; if (k == 0) o.gamma0 = gamma;
      *sout = o;
    }
    return;
  }
  const double alpha = gamma / denom;
  const int c = blockIdx.x * VEC_BLOCK + threadIdx.x;
  double acc[2] = {0, 0};
  if (c < n_cams && c != fixed_cam) {
    double zz[6], w[6], pp[6], ss[6], rr[6], hz[6];
    for (int q = 0; q < 6; ++q) zz[q] = z[6 * c + q];
    sym6_mul(Hccd + 21 * c, zz, w);
    for (int q = 0; q < 6; ++q) {
      w[q] -= comm[6 * c + q];
      pp[q] = zz[q] + beta * p[6 * c + q];
      ss[q] = w[q] + beta * s[6 * c + q];
      x[6 * c + q] += alpha * pp[q];
      rr[q] = r[6 * c + q] - alpha * ss[q];
      p[6 * c + q] = pp[q]; s[6 * c + q] = ss[q]; r[6 * c + q] = rr[q];
    }
    sym6_mul(Minv + 21 * c, rr, zz);
    sym6_mul(Hccd + 21 * c, zz, hz);
    const double* M = cs + CS * c + 12;
    for (int q = 0; q < 6; ++q) {
      z[6 * c + q] = zz[q];
      acc[0] += rr[q] * zz[q];
      acc[1] += zz[q] * hz[q];
    }
    vtil[6 * c + 0] = M[0] * zz[0] + M[1] * zz[1] + M[2] * zz[2];
    vtil[6 * c + 1] = M[3] * zz[0] + M[4] * zz[1] + M[5] * zz[2];
    vtil[6 * c + 2] = M[6] * zz[0] + M[7] * zz[1] + M[8] * zz[2];
    vtil[6 * c + 3] = zz[3]; vtil[6 * c + 4] = zz[4]; vtil[6 * c + 5] = zz[5];
  }
  block_sum<2>(acc, sm);
  if (threadIdx.x == 0) {
    double* pv = partV + (size_t)((k + 1) & 1) * 2 * nblkV;
    pv[2 * blockIdx.x] = acc[0];
    pv[2 * blockIdx.x + 1] = acc[1];
    if (blockIdx.x == 0) {
      PcgState o = sin;
      o.gamma_prev = gamma; o.alpha_prev = alpha;
      o.gamma0 = (k == 0) ? gamma : sin.gamma0;
      o.iters = k + 1;
      *sout = o;
    }
  }
}

// K7a: camera update.  cams_trial = cams + dc, vtil = (M dc_r, dc_t) for the back
// substitution, and the camera-side scalars -> partC[block][5]:
//   bc.dc, sum Dc dc^2, dc.r_pcg, |dc|^2, |cams|^2
__global__ void __launch_bounds__(VEC_BLOCK)
k_cam_update(const double* __restrict__ cams, const double* __restrict__ dc, const double* __restrict__ rpcg,
             const double* __restrict__ Hcc, const double* __restrict__ bc, const double* __restrict__ cs,
             int n_cams, int fixed_cam, double* __restrict__ cams_trial, double* __restrict__ vtil,
             double* __restrict__ partC) {
  __shared__ double sm[5];
  const int c = blockIdx.x * VEC_BLOCK + threadIdx.x;
  double acc[5] = {0, 0, 0, 0, 0};
  if (c < n_cams) {
    double d[6];
    for (int q = 0; q < 6; ++q) d[q] = (c == fixed_cam) ? 0.0 : dc[6 * c + q];
    for (int q = 0; q < 6; ++q) {
      const double xq = cams[6 * c + q];
      cams_trial[6 * c + q] = xq + d[q];
      acc[0] += bc[6 * c + q] * d[q];
      acc[1] += fmax(Hcc[21 * c + U6(q, q)], DIAG_FLOOR) * d[q] * d[q];
      acc[2] += d[q] * ((c == fixed_cam) ? 0.0 : rpcg[6 * c + q]);
      acc[3] += d[q] * d[q];
      acc[4] += xq * xq;
    }
    const double* M = cs + CS * c + 12;
    vtil[6 * c + 0] = M[0] * d[0] + M[1] * d[1] + M[2] * d[2];
    vtil[6 * c + 1] = M[3] * d[0] + M[4] * d[1] + M[5] * d[2];
    vtil[6 * c + 2] = M[6] * d[0] + M[7] * d[1] + M[8] * d[2];
    vtil[6 * c + 3] = d[3]; vtil[6 * c + 4] = d[4]; vtil[6 * c + 5] = d[5];
  }
  block_sum<5>(acc, sm);
  if (threadIdx.x == 0) for (int q = 0; q < 5; ++q) partC[5 * blockIdx.x + q] = acc[q];
}

// out[j] = sum_i part[i * ncols + j]   (single workgroup, fixed order)
__global__ void __launch_bounds__(256)
k_reduce_cols(const double* __restrict__ part, int nrows, int ncols, double* __restrict__ out) {
  __shared__ double sm[4];
  for (int j = 0; j < ncols; ++j) {
    double a[1] = {0.0};
    for (int i = threadIdx.x; i < nrows; i += 256) a[0] += part[(size_t)i * ncols + j];
    block_sum<1>(a, sm);
    if (threadIdx.x == 0) out[j] = a[0];
  }
}

// out[0] = max |v[i]|  (single workgroup)
__global__ void __launch_bounds__(256)
k_absmax(const double* __restrict__ v, size_t n, double* __restrict__ out) {
  __shared__ double sm[4];
  double m = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 256) m = fmax(m, fabs(v[i]));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

// Not-converged PCG state for the test / bench hooks that run one pass in isolation.
__global__ void k_pcg_reset(PcgState* __restrict__ st, double* __restrict__ partV, int nblkV) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    PcgState s0 = {1.0, 1.0, 1.0, 0.0, 0, 0, 0, 0};
    st[0] = s0; st[1] = s0;
    for (int b = 0; b < 4 * nblkV; ++b) partV[b] = (b & 1) ? 0.0 : 1.0;
  }
}

// out = (Hcc + lam Dc) v - Wy   (test hook behind ba_schur_apply; fixed row = identity)
__global__ void k_schur_combine(const double* __restrict__ Hccd, const double* __restrict__ v,
                                const double* __restrict__ comm, int n_cams, int fixed_cam, double* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  double vv[6], w[6];
  for (int q = 0; q < 6; ++q) vv[q] = v[6 * c + q];
  sym6_mul(Hccd + 21 * c, vv, w);
  for (int q = 0; q < 6; ++q) out[6 * c + q] = (c == fixed_cam) ? vv[q] : w[q] - comm[6 * c + q];
}

// vtil = (M v_r, v_t) for an arbitrary camera vector (test hook / back substitution)
__global__ void k_vtil(const double* __restrict__ v, const double* __restrict__ cs, int n_cams, int fixed_cam,
                       double* __restrict__ vtil) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cams) return;
  const double* M = cs + CS * c + 12;
  double d[6];
  for (int q = 0; q < 6; ++q) d[q] = (c == fixed_cam) ? 0.0 : v[6 * c + q];
  vtil[6 * c + 0] = M[0] * d[0] + M[1] * d[1] + M[2] * d[2];
  vtil[6 * c + 1] = M[3] * d[0] + M[4] * d[1] + M[5] * d[2];
  vtil[6 * c + 2] = M[6] * d[0] + M[7] * d[1] + M[8] * d[2];
  vtil[6 * c + 3] = d[3]; vtil[6 * c + 4] = d[4]; vtil[6 * c + 5] = d[5];
}

}  // namespace ba
