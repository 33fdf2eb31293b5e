// Camera models of the solve step.  Every kernel of the LM / Schur / PCG loop (ba_kernels.hpp) is written once and
// instantiated per model; a model supplies the block sizes, the layout of a camera's row in the point passes' table and
// the per-observation arithmetic (projection, residual, the factors of the analytic Jacobian blocks):
//
//   Pinhole   the reference's camera: cv2.projectPoints(X, rvec, tvec, K, None) with one shared K
//             (src/bundle_adjuster.py:67); 6 parameters per camera [rvec | t], 2x6 / 2x3 blocks
//   BalCam    the BAL camera [rvec | t | f k1 k2] (BASELINE config 5 is stated on a BAL problem; the reference has
//             no counterpart): P = R X + t, p = -P.xy / P.z, proj = f (1 + k1 |p|^2 + k2 |p|^4) p; 9 parameters per
//             camera, 2x9 / 2x3 blocks, (f, k1, k2) adjusted per camera
//
// Common factorisation (additive rotation-vector update through the SO(3) right Jacobian M, ba_device.hpp):
//     Jp = -Pm                      Pm = (d proj / d Xc) R                        (2x3)
//     Jc = [ (Pm_row x X) M | -D_row | model-specific intrinsics columns ]        D = d proj / d Xc (2x3)
// Camera passes accumulate with the PRE-M rows (Pm_row x X, ...) and the consumer applies M once per camera
// (congruence diag(M, I) for blocks, M^T for vectors); a camera vector reaches the point passes as vt = (M v_r, v_t, ..).
#pragma once
#include "ba_device.hpp"

namespace ba {

// ------------------------------------------------------------------------------------------------ pinhole
struct Pinhole {
  static constexpr int ID = 0;
  static constexpr int NB = 6;          // parameters per camera
  static constexpr int NH = 21;         // packed upper triangle of a camera block
  static constexpr int NL = NH + NB;    // running sums of the camera half of a linearisation
  static constexpr int CAM = 12;        // per-camera doubles a camera pass holds: R[9] t[3]
  static constexpr int TA = 18;         // row stride of the point passes' camera table: R[9] t[3] | vt[6]
  static constexpr int LIN_ROW = 12;    // leading doubles of a table row the linearisation reads
  static constexpr int SCH_ROW = 18;    // ... the Schur point passes read
  static constexpr int VOFF = 12;       // offset of vt in a table row
  static constexpr int VC = 16;         // cameras per wave in the camera-vector kernels (k_pcg_setup / k_pcg_step / k_cam_update)

  template <typename T>
  struct Obs {                           // what one observation contributes, in T (double, or float in the fp32-Jacobian PCG passes)
    GeomT<T> g;
  };

  // camera pass: scalar loads of the wave-uniform camera (k_cam_schur) / vector loads (row-form kernels)
  template <typename T>
  __device__ static __forceinline__ void load_cam(const double* __restrict__ cs, const double* __restrict__, int c, T (&cam)[CAM]) {
    const double* camd = cs + CS * (size_t)c;
#pragma unroll
    for (int q = 0; q < CAM; ++q) cam[q] = (T)camd[q];
  }
  __device__ static __forceinline__ void load_cam_vec(const double* __restrict__ cs, const double* __restrict__, int c, double (&cam)[CAM]) {
    const double2* cp = (const double2*)(cs + CS * (size_t)c);
#pragma unroll
    for (int q = 0; q < CAM / 2; ++q) { const double2 t = cp[q]; cam[2 * q] = t.x; cam[2 * q + 1] = t.y; }
  }
  // table row of camera c from its state (k_cam_prepare, k_cam_update): the geometry part
  __device__ static __forceinline__ void table_row(const double* __restrict__ cs_c, const double* __restrict__, double* __restrict__ row) {
#pragma unroll
    for (int q = 0; q < 12; ++q) row[q] = cs_c[q];
  }

  // geometry with the exact division (residual, linearisation) / with v_rcp + Newton (Schur-operator passes)
  template <bool FAST, typename T, typename CamT>
  __device__ static __forceinline__ void geom(const CamT* __restrict__ cam, T X0, T X1, T X2, T fx, T fy, Obs<T>& o) {
    obs_geom_t<FAST, T, CamT>(cam, X0, X1, X2, fx, fy, o.g);
  }
  template <typename T>
  __device__ static __forceinline__ void residual(const Obs<T>& o, T u, T v, T fx, T fy, T cx, T cy, T& ru, T& rv) {
    ru = u - (o.g.xh * fx + cx);
    rv = v - (o.g.yh * fy + cy);
  }
  template <typename T>
  __device__ static __forceinline__ const T* pm(const Obs<T>& o) { return o.g.P; }       // Pm = -Jp, row-major 2x3
  // pre-M rows of Jc: J0 = [P0 x X | -d00 0 -d02], J1 = [P1 x X | 0 -d11 -d12]
  __device__ static __forceinline__ void jac_rows(const Obs<double>& o, double X0, double X1, double X2, double (&J0)[NB], double (&J1)[NB]) {
    const Geom& g = o.g;
    J0[0] = g.P[1] * X2 - g.P[2] * X1; J0[1] = g.P[2] * X0 - g.P[0] * X2; J0[2] = g.P[0] * X1 - g.P[1] * X0;
    J1[0] = g.P[4] * X2 - g.P[5] * X1; J1[1] = g.P[5] * X0 - g.P[3] * X2; J1[2] = g.P[3] * X1 - g.P[4] * X0;
    J0[3] = -g.d00; J0[4] = 0.0;    J0[5] = -g.d02;
    J1[3] = 0.0;    J1[4] = -g.d11; J1[5] = -g.d12;
  }
  // Jc vt (two scalars, before the weights), vt = (M v_r, v_t):  P (X x vt_r) - dpi vt_t
  template <typename T>
  __device__ static __forceinline__ void jc_times(const Obs<T>& o, T X0, T X1, T X2, const T* __restrict__ v, T& s0, T& s1) {
    const GeomT<T>& g = o.g;
    const T q0 = X1 * v[2] - X2 * v[1], q1 = X2 * v[0] - X0 * v[2], q2 = X0 * v[1] - X1 * v[0];
    s0 = g.P[0] * q0 + g.P[1] * q1 + g.P[2] * q2 - (g.d00 * v[3] + g.d02 * v[5]);
    s1 = g.P[3] * q0 + g.P[4] * q1 + g.P[5] * q2 - (g.d11 * v[4] + g.d12 * v[5]);
  }
  // acc += Jc_preM^T s  (s = the two weighted scalars), sums in fp64
  template <typename T>
  __device__ static __forceinline__ void jct_accumulate(const Obs<T>& o, T X0, T X1, T X2, T s0, T s1, double (&acc)[NB]) {
    const GeomT<T>& g = o.g;
    const T e0 = g.P[0] * s0 + g.P[3] * s1, e1 = g.P[1] * s0 + g.P[4] * s1, e2 = g.P[2] * s0 + g.P[5] * s1;
    acc[0] += (double)(e1 * X2 - e2 * X1);
    acc[1] += (double)(e2 * X0 - e0 * X2);
    acc[2] += (double)(e0 * X1 - e1 * X0);
    acc[3] -= (double)(g.d00 * s0);
    acc[4] -= (double)(g.d11 * s1);
    acc[5] -= (double)(g.d02 * s0 + g.d12 * s1);
  }
};

// ---------------------------------------------------------------------------------------------------- BAL
struct BalCam {
  static constexpr int ID = 1;
  static constexpr int NB = 9;
  static constexpr int NH = 45;
  static constexpr int NL = NH + NB;
  static constexpr int CAM = 15;        // R[9] t[3] f k1 k2
  // Table row R[9] t[3] f k1 k2 | vt[9] | pad[2]: 26 doubles = 52 dwords.  The stride is chosen for the LDS banks:
  // 52 c mod 64 runs through all sixteen multiples of 4, so the sixteen rows a ds_read_b128 group touches fall into
  // distinct bank classes as often as the pinhole's 36-dword rows do (a 48-dword row has four classes only).
  static constexpr int TA = 26;
  static constexpr int LIN_ROW = 16;    // (15 used; loads are 16 bytes wide)
  static constexpr int SCH_ROW = 24;
  static constexpr int VOFF = 15;
#ifndef BA_VEC_CAMS_BAL
#define BA_VEC_CAMS_BAL 8
#endif
  static constexpr int VC = BA_VEC_CAMS_BAL;   // (a 9-parameter camera's slices are 2.2x the pinhole's: half the cameras per wave keep the loads in flight per lane the same)

  template <typename T>
  struct Obs {
    T p0, p1, n2, rad, f;
    T A[6];                              // d proj / d P, rows (u, v): full 2x3 (the radial term couples x and y)
    T B[6];                              // A R  (= -Jp)
  };

  template <typename T>
  __device__ static __forceinline__ void load_cam(const double* __restrict__ cs, const double* __restrict__ intr, int c, T (&cam)[CAM]) {
    const double* camd = cs + CS * (size_t)c;
#pragma unroll
    for (int q = 0; q < 12; ++q) cam[q] = (T)camd[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) cam[12 + q] = (T)intr[3 * (size_t)c + q];
  }
  __device__ static __forceinline__ void load_cam_vec(const double* __restrict__ cs, const double* __restrict__ intr, int c, double (&cam)[CAM]) {
    const double2* cp = (const double2*)(cs + CS * (size_t)c);
#pragma unroll
    for (int q = 0; q < 6; ++q) { const double2 t = cp[q]; cam[2 * q] = t.x; cam[2 * q + 1] = t.y; }
#pragma unroll
    for (int q = 0; q < 3; ++q) cam[12 + q] = intr[3 * (size_t)c + q];
  }
  __device__ static __forceinline__ void table_row(const double* __restrict__ cs_c, const double* __restrict__ intr_c, double* __restrict__ row) {
#pragma unroll
    for (int q = 0; q < 12; ++q) row[q] = cs_c[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) row[12 + q] = intr_c[q];
  }

  template <bool FAST, typename T, typename CamT>
  __device__ static __forceinline__ void geom(const CamT* __restrict__ cam, T X0, T X1, T X2, T, T, Obs<T>& g) {
    const T Px = cam[0] * X0 + cam[1] * X1 + cam[2] * X2 + cam[9];
    const T Py = cam[3] * X0 + cam[4] * X1 + cam[5] * X2 + cam[10];
    const T Pz = cam[6] * X0 + cam[7] * X1 + cam[8] * X2 + cam[11];
    const T iz = (Pz != T(0)) ? (FAST ? recip_fast(Pz) : T(1) / Pz) : T(1);            // guarded like the pinhole (obs_geom_t)
    const T f = cam[12], k1 = cam[13], k2 = cam[14];
    const T p0 = -Px * iz, p1 = -Py * iz;
    const T n2 = p0 * p0 + p1 * p1;
    const T rad = T(1) + n2 * (k1 + k2 * n2), drad = k1 + T(2) * k2 * n2;
    g.p0 = p0; g.p1 = p1; g.n2 = n2; g.rad = rad; g.f = f;
    const T d00 = f * (rad + T(2) * drad * p0 * p0), d01 = f * T(2) * drad * p0 * p1, d11 = f * (rad + T(2) * drad * p1 * p1);
    // d p / d P = -iz [1 0 p0; 0 1 p1]
    g.A[0] = -iz * d00; g.A[1] = -iz * d01; g.A[2] = -iz * (d00 * p0 + d01 * p1);
    g.A[3] = -iz * d01; g.A[4] = -iz * d11; g.A[5] = -iz * (d01 * p0 + d11 * p1);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) g.B[3 * r + k] = g.A[3 * r] * cam[k] + g.A[3 * r + 1] * cam[3 + k] + g.A[3 * r + 2] * cam[6 + k];
  }
  template <typename T>
  __device__ static __forceinline__ void residual(const Obs<T>& g, T u, T v, T, T, T, T, T& ru, T& rv) {
    ru = u - g.f * g.rad * g.p0;
    rv = v - g.f * g.rad * g.p1;
  }
  template <typename T>
  __device__ static __forceinline__ const T* pm(const Obs<T>& g) { return g.B; }
  // pre-M rows: [B_row x X | -A_row | -rad p_row | -f n2 p_row | -f n2^2 p_row]
  __device__ static __forceinline__ void jac_rows(const Obs<double>& g, double X0, double X1, double X2, double (&J0)[NB], double (&J1)[NB]) {
    J0[0] = g.B[1] * X2 - g.B[2] * X1; J0[1] = g.B[2] * X0 - g.B[0] * X2; J0[2] = g.B[0] * X1 - g.B[1] * X0;
    J1[0] = g.B[4] * X2 - g.B[5] * X1; J1[1] = g.B[5] * X0 - g.B[3] * X2; J1[2] = g.B[3] * X1 - g.B[4] * X0;
#pragma unroll
    for (int q = 0; q < 3; ++q) { J0[3 + q] = -g.A[q]; J1[3 + q] = -g.A[3 + q]; }
    const double fn = g.f * g.n2, fnn = fn * g.n2;
    J0[6] = -g.rad * g.p0; J1[6] = -g.rad * g.p1;
    J0[7] = -fn * g.p0;    J1[7] = -fn * g.p1;
    J0[8] = -fnn * g.p0;   J1[8] = -fnn * g.p1;
  }
  template <typename T>
  __device__ static __forceinline__ void jc_times(const Obs<T>& g, T X0, T X1, T X2, const T* __restrict__ v, T& s0, T& s1) {
    const T q0 = X1 * v[2] - X2 * v[1], q1 = X2 * v[0] - X0 * v[2], q2 = X0 * v[1] - X1 * v[0];
    const T ki = g.rad * v[6] + g.f * g.n2 * (v[7] + g.n2 * v[8]);
    s0 = g.B[0] * q0 + g.B[1] * q1 + g.B[2] * q2 - (g.A[0] * v[3] + g.A[1] * v[4] + g.A[2] * v[5]) - ki * g.p0;
    s1 = g.B[3] * q0 + g.B[4] * q1 + g.B[5] * q2 - (g.A[3] * v[3] + g.A[4] * v[4] + g.A[5] * v[5]) - ki * g.p1;
  }
  template <typename T>
  __device__ static __forceinline__ void jct_accumulate(const Obs<T>& g, T X0, T X1, T X2, T s0, T s1, double (&acc)[NB]) {
    const T e0 = g.B[0] * s0 + g.B[3] * s1, e1 = g.B[1] * s0 + g.B[4] * s1, e2 = g.B[2] * s0 + g.B[5] * s1;
    acc[0] += (double)(e1 * X2 - e2 * X1);
    acc[1] += (double)(e2 * X0 - e0 * X2);
    acc[2] += (double)(e0 * X1 - e1 * X0);
    acc[3] -= (double)(g.A[0] * s0 + g.A[3] * s1);
    acc[4] -= (double)(g.A[1] * s0 + g.A[4] * s1);
    acc[5] -= (double)(g.A[2] * s0 + g.A[5] * s1);
    const T ps = g.p0 * s0 + g.p1 * s1, fn = g.f * g.n2;
    acc[6] -= (double)(g.rad * ps);
    acc[7] -= (double)(fn * ps);
    acc[8] -= (double)(fn * g.n2 * ps);
  }
};

}  // namespace ba
