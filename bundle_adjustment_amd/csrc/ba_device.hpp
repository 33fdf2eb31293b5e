// Device-side building blocks shared by every kernel: camera state, per-observation
// geometry (projection + the factors of the analytic Jacobian blocks), small dense
// algebra on packed symmetric blocks.  (Reductions: ba_dpp.hpp.)
//
// Jacobian factorisation used everywhere (residual = observed - pi(R X + t), additive
// rotation-vector update as scipy applies it, scipy/optimize/_lsq/trf.py:497-498):
//     dpi = d pi / d Xc                      (2x3, two structural zeros)
//     P   = dpi R                            (2x3)
//     Jp  = -P                               d res / d X
//     Jt  = -dpi                             d res / d t
//     Jr  =  P [X]x M,  M = J_r(rvec)        d res / d rvec  (SO(3) right Jacobian)
// so no per-observation matrix is ever stored: R, t, M come from the per-camera state and
// X from the point, and every pass over the observation list recomputes P and dpi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ba {

constexpr int CS = 24;   // doubles of state per camera: R[9] t[3] M[9] pad[3]
constexpr double DBL_EPS = 2.220446049250313e-16;
constexpr double DIAG_FLOOR = 1e-12;

// index of (i,j), i<=j, in the packed upper triangle of a symmetric NB x NB block (row-major: 00 01 .. 0(NB-1) 11 ..).
// Camera blocks are 6x6 for the reference's pinhole (rvec | t) and 9x9 for the BAL camera (rvec | t | f k1 k2).
__host__ __device__ constexpr int UT(int nb, int i, int j) { return i * nb - (i * (i - 1)) / 2 + (j - i); }
__host__ __device__ constexpr int ST(int nb, int i, int j) { return i <= j ? UT(nb, i, j) : UT(nb, j, i); }
__host__ __device__ constexpr int U6(int i, int j) { return UT(6, i, j); }
__host__ __device__ constexpr int S6(int i, int j) { return ST(6, i, j); }
__host__ __device__ constexpr int U3(int i, int j) { return i * 3 - (i * (i - 1)) / 2 + (j - i); }
__host__ __device__ constexpr int S3(int i, int j) { return i <= j ? U3(i, j) : U3(j, i); }

// ---- camera state -----------------------------------------------------------------
// R follows cv2.Rodrigues(vector): identity below DBL_EPSILON, else
// cos I + (1-cos) k k^T + sin [k]x.  M = I - b [r]x + d [r]x^2 with series below 0.05 rad.
__device__ inline void camera_state(const double* __restrict__ cam, double* __restrict__ cs) {
  const double rx = cam[0], ry = cam[1], rz = cam[2];
  const double t2 = rx * rx + ry * ry + rz * rz;
  const double th = sqrt(t2);
  double R[9];
  if (th < DBL_EPS) {
    R[0] = 1; R[1] = 0; R[2] = 0; R[3] = 0; R[4] = 1; R[5] = 0; R[6] = 0; R[7] = 0; R[8] = 1;
  } else {
    const double c = cos(th), s = sin(th), c1 = 1.0 - c;
    const double kx = rx / th, ky = ry / th, kz = rz / th;
    R[0] = c + c1 * kx * kx;      R[1] = c1 * kx * ky - s * kz; R[2] = c1 * kx * kz + s * ky;
    R[3] = c1 * kx * ky + s * kz; R[4] = c + c1 * ky * ky;      R[5] = c1 * ky * kz - s * kx;
    R[6] = c1 * kx * kz - s * ky; R[7] = c1 * ky * kz + s * kx; R[8] = c + c1 * kz * kz;
  }
  double b, d;
  if (th < 0.05) {
    b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0 - t2 * t2 * t2 / 40320.0;
    d = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0 - t2 * t2 * t2 / 362880.0;
  } else {
    b = (1.0 - cos(th)) / t2;
    d = (th - sin(th)) / (t2 * th);
  }
  // [r]x^2 = r r^T - t2 I
  double M[9];
  M[0] = 1.0 + d * (rx * rx - t2); M[1] = b * rz + d * rx * ry;     M[2] = -b * ry + d * rx * rz;
  M[3] = -b * rz + d * rx * ry;    M[4] = 1.0 + d * (ry * ry - t2); M[5] = b * rx + d * ry * rz;
  M[6] = b * ry + d * rx * rz;     M[7] = -b * rx + d * ry * rz;    M[8] = 1.0 + d * (rz * rz - t2);
#pragma unroll
  for (int i = 0; i < 9; ++i) cs[i] = R[i];
  cs[9] = cam[3]; cs[10] = cam[4]; cs[11] = cam[5];
#pragma unroll
  for (int i = 0; i < 9; ++i) cs[12 + i] = M[i];
  cs[21] = 0; cs[22] = 0; cs[23] = 0;
}

// ---- per-observation geometry ------------------------------------------------------
// T = double everywhere except the PCG passes of jacobian_precision = 1 (config 5: Jacobian
// blocks in fp32, every accumulation in fp64).
template <typename T>
struct GeomT {
  T xh, yh;                 // Xc.x / Xc.z, Xc.y / Xc.z
  T d00, d02, d11, d12;     // dpi = [d00 0 d02; 0 d11 d12]
  T P[6];                   // dpi R, row-major 2x3
};
using Geom = GeomT<double>;

// 1 / z for the Schur-operator passes (PCG, right-hand side, back substitution): v_rcp + Newton steps, 5
// instructions against the 14 of the IEEE division sequence; the last bit may differ from 1.0 / z.  The
// residual and the linearisation (what parity with the reference is measured on) keep the exact division.
__device__ inline double recip_fast(double z) {
  double r = __builtin_amdgcn_rcp(z);
  double e = fma(-z, r, 1.0);
  r = fma(r, e, r);
  e = fma(-z, r, 1.0);
  return fma(r, e, r);
}
__device__ inline float recip_fast(float z) {
  float r = __builtin_amdgcn_rcpf(z);
  return fmaf(r, fmaf(-z, r, 1.0f), r);
}

template <bool FAST, typename T, typename CamT>
__device__ inline void obs_geom_t(const CamT* __restrict__ cs, const T X0, const T X1, const T X2,
                                  const T fx, const T fy, GeomT<T>& g) {
  const T Xc0 = cs[0] * X0 + cs[1] * X1 + cs[2] * X2 + cs[9];
  const T Xc1 = cs[3] * X0 + cs[4] * X1 + cs[5] * X2 + cs[10];
  const T Xc2 = cs[6] * X0 + cs[7] * X1 + cs[8] * X2 + cs[11];
  const T iz = (Xc2 != T(0)) ? (FAST ? recip_fast(Xc2) : T(1) / Xc2) : T(1);   // cv2.projectPoints guards z == 0 as 1
  g.xh = Xc0 * iz;
  g.yh = Xc1 * iz;
  g.d00 = fx * iz;
  g.d02 = -fx * g.xh * iz;
  g.d11 = fy * iz;
  g.d12 = -fy * g.yh * iz;
  g.P[0] = g.d00 * cs[0] + g.d02 * cs[6];
  g.P[1] = g.d00 * cs[1] + g.d02 * cs[7];
  g.P[2] = g.d00 * cs[2] + g.d02 * cs[8];
  g.P[3] = g.d11 * cs[3] + g.d12 * cs[6];
  g.P[4] = g.d11 * cs[4] + g.d12 * cs[7];
  g.P[5] = g.d11 * cs[5] + g.d12 * cs[8];
}
template <typename T, typename CamT>
__device__ inline void obs_geom(const CamT* __restrict__ cs, const T X0, const T X1, const T X2,
                                const T fx, const T fy, GeomT<T>& g) {
  obs_geom_t<false, T, CamT>(cs, X0, X1, X2, fx, fy, g);
}
template <typename T, typename CamT>
__device__ inline void obs_geom_fast(const CamT* __restrict__ cs, const T X0, const T X1, const T X2,
                                     const T fx, const T fy, GeomT<T>& g) {
  obs_geom_t<true, T, CamT>(cs, X0, X1, X2, fx, fy, g);
}

// projection only (trial-point cost and the residual entry point)
template <typename CamT>
__device__ inline void obs_project(const CamT* __restrict__ cs, const double X0, const double X1, const double X2,
                                   double& xh, double& yh) {
  const double Xc0 = cs[0] * X0 + cs[1] * X1 + cs[2] * X2 + cs[9];
  const double Xc1 = cs[3] * X0 + cs[4] * X1 + cs[5] * X2 + cs[10];
  const double Xc2 = cs[6] * X0 + cs[7] * X1 + cs[8] * X2 + cs[11];
  const double iz = (Xc2 != 0.0) ? 1.0 / Xc2 : 1.0;
  xh = Xc0 * iz;
  yh = Xc1 * iz;
}

// Huber with threshold C (scipy least_squares.py:169-178 scaled by f_scale):
// rho-term f_scale^2 rho((f/C)^2) and IRLS weight rho'.
__device__ inline void huber(const double f, const double C, double& term, double& w) {
  const double a = fabs(f);
  if (a <= C) { term = f * f; w = 1.0; }
  else        { term = 2.0 * C * a - C * C; w = C / a; }
}

// ---- small dense algebra -----------------------------------------------------------
__device__ inline void sym3_inverse(const double* __restrict__ h, double* __restrict__ inv) {
  // h, inv packed upper: 00 01 02 11 12 22
  const double a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5];
  const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  inv[0] = c00 * id; inv[1] = c01 * id; inv[2] = c02 * id;
  inv[3] = (a * f - c * c) * id; inv[4] = (b * c - a * e) * id; inv[5] = (a * d - b * b) * id;
}

__device__ inline void sym3_mul(const double* __restrict__ h, const double* __restrict__ v, double* __restrict__ o) {
  o[0] = h[0] * v[0] + h[1] * v[1] + h[2] * v[2];
  o[1] = h[1] * v[0] + h[3] * v[1] + h[4] * v[2];
  o[2] = h[2] * v[0] + h[4] * v[1] + h[5] * v[2];
}

template <int NB>
__device__ inline void symN_mul(const double* __restrict__ h, const double* __restrict__ v, double* __restrict__ o) {
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    double s = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) s += h[ST(NB, i, j)] * v[j];
    o[i] = s;
  }
}
__device__ inline void sym6_mul(const double* __restrict__ h, const double* __restrict__ v, double* __restrict__ o) { symN_mul<6>(h, v, o); }

// inverse of a symmetric positive definite NB x NB block (packed upper in, packed upper out) by
// Cholesky; a non-positive pivot is replaced by DIAG_FLOOR so the result stays finite.  Fully unrolled: every
// index is a compile-time constant, the structural zeros of L and L^-1 never exist.
template <int NB>
__device__ inline void spdN_inverse(const double* __restrict__ h, double* __restrict__ inv) {
  double L[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) L[i][j] = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    double s = h[UT(NB, j, j)];
#pragma unroll
    for (int k = 0; k < NB; ++k) if (k < j) s -= L[j][k] * L[j][k];
    s = (s > DIAG_FLOOR) ? s : DIAG_FLOOR;
    const double l = sqrt(s);
    L[j][j] = l;
    const double il = 1.0 / l;
#pragma unroll
    for (int i = 0; i < NB; ++i) if (i > j) {
      double t = h[UT(NB, j, i)];
#pragma unroll
      for (int k = 0; k < NB; ++k) if (k < j) t -= L[i][k] * L[j][k];
      L[i][j] = t * il;
    }
  }
  // Linv (lower)
  double Li[NB][NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) Li[i][j] = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    Li[j][j] = 1.0 / L[j][j];
#pragma unroll
    for (int i = 0; i < NB; ++i) if (i > j) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < NB; ++k) if (k >= j && k < i) t -= L[i][k] * Li[k][j];
      Li[i][j] = t / L[i][i];
    }
  }
  // inv = Li^T Li
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) if (j >= i) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < NB; ++k) if (k >= j) t += Li[k][i] * Li[k][j];
      inv[UT(NB, i, j)] = t;
    }
}
__device__ inline void spd6_inverse(const double* __restrict__ h, double* __restrict__ inv) { spdN_inverse<6>(h, inv); }

}  // namespace ba
