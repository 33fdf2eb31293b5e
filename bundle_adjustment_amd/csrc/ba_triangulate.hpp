// Batched two-view triangulation + cheirality test: the step of the reference's pipeline
// that creates the landmarks bundle adjustment refines, src/pipeline.py:315-336 (_triangulate_points):
//     P1 = K [I | 0],  P2 = K [R_rel | t_rel]
//     X_h = cv2.triangulatePoints(P1, P2, pts1^T, pts2^T)          (linear DLT, one 4x4 system per point)
//     X   = X_h[:3] / (X_h[3] + 1e-6)
//     keep X when z > 0 in camera 1 and (R_rel X + t_rel).z > 0 in camera 2
// cv2.triangulatePoints (OpenCV, not installed here: restated from its published algorithm) stacks, per point,
//     A = [x1 P1[2] - P1[0];  y1 P1[2] - P1[1];  x2 P2[2] - P2[0];  y2 P2[2] - P2[1]]        (4x4)
// and returns the right singular vector of A's smallest singular value.  Here: thread per point, one-sided Jacobi SVD of
// the 4x4 A itself (jacobi_svd4), column of V that belongs to the smallest singular value.  The sign of a singular vector
// is arbitrary (and the reference's "+ 1e-6" makes its result depend on it at the 1e-6 level): this kernel and the
// oracle fix it as X_h[3] >= 0.  Parity unpinned at the cv2 boundary, like every other cv2 call of the path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ba {

struct TriView { double P1[12], P2[12], R[9], t[3]; };

// One-sided (Hestenes) Jacobi SVD of the 4x4 A itself: plane rotations from the right make the columns of U = A V mutually
// orthogonal; their norms are then the singular values and the columns of V the right singular vectors.  Working on A
// rather than on A^T A keeps the conditioning of the problem (A^T A squares it: for a low-parallax pair -- a new keyframe
// right after the last one -- the eigenvector of the smallest eigenvalue of A^T A loses half the digits; measured against
// LAPACK's SVD at a 1 mm baseline: 7e-8 relative through A^T A, 3e-10 through A).
__device__ inline void jacobi_svd4(double (&U)[4][4], double (&V)[4][4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { al += U[k][p] * U[k][p]; be += U[k][q] * U[k][q]; ga += U[k][p] * U[k][q]; }
        if (!(fabs(ga) > 1e-17 * sqrt(al * be))) continue;            // already orthogonal to round-off
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double tt = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(zeta * zeta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double up = U[k][p], uq = U[k][q];
          U[k][p] = c * up - s * uq;
          U[k][q] = s * up + c * uq;
          const double vp = V[k][p], vq = V[k][q];
          V[k][p] = c * vp - s * vq;
          V[k][q] = s * vp + c * vq;
        }
      }
    }
    if (!rotated) break;
  }
}

__global__ void __launch_bounds__(256)
k_triangulate(TriView v, int64_t n, const double2* __restrict__ pts1, const double2* __restrict__ pts2,
              double* __restrict__ xyz, uint8_t* __restrict__ valid) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double2 a = pts1[i], b = pts2[i];
  double A[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    A[0][k] = a.x * v.P1[8 + k] - v.P1[k];
    A[1][k] = a.y * v.P1[8 + k] - v.P1[4 + k];
    A[2][k] = b.x * v.P2[8 + k] - v.P2[k];
    A[3][k] = b.y * v.P2[8 + k] - v.P2[4 + k];
  }
  double V[4][4];
  jacobi_svd4(A, V);                        // A <- A V: column norms = singular values
  double sv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) sv[k] = A[0][k] * A[0][k] + A[1][k] * A[1][k] + A[2][k] * A[2][k] + A[3][k] * A[3][k];
  int best = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k) if (sv[k] < sv[best]) best = k;
  double X[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {           // (selection without dynamic register indexing)
    X[k] = V[k][0];
    if (best == 1) X[k] = V[k][1];
    if (best == 2) X[k] = V[k][2];
    if (best == 3) X[k] = V[k][3];
  }
  if (X[3] < 0.0) { X[0] = -X[0]; X[1] = -X[1]; X[2] = -X[2]; X[3] = -X[3]; }
  const double iw = 1.0 / (X[3] + 1e-6);                       // src/pipeline.py:324
  const double x = X[0] * iw, y = X[1] * iw, z = X[2] * iw;
  const double z2 = v.R[6] * x + v.R[7] * y + v.R[8] * z + v.t[2];
  xyz[3 * i] = x; xyz[3 * i + 1] = y; xyz[3 * i + 2] = z;
  valid[i] = (z > 0.0 && z2 > 0.0) ? 1 : 0;                      // :328-334
}

}  // namespace ba
