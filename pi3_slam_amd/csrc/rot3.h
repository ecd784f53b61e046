// Closest proper rotation to a 3x3 matrix (orthogonal Procrustes with the reflection fix), in double, one lane.
//
// R = argmax_{R in SO(3)} sum_ij R_ij * A_ij.  This is what the reference gets from an SVD plus a determinant fix:
//   - CameraHead.svd_orthogonalize (pi3/models/layers/camera_head.py:74-93): with A = row-normalised 3x3 output,
//     A^T = U S V^T, R = V diag(1,1,det(V U^T)) U^T;
//   - the closed-form Sim(3) rotation (Umeyama 1991), A = cross-covariance.
// Instead of an SVD we take the dominant eigenvector of Horn's symmetric 4x4 matrix (Horn 1987, built from S = A^T):
// it yields the same maximiser, never returns a reflection, and needs only a 4x4 Jacobi sweep.
#pragma once

__host__ __device__ inline void nearest_rotation_d(const double A[9], double R[9]) {
  // S_ab = A_ba
  const double Sxx = A[0], Sxy = A[3], Sxz = A[6];
  const double Syx = A[1], Syy = A[4], Syz = A[7];
  const double Szx = A[2], Szy = A[5], Szz = A[8];
  double N[4][4] = {
      {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
      {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
      {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
      {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < 4; ++p) {
      diag += N[p][p] * N[p][p];
      for (int q = p + 1; q < 4; ++q) off += N[p][q] * N[p][q];
    }
    if (off <= 1e-30 * (diag + 1e-300)) break;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = N[p][q];
        if (apq == 0.0) continue;
        const double theta = (N[q][q] - N[p][p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {  // columns p, q of N
          const double nkp = N[k][p], nkq = N[k][q];
          N[k][p] = c * nkp - s * nkq;
          N[k][q] = s * nkp + c * nkq;
        }
        for (int k = 0; k < 4; ++k) {  // rows p, q of N
          const double npk = N[p][k], nqk = N[q][k];
          N[p][k] = c * npk - s * nqk;
          N[q][k] = s * npk + c * nqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq;
          V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i)
    if (N[i][i] > N[best][best]) best = i;
  double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
  const double nn = 1.0 / sqrt(w * w + x * x + y * y + z * z);
  w *= nn; x *= nn; y *= nn; z *= nn;
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
