// Bundle adjustment of one chunk on the device: the refinement the reference runs through
// pytheia / Ceres after building a chunk reconstruction (utils/chunk_reconstruction.py:188-219: 10 iterations, Huber
// 2.0, DENSE_SCHUR, then SetOutlierTracksToUnestimated(…, 2, 0.25)) and after each chunk alignment with pose priors on
// the overlap views (utils/reconstruction_alignment.py:107-171: orientation prior cov 2 I, position prior cov 25 I,
// 50 iterations, Huber 3.0, outliers (3, 0.25)).  pytheia 0.2.9 / Ceres are not vendored and not installable offline:
// the algorithm is RESTATED from the published method (Levenberg-Marquardt trust region with the Schur complement on
// the points, Huber loss as iteratively re-weighted least squares) - parity with Ceres' iterates is UNPINNED
// (oracle/ba_ref.py restates the same algorithm in numpy and cross-checks the optimum with scipy).
//
// Problem layout (dense, no index lists): a track is (source frame s, keypoint k); camera t observes it iff
// valid[s][t][k].  Observations are what pi3_project_observations emits, with the diagonal t == s holding the keypoint
// pixel itself.  N <= 128 cameras, K keypoints per frame, fp64 throughout.
//   camera t: R_t (world -> camera, row-major 9) and centre C_t (3): pose[t] = [R | C] (12 doubles); intrinsics
//             (fx, fy, cx, cy) fixed (Theia's default intrinsics_to_optimize = NONE)
//   residual: r = (fx x/z + cx - u, fy y/z + cy - v),  (x, y, z) = R (X - C)
//   update:   R <- exp([dw]x) R,  C <- C + dC,  X <- X + dX
// One LM iteration = fixed sequence of kernels; accept / reject and the trust-region radius live in a device-side state
// block (no host synchronisation inside the loop).  Every reduction has a fixed order: results do not depend on
// scheduling.
#include "common.h"

#define BA_MAXN 128
#define BA_SLICES 32

struct BaState {
  double cost, cost_new, radius, decrease, model_change, iters, accepted_steps, done, initial_cost, chol_fail, accept;
};

struct BaProblem {
  const float* uv;            // [N][N][K][2]  (source, target, keypoint)
  const uint8_t* valid;       // [N][N][K]
  const float* uvT;           // [N][K][N][2]  (source, keypoint, target)
  const uint8_t* validT;      // [N][K][N]
  const double* intr;         // [N][4]
  const double* prior_R;      // [N][9] or null
  const double* prior_C;      // [N][3]
  const uint8_t* prior_flag;  // [N]
  double sqrt_info_rot, sqrt_info_pos, huber;
  int N, K;
  int homogeneous;            // 1: tracks step in the tangent space of their homogeneous 4-vector (see ba_point_frame)
  int invdepth;               // 1: one inverse-depth parameter per track (see "inverse depth" below)
};

__device__ __forceinline__ double huber_rho(double s, double a, double& w) {
  if (s <= a * a) { w = 1.0; return s; }
  const double r = sqrt(s);
  w = a / r;
  return 2.0 * a * r - a * a;
}

// The reference's point parametrization (ba_options.use_homogeneous_point_parametrization = True unless
// --use-inverse-depth: utils/reconstruction_alignment.py:147-152, utils/chunk_reconstruction.py:199-204): a track is the 4-vector
// h = [X, 1] / |[X, 1]| under ceres::HomogeneousVectorParameterization - the solver steps delta (3) in the tangent space of
// h's unit sphere, h' = Plus(h, delta) = H(h) [sin(|delta|/2) delta / |delta|, cos(|delta|/2)] with the Householder
// reflection H(h) = I - beta v v^T that maps h onto e_4.  The objective is the same function of X = h[:3] / h[3]; what
// changes is the coordinate system the LM damping (diag of J^T J) acts in, i.e. the path, not the optimum (oracle:
// ba_ref.point_frames / points_plus; tests/test_ba_oracle.py measures the difference).  h is re-derived from X each
// iteration (identical to carrying it while h[3] > 0).
//   ba_point_frame: T = d X / d delta at 0 = (1 / w) [I | -X] . 0.5 H[:, :3]   (row-major 3 x 3), w = h[3] = 1 / |[X, 1]|
struct BaFrame { double T[9]; double v[4]; double beta, w; };
__device__ __forceinline__ void ba_point_frame(const double X[3], BaFrame& f) {
  const double n = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + 1.0);
  const double w = 1.0 / n;
  const double h[3] = {X[0] * w, X[1] * w, X[2] * w};
  const double sigma = h[0] * h[0] + h[1] * h[1] + h[2] * h[2];
  f.w = w;
  f.v[3] = 1.0;
  if (sigma <= 1e-300) {           // X = 0: h = e_4 already, H = I (pivot w > 0)
    f.beta = 0.0;
    f.v[0] = h[0]; f.v[1] = h[1]; f.v[2] = h[2];
  } else {
    const double mu = sqrt(w * w + sigma);
    const double vp = -sigma / (w + mu);              // pivot > 0 branch of ceres::internal::ComputeHouseholderVector
    f.beta = 2.0 * vp * vp / (sigma + vp * vp);
    f.v[0] = h[0] / vp; f.v[1] = h[1] / vp; f.v[2] = h[2] / vp;
  }
  const double s = 0.5 * n;                            // 0.5 |h| / w with |h| = 1
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double Hab = (a == b ? 1.0 : 0.0) - f.beta * f.v[a] * f.v[b];
      const double H3b = -f.beta * f.v[3] * f.v[b];
      f.T[3 * a + b] = s * (Hab - X[a] * H3b);
    }
}
// X after the tangent step d on the sphere of its homogeneous vector
__device__ __forceinline__ void ba_point_plus(const double X[3], const BaFrame& f, const double d[3], double Xn[3]) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (nd == 0.0) { Xn[0] = X[0]; Xn[1] = X[1]; Xn[2] = X[2]; return; }
  const double sc = sin(0.5 * nd) / nd;
  double y[4] = {d[0] * sc, d[1] * sc, d[2] * sc, cos(0.5 * nd)};
  const double vy = f.beta * (f.v[0] * y[0] + f.v[1] * y[1] + f.v[2] * y[2] + f.v[3] * y[3]);
#pragma unroll
  for (int c = 0; c < 4; ++c) y[c] -= f.v[c] * vy;     // |h| = 1
  Xn[0] = y[0] / y[3]; Xn[1] = y[1] / y[3]; Xn[2] = y[2] / y[3];
}

// residual + Jacobians of one observation; returns false when the point is not in front of the camera.
// T (or null): the track's frame; the point columns Jp then refer to the tangent step (Jc never does).
__device__ __forceinline__ bool ba_project(const double* pose, const double* in4, const double* X, double u, double v,
                                           double r[2], double Jc[2][6], double Jp[2][3], const double* T = nullptr) {
  const double* R = pose;
  const double d0 = X[0] - pose[9], d1 = X[1] - pose[10], d2 = X[2] - pose[11];
  const double x = R[0] * d0 + R[1] * d1 + R[2] * d2;
  const double y = R[3] * d0 + R[4] * d1 + R[5] * d2;
  const double z = R[6] * d0 + R[7] * d1 + R[8] * d2;
  if (!(z > 1e-9)) return false;
  const double iz = 1.0 / z, fx = in4[0], fy = in4[1];
  r[0] = fx * x * iz + in4[2] - u;
  r[1] = fy * y * iz + in4[3] - v;
  const double a0 = fx * iz, a2 = -fx * x * iz * iz, b1 = fy * iz, b2 = -fy * y * iz * iz;   // d(u,v)/d(x,y,z)
  // point: Jpi R
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    Jp[0][c] = a0 * R[c] + a2 * R[6 + c];
    Jp[1][c] = b1 * R[3 + c] + b2 * R[6 + c];
  }
  // rotation (local): dp = dw x p  ->  Jpi (-[p]x)
  Jc[0][0] = a2 * y;            Jc[0][1] = a0 * z - a2 * x;   Jc[0][2] = -a0 * y;
  Jc[1][0] = -b1 * z + b2 * y;  Jc[1][1] = -b2 * x;           Jc[1][2] = b1 * x;
  // centre: -Jpi R
#pragma unroll
  for (int c = 0; c < 3; ++c) { Jc[0][3 + c] = -Jp[0][c]; Jc[1][3 + c] = -Jp[1][c]; }
  if (T) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const double j0 = Jp[q][0], j1 = Jp[q][1], j2 = Jp[q][2];
#pragma unroll
      for (int c = 0; c < 3; ++c) Jp[q][c] = j0 * T[c] + j1 * T[3 + c] + j2 * T[6 + c];
    }
  }
  return true;
}

// deterministic block reduction of `n` values per thread (fixed tree through LDS); result valid in thread 0
template <int NT>
__device__ __forceinline__ double ba_block_sum(double v, double* red, int tid) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  if (tid == 0)
    for (int w = 0; w < NT / 64; ++w) s += red[w];
  return s;
}

// ---- pass 1: per track: C_i = sum w Jp^T Jp, g_i = sum w Jp^T r, cost_i = sum rho   (thread per track)
__global__ __launch_bounds__(256) void ba_linearize_points(BaProblem pb, const double* __restrict__ pts,
                                                           const double* __restrict__ poses, double* __restrict__ Cblk,
                                                           double* __restrict__ gp, double* __restrict__ cost_part,
                                                           const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double cost = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double C[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    BaFrame fr;
    if (pb.homogeneous) ba_point_frame(X, fr);
    for (int t = 0; t < N; ++t) {
      const long o = ((long)s * N + t) * K + k;
      if (!pb.valid[o]) continue;
      double r[2], Jc[2][6], Jp[2][3];
      if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp,
                      pb.homogeneous ? fr.T : nullptr))
        continue;
      double w;
      cost += huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
      C[0] += w * (Jp[0][0] * Jp[0][0] + Jp[1][0] * Jp[1][0]);
      C[1] += w * (Jp[0][0] * Jp[0][1] + Jp[1][0] * Jp[1][1]);
      C[2] += w * (Jp[0][0] * Jp[0][2] + Jp[1][0] * Jp[1][2]);
      C[3] += w * (Jp[0][1] * Jp[0][1] + Jp[1][1] * Jp[1][1]);
      C[4] += w * (Jp[0][1] * Jp[0][2] + Jp[1][1] * Jp[1][2]);
      C[5] += w * (Jp[0][2] * Jp[0][2] + Jp[1][2] * Jp[1][2]);
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] += w * (Jp[0][c] * r[0] + Jp[1][c] * r[1]);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) Cblk[6 * i + c] = C[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) gp[3 * i + c] = g[c];
  }
  const double tot = ba_block_sum<256>(cost, red, tid);
  if (tid == 0) cost_part[blockIdx.x] = 0.5 * tot;
}

// cost only (candidate parameters)
__global__ __launch_bounds__(256) void ba_cost_points(BaProblem pb, const double* __restrict__ pts,
                                                      const double* __restrict__ poses, double* __restrict__ cost_part,
                                                      const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double cost = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    for (int t = 0; t < N; ++t) {
      const long o = ((long)s * N + t) * K + k;
      if (!pb.valid[o]) continue;
      double r[2], Jc[2][6], Jp[2][3];
      if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp))
        continue;
      double w;
      cost += huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
    }
  }
  const double tot = ba_block_sum<256>(cost, red, tid);
  if (tid == 0) cost_part[blockIdx.x] = 0.5 * tot;
}

// log map of a rotation matrix (row-major), angle-axis vector
__device__ __forceinline__ void ba_log_so3(const double* R, double w[3]) {
  const double tr = R[0] + R[4] + R[8];
  double c = 0.5 * (tr - 1.0);
  c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
  const double th = acos(c);
  const double vx = R[7] - R[5], vy = R[2] - R[6], vz = R[3] - R[1];
  double f;
  if (th < 1e-8) f = 0.5;
  else f = th / (2.0 * sin(th));       // (angles near pi do not occur between a pose and its prior here)
  w[0] = f * vx; w[1] = f * vy; w[2] = f * vz;
}

// ---- inverse depth (the reference's --use-inverse-depth: reconstruction.InitializeInverseDepth() +
// ba_options.use_inverse_depth_parametrization = True, utils/chunk_reconstruction.py:187-204,
// utils/reconstruction_alignment.py:147-152).  A track (s, k) is ONE parameter, the inverse depth rho along the bearing b
// of its keypoint in its reference view s (the first view that observed it: its own frame):
//     X = C_s + R_s^T (b / rho),   b = ((u - cx) / fx, (v - cy) / fy, 1)
// so every observation by another camera t depends on (pose_t, pose_s, rho) and the reference view's own observation is
// met by construction (no residual).  The stored points stay Euclidean and ON their rays (snapped at the start, every
// candidate built from the candidate pose and rho), so a kernel derives p_s = R_s (X - C_s) and rho = 1 / p_s.z from
// them.  Chain rule through X: dX/dC_s = I, dX/dw_s = R_s^T [p_s]x (R_s <- exp([w]x) R_s), dX/drho = -(X - C_s) / rho.
// The linear algebra reuses the Euclidean kernels with the scalar embedded in their 3 x 3 point blocks
// (C_i = diag(c, 1, 1), g_i = (g, 0, 0), E_it = [e | 0 | 0]); what is new are the reference camera's entries in a track's
// E row, its diagonal terms, and the direct camera-camera blocks w Ja^T Jt of the normal equations (ba_id_cross_blocks).
__device__ __forceinline__ void ba_id_chain(const double* pose_s, const double X[3], const double Jp[2][3], double Ja[2][6],
                                            double jr[2]) {
  const double* R = pose_s;
  const double d[3] = {X[0] - pose_s[9], X[1] - pose_s[10], X[2] - pose_s[11]};
  const double p[3] = {R[0] * d[0] + R[1] * d[1] + R[2] * d[2], R[3] * d[0] + R[4] * d[1] + R[5] * d[2],
                       R[6] * d[0] + R[7] * d[1] + R[8] * d[2]};
  const double inv_rho = p[2];      // rho = 1 / p.z
  // M = R^T [p]x, column j = R^T (p x e_j):  [p]x = [[0, -pz, py], [pz, 0, -px], [-py, px, 0]]
  const double c0[3] = {0.0, p[2], -p[1]}, c1[3] = {-p[2], 0.0, p[0]}, c2[3] = {p[1], -p[0], 0.0};
  double M[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    M[a][0] = R[a] * c0[0] + R[3 + a] * c0[1] + R[6 + a] * c0[2];
    M[a][1] = R[a] * c1[0] + R[3 + a] * c1[1] + R[6 + a] * c1[2];
    M[a][2] = R[a] * c2[0] + R[3 + a] * c2[1] + R[6 + a] * c2[2];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Ja[q][c] = Jp[q][0] * M[0][c] + Jp[q][1] * M[1][c] + Jp[q][2] * M[2][c];
      Ja[q][3 + c] = Jp[q][c];
    }
    jr[q] = -(Jp[q][0] * d[0] + Jp[q][1] * d[1] + Jp[q][2] * d[2]) * inv_rho;      // Jp . dX/drho, dX/drho = -(X - C_s) / rho
  }
}

// ---- pass 2: per camera t: B_t = sum w Jc^T Jc (21 values), g_t = sum w Jc^T r, + pose priors.  One WG per camera.
__global__ __launch_bounds__(256) void ba_camera_blocks(BaProblem pb, const double* __restrict__ pts,
                                                        const double* __restrict__ poses, double* __restrict__ Bblk,
                                                        double* __restrict__ gc, double* __restrict__ prior_cost,
                                                        const BaState* st) {
  __shared__ double red[4];
  __shared__ double acc[27];
  const int N = pb.N, K = pb.K, t = blockIdx.x, tid = threadIdx.x;
  if (st->done != 0.0) return;
  double a[27];
#pragma unroll
  for (int c = 0; c < 27; ++c) a[c] = 0.0;
  for (long i = tid; i < (long)N * K; i += 256) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const long o = ((long)s * N + t) * K + k;
    if (!pb.valid[o]) continue;
    if (pb.invdepth && s == t) continue;      // a track's reference-view observation carries no residual
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double r[2], Jc[2][6], Jp[2][3];
    if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp)) continue;
    double w;
    huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
    int q = 0;
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int c = p; c < 6; ++c) a[q++] += w * (Jc[0][p] * Jc[0][c] + Jc[1][p] * Jc[1][c]);
#pragma unroll
    for (int p = 0; p < 6; ++p) a[21 + p] += w * (Jc[0][p] * r[0] + Jc[1][p] * r[1]);
  }
  if (pb.invdepth) {      // camera t as the REFERENCE view of its own tracks: every other observation of them moves with it
    for (long e = tid; e < (long)K * N; e += 256) {
      const int k = (int)(e / N), t2 = (int)(e - (long)k * N);
      if (t2 == t) continue;
      const long i = (long)t * K + k;
      const long o = ((long)t * N + t2) * K + k;
      if (!pb.valid[o] || !pb.valid[((long)t * N + t) * K + k]) continue;
      const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
      double r[2], Jc[2][6], Jp[2][3], Ja[2][6], jr[2];
      if (!ba_project(poses + 12 * t2, pb.intr + 4 * t2, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp)) continue;
      ba_id_chain(poses + 12 * t, X, Jp, Ja, jr);
      double w;
      huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
      int q = 0;
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int c = p; c < 6; ++c) a[q++] += w * (Ja[0][p] * Ja[0][c] + Ja[1][p] * Ja[1][c]);
#pragma unroll
      for (int p = 0; p < 6; ++p) a[21 + p] += w * (Ja[0][p] * r[0] + Ja[1][p] * r[1]);
    }
  }
  for (int c = 0; c < 27; ++c) {
    const double v = ba_block_sum<256>(a[c], red, tid);
    if (tid == 0) acc[c] = v;
  }
  __syncthreads();
  if (tid == 0) {
    double pc = 0.0;
    if (pb.prior_flag && pb.prior_flag[t]) {
      // orientation prior: r = sr * log(R R0^T), dr/dw ~ I;  position prior: r = sp * (C - C0)
      const double* R = poses + 12 * t;
      const double* R0 = pb.prior_R + 9 * t;
      double E[9];
      for (int p = 0; p < 3; ++p)
        for (int c = 0; c < 3; ++c) E[3 * p + c] = R[3 * p] * R0[3 * c] + R[3 * p + 1] * R0[3 * c + 1] + R[3 * p + 2] * R0[3 * c + 2];
      double w3[3];
      ba_log_so3(E, w3);
      const double sr2 = pb.sqrt_info_rot * pb.sqrt_info_rot, sp2 = pb.sqrt_info_pos * pb.sqrt_info_pos;
      const int dq[6] = {0, 6, 11, 15, 18, 20};   // packed index of the diagonal entries (p, p)
      for (int p = 0; p < 3; ++p) {
        acc[dq[p]] += sr2;
        acc[21 + p] += sr2 * w3[p];
        pc += 0.5 * sr2 * w3[p] * w3[p];
        const double dcp = poses[12 * t + 9 + p] - pb.prior_C[3 * t + p];
        acc[dq[3 + p]] += sp2;
        acc[24 + p] += sp2 * dcp;
        pc += 0.5 * sp2 * dcp * dcp;
      }
    }
    for (int c = 0; c < 21; ++c) Bblk[21 * t + c] = acc[c];
    for (int c = 0; c < 6; ++c) gc[6 * t + c] = acc[21 + c];
    prior_cost[t] = pc;
  }
}

// prior cost only (candidate parameters)
__global__ void ba_prior_cost(BaProblem pb, const double* __restrict__ poses, double* __restrict__ prior_cost,
                              const BaState* st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (st->done != 0.0 || t >= pb.N) return;
  double pc = 0.0;
  if (pb.prior_flag && pb.prior_flag[t]) {
    const double* R = poses + 12 * t;
    const double* R0 = pb.prior_R + 9 * t;
    double E[9], w3[3];
    for (int p = 0; p < 3; ++p)
      for (int c = 0; c < 3; ++c) E[3 * p + c] = R[3 * p] * R0[3 * c] + R[3 * p + 1] * R0[3 * c + 1] + R[3 * p + 2] * R0[3 * c + 2];
    ba_log_so3(E, w3);
    const double sr2 = pb.sqrt_info_rot * pb.sqrt_info_rot, sp2 = pb.sqrt_info_pos * pb.sqrt_info_pos;
    for (int p = 0; p < 3; ++p) {
      const double dcp = poses[12 * t + 9 + p] - pb.prior_C[3 * t + p];
      pc += 0.5 * sr2 * w3[p] * w3[p] + 0.5 * sp2 * dcp * dcp;
    }
  }
  prior_cost[t] = pc;
}

// damped inverse of a track's 3x3 block: (C + D)^-1 with D = clamp(diag(C)) / radius; returns false if singular
__device__ __forceinline__ bool ba_point_inverse(const double* C6, double radius, double Ci[6], double D[3]) {
  double c00 = C6[0], c01 = C6[1], c02 = C6[2], c11 = C6[3], c12 = C6[4], c22 = C6[5];
  D[0] = fmin(fmax(c00, 1e-6), 1e32) / radius;
  D[1] = fmin(fmax(c11, 1e-6), 1e32) / radius;
  D[2] = fmin(fmax(c22, 1e-6), 1e32) / radius;
  c00 += D[0]; c11 += D[1]; c22 += D[2];
  const double m0 = c11 * c22 - c12 * c12, m1 = c02 * c12 - c01 * c22, m2 = c01 * c12 - c02 * c11;
  const double det = c00 * m0 + c01 * m1 + c02 * m2;
  if (!(fabs(det) > 1e-300)) return false;
  const double id = 1.0 / det;
  Ci[0] = m0 * id; Ci[1] = m1 * id; Ci[2] = m2 * id;
  Ci[3] = (c00 * c22 - c02 * c02) * id; Ci[4] = (c01 * c02 - c00 * c12) * id; Ci[5] = (c00 * c11 - c01 * c01) * id;
  return true;
}

// ---- pass 3a: per observation (track i, camera t) the 6 x 3 block E_it = w Jc^T Jp of the normal equations, once per
// iteration, in the [track][camera] layout: a track's blocks are contiguous (144 B each), which is what pass 3b reads.
// ok[i N + t] = 1 when the observation exists and the point is in front of the camera.
__global__ __launch_bounds__(256) void ba_obs_blocks(BaProblem pb, const double* __restrict__ pts,
                                                     const double* __restrict__ poses, double* __restrict__ Eblk,
                                                     uint8_t* __restrict__ ok, const BaState* st) {
  const int N = pb.N;
  const long o = (long)blockIdx.x * 256 + threadIdx.x;
  if (st->done != 0.0 || o >= (long)N * pb.K * N) return;
  const long i = o / N;
  const int t = (int)(o - i * N);
  uint8_t good = 0;
  if (pb.validT[o]) {
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double r[2], Jc[2][6], Jp[2][3], w;
    BaFrame fr;
    if (pb.homogeneous) ba_point_frame(X, fr);
    if (ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uvT[2 * o], (double)pb.uvT[2 * o + 1], r, Jc, Jp,
                   pb.homogeneous ? fr.T : nullptr)) {
      huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
      double* E = Eblk + 18 * o;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        E[3 * a + 0] = w * (Jc[0][a] * Jp[0][0] + Jc[1][a] * Jp[1][0]);
        E[3 * a + 1] = w * (Jc[0][a] * Jp[0][1] + Jc[1][a] * Jp[1][1]);
        E[3 * a + 2] = w * (Jc[0][a] * Jp[0][2] + Jc[1][a] * Jp[1][2]);
      }
      good = 1;
    }
  }
  ok[o] = good;
}

// ---- pass 3b: Schur complement rows.  WG (camera j, slice): lanes = cameras k' <= j (S is symmetric and only its lower
// triangle is factored).  S_part[slice][6j+a][6k'+b] -= sum_i (E_ij Cinv_i) E_ik'^T and the slice's share of
// rhs_j = -(g_j - sum_i E_ij Cinv_i g_i).  Per track of camera j: one broadcast read of E_ij, one 144-byte read per lane.
__global__ __launch_bounds__(BA_MAXN) void ba_schur_rows(BaProblem pb, const double* __restrict__ Eblk,
                                                         const uint8_t* __restrict__ ok,
                                                         const double* __restrict__ Cblk, const double* __restrict__ gp,
                                                         double* __restrict__ S_part, double* __restrict__ rhs_part,
                                                         const BaState* st) {
  const int N = pb.N, K = pb.K, j = blockIdx.x, slice = blockIdx.y, kp = threadIdx.x;
  if (st->done != 0.0) return;
  const int n6 = 6 * N;
  const double radius = st->radius;
  double S[6][6], rj[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    rj[a] = 0.0;
#pragma unroll
    for (int b = 0; b < 6; ++b) S[a][b] = 0.0;
  }
  const bool lane_cam = kp <= j;                      // lower triangle (kp < N follows from j < N)
  const bool wave_has_cams = (kp & ~63) <= j;         // the second wave has nothing to add while j < 64 (rhs_j is wave 0's)
  const int s0 = (N * slice) / BA_SLICES, s1 = (N * (slice + 1)) / BA_SLICES;
  // tracks of the slice, 64 at a time: each lane tests one track for an observation by camera j, the ballot is the
  // list of tracks to work on (both waves of the workgroup scan the same 64 tracks and walk the same bits, in
  // ascending track order: the sums keep the order a sequential walk gives them)
  const long i0 = (long)s0 * K, i1 = (long)s1 * K;
  for (long base = i0; base < i1; base += 64) {
    const long il = base + (kp & 63);
    unsigned long long seen = __ballot(il < i1 && ok[il * N + j] != 0);
    if (!wave_has_cams) seen = 0;
    while (seen) {
      const int bit = __builtin_ctzll(seen);
      seen &= seen - 1;
      const long i = base + bit;
      double Ci[6], D[3];
      if (!ba_point_inverse(Cblk + 6 * i, radius, Ci, D)) continue;
      const double* Ej = Eblk + 18 * (i * N + j);     // uniform address: one broadcast load
      double Y[6][3];   // E_ij Cinv
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double e0 = Ej[3 * a], e1 = Ej[3 * a + 1], e2 = Ej[3 * a + 2];
        Y[a][0] = e0 * Ci[0] + e1 * Ci[1] + e2 * Ci[2];
        Y[a][1] = e0 * Ci[1] + e1 * Ci[3] + e2 * Ci[4];
        Y[a][2] = e0 * Ci[2] + e1 * Ci[4] + e2 * Ci[5];
      }
      const double g0 = gp[3 * i], g1 = gp[3 * i + 1], g2 = gp[3 * i + 2];
#pragma unroll
      for (int a = 0; a < 6; ++a) rj[a] += Y[a][0] * g0 + Y[a][1] * g1 + Y[a][2] * g2;
      // this lane's camera k'
      if (!lane_cam || !ok[i * N + kp]) continue;
      const double* Ek = Eblk + 18 * (i * N + kp);
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const double e0 = Ek[3 * b], e1 = Ek[3 * b + 1], e2 = Ek[3 * b + 2];
#pragma unroll
        for (int a = 0; a < 6; ++a) S[a][b] -= Y[a][0] * e0 + Y[a][1] * e1 + Y[a][2] * e2;
      }
    }
  }
  if (kp < N) {      // cameras above j: zeros (the upper triangle is never read by the factorisation)
    double* out = S_part + (long)slice * n6 * n6;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) out[(long)(6 * j + a) * n6 + 6 * kp + b] = S[a][b];
  }
  if (kp == 0)
    for (int a = 0; a < 6; ++a) rhs_part[((long)slice * N + j) * 6 + a] = rj[a];
}

// ---- pass 3b, matrix-core form (round 4).  S_part[slice] -= Y E^T is a GEMM over k = (track, coordinate): rows
// (camera j, a) of Y = E Cinv against rows (camera k', b) of E.  In the [track][camera][6][3] layout of Eblk the 8
// cameras of a tile block are one contiguous 1 152-byte run per track, so a wave that owns a 48 x 48 tile (8 x 8 cameras,
// 3 x 3 MFMA tiles of v_mfma_f64_16x16x4_f64; the fourth k of an MFMA is padding) reads its operands straight from
// global memory: lane (row r, coordinate c) takes Y[3 r + c] as the A operand and E[3 q + c] as the B operand, where
// Y = E Cinv is written once per iteration by ba_y_blocks (with the track's damped inverse from ba_point_inverses),
// which also zeroes E and Y where there is no observation: the loop has no validity test and no branch.  Same slices
// and the same ascending track order inside a slice as ba_schur_rows, so the result is reproducible from run to run;
// against that kernel it differs by the rounding of the 3-term dot products (the oracle gates are 1e-9).
// 2.2 ms -> see profiles/EXPERIMENTS.md at 100 cameras x 20 000 tracks: the row form spends its time in 128-lane
// workgroups of which one lane in five has an observation to work on.
typedef double ba_v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void ba_point_inverses(long nk, const double* __restrict__ Cblk,
                                                         double* __restrict__ Cinv, const BaState* st) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (st->done != 0.0 || i >= nk) return;
  double Ci[6], D[3];
  const bool good = ba_point_inverse(Cblk + 6 * i, st->radius, Ci, D);
#pragma unroll
  for (int q = 0; q < 6; ++q) Cinv[6 * i + q] = good ? Ci[q] : 0.0;
}

// Y = E Cinv per observation, zeros where there is no observation - in Y AND in E itself, so that the tile kernel below
// needs no validity test at all (its loop is loads + MFMAs, which the compiler can pipeline across tracks).
__global__ __launch_bounds__(256) void ba_y_blocks(long nobs, int N, double* __restrict__ Eblk,
                                                   const uint8_t* __restrict__ ok, const double* __restrict__ Cinv,
                                                   double* __restrict__ Yblk, const BaState* st) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;      // one row (observation o, a) of three doubles per thread
  if (st->done != 0.0 || e >= 6 * nobs) return;
  const long o = e / 6;
  double* E = Eblk + 3 * e;
  double* Y = Yblk + 3 * e;
  if (ok[o]) {
    const double* ci = Cinv + 6 * (o / N);
    const double e0 = E[0], e1 = E[1], e2 = E[2];
    Y[0] = e0 * ci[0] + e1 * ci[1] + e2 * ci[2];
    Y[1] = e0 * ci[1] + e1 * ci[3] + e2 * ci[4];
    Y[2] = e0 * ci[2] + e1 * ci[4] + e2 * ci[5];
  } else {
    E[0] = 0.0; E[1] = 0.0; E[2] = 0.0;
    Y[0] = 0.0; Y[1] = 0.0; Y[2] = 0.0;
  }
}

__global__ __launch_bounds__(64) void ba_schur_tiles(BaProblem pb, const double* __restrict__ Eblk,
                                                     const double* __restrict__ Yblk, const double* __restrict__ gp,
                                                     double* __restrict__ S_part, double* __restrict__ rhs_part,
                                                     int nsl, const BaState* st) {
  if (st->done != 0.0) return;
  // (giving all tiles of a slice to one XCD, so that its L2 serves the 91 re-reads of the slice's operands, was measured
  // and is slower: 1.94 against 1.64 ms before the pipelining - the memory side is not what bounds this kernel)
  const int N = pb.N, K = pb.K, lane = threadIdx.x;
  const int slice = blockIdx.y, tile = blockIdx.x;
  int Jb = 0;
  while ((Jb + 1) * (Jb + 2) / 2 <= tile) ++Jb;                 // lower-triangle tile number -> (Jb, Kb), Kb <= Jb
  const int Kb = tile - Jb * (Jb + 1) / 2;
  const int j0 = 8 * Jb, k0 = 8 * Kb, n6 = 6 * N;
  const int r16 = lane & 15, c = lane >> 4;                      // row inside a 16-row MFMA tile; k index (3 = padding)
  const long i0 = (long)((N * slice) / nsl) * K, i1 = (long)((N * (slice + 1)) / nsl) * K;
  ba_v4f64 acc[3][3];
  double racc[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) acc[a][b] = (ba_v4f64){0.0, 0.0, 0.0, 0.0};
  // this lane's rows: r = 16 s + r16 of the block -> camera r / 6 (past the last camera in the last block: nothing to
  // load, the run belongs to the next track), element 3 r + c of the block's run
  bool inA[3], inB[3];
#pragma unroll
  for (int sb = 0; sb < 3; ++sb) {
    const int r = 16 * sb + r16;
    inA[sb] = j0 + r / 6 < N && c < 3;
    inB[sb] = k0 + r / 6 < N && c < 3;
  }
  const int lo = 3 * r16 + (c < 3 ? c : 0);
  const bool rhs_tile = Kb == 0;
  // four tracks per trip: their 24 operand loads (and the rhs terms) are issued before the 36 MFMAs, so the loads of a
  // trip overlap the matrix work of the one before (hipcc does not unroll a loop of unknown trip count by itself here,
  // and a lone track per trip left every wave waiting on memory: 22 % of the f64 MFMA rate)
  auto step = [&](long i, double (&Y)[3], double (&B)[3]) {
    const double* Ya = Yblk + 18 * (i * N + j0) + lo;
    const double* Eb = Eblk + 18 * (i * N + k0) + lo;
    const double g = (rhs_tile && c < 3) ? gp[3 * i + c] : 0.0;
    // unconditional loads at constant offsets from one pointer per operand (lanes past the last camera read the next
    // track's run - the workspace is padded for the last track - and drop the value)
#pragma unroll
    for (int sb = 0; sb < 3; ++sb) {
      const double y = Ya[48 * sb], b = Eb[48 * sb];
      Y[sb] = inA[sb] ? y : 0.0;
      B[sb] = inB[sb] ? b : 0.0;
      racc[sb] += Y[sb] * g;
    }
  };
  auto mma = [&](const double (&Y)[3], const double (&B)[3]) {
#pragma unroll
    for (int sa = 0; sa < 3; ++sa)
#pragma unroll
      for (int sb = 0; sb < 3; ++sb)
        acc[sa][sb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Y[sa], B[sb], acc[sa][sb], 0, 0, 0);
  };
  // software pipeline over groups of BA_TG tracks with two register sets: the loads of group t + 1 are in flight while
  // the MFMAs of group t run
#define BA_TG 2
  auto loadg = [&](long i, double (&Y)[BA_TG][3], double (&B)[BA_TG][3]) {
#pragma unroll
    for (int u = 0; u < BA_TG; ++u) step(i + u, Y[u], B[u]);
  };
  auto mmag = [&](const double (&Y)[BA_TG][3], const double (&B)[BA_TG][3]) {
#pragma unroll
    for (int u = 0; u < BA_TG; ++u) mma(Y[u], B[u]);
  };
  long i = i0;
  const long ngrp = (i1 - i0) / BA_TG;
  if (ngrp > 0) {
    double Ya4[BA_TG][3], Ba4[BA_TG][3], Yb4[BA_TG][3], Bb4[BA_TG][3];
    loadg(i, Ya4, Ba4);
    long gidx = 0;
    for (; gidx + 2 <= ngrp; gidx += 2) {
      loadg(i + BA_TG, Yb4, Bb4);
      mmag(Ya4, Ba4);
      if (gidx + 2 < ngrp) loadg(i + 2 * BA_TG, Ya4, Ba4);
      mmag(Yb4, Bb4);
      i += 2 * BA_TG;
    }
    if (gidx < ngrp) {       // an odd group left: its loads were issued by the last trip (or above when ngrp == 1)
      mmag(Ya4, Ba4);
      i += BA_TG;
    }
  }
#undef BA_TG
  for (; i < i1; ++i) {
    double Y0[3], B0[3];
    step(i, Y0, B0);
    mma(Y0, B0);
  }
  // D[i][j] of a 16 x 16 tile: lane l holds i = l / 16 + 4 v, j = l % 16 in element v (tools/micro/mfma_f64_layout.hip)
  double* out = S_part + (long)slice * n6 * n6;
#pragma unroll
  for (int sa = 0; sa < 3; ++sa)
#pragma unroll
    for (int sb = 0; sb < 3; ++sb)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = 6 * j0 + 16 * sa + c + 4 * v, col = 6 * k0 + 16 * sb + r16;
        if (row < n6 && col < n6) out[(long)row * n6 + col] = -acc[sa][sb][v];
      }
  if (rhs_tile) {     // rhs_j = sum_i Y_ij g_i: the three coordinate lanes of a row add up (lanes l, l + 16, l + 32)
#pragma unroll
    for (int sa = 0; sa < 3; ++sa) {
      double v = racc[sa];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      const int row = 6 * j0 + 16 * sa + r16;
      if (c == 0 && row < n6) rhs_part[(long)slice * n6 + row] = v;
    }
  }
}

// ---- pass 4: the reduced camera system.  S = sum of slices + B + D (lower triangle used), rhs = -(g_c - sum slices),
// then a BLOCKED right-looking Cholesky (panels of BA_NB columns: factor the diagonal block in LDS, solve the panel rows
// against it, rank-BA_NB update of the trailing matrix by a grid of tiles) and blocked triangular solves.  The factor
// goes to its own array L, S is only ever updated below the current panel.  (The first version factored column by
// column in one workgroup with the matrix in L2: 22 ms of the 34 ms an iteration took at N = 100.)
#define BA_NB 48

__global__ __launch_bounds__(256) void ba_assemble_cameras(int N, const double* __restrict__ S_part,
                                                           const double* __restrict__ rhs_part,
                                                           const double* __restrict__ Bblk, const double* __restrict__ gc,
                                                           double* __restrict__ S, double* __restrict__ dcam,
                                                           double* __restrict__ Dcam, int nsl, BaState* st) {
  if (st->done != 0.0) return;
  const int n = 6 * N;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e == 0) st->chol_fail = 0.0;
  if (e < (long)n * n) {
    const int r = (int)(e / n), c = (int)(e - (long)r * n);
    double v = 0.0;
    for (int sl = 0; sl < nsl; ++sl) v += S_part[(long)sl * n * n + e];
    if (r / 6 == c / 6) {      // + B_t (symmetric, packed upper) + the LM diagonal
      const int t = r / 6, a6 = r - 6 * t, b6 = c - 6 * t;
      const int p = a6 < b6 ? a6 : b6, q = a6 < b6 ? b6 : a6;
      double bv = Bblk[21 * t + p * 6 - p * (p - 1) / 2 + (q - p)];
      if (a6 == b6) {
        const double d = fmin(fmax(bv, 1e-6), 1e32) / st->radius;
        Dcam[r] = d;
        bv += d;
      }
      v += bv;
    }
    S[e] = v;
  }
  if (e < n) {
    double v = 0.0;
    for (int sl = 0; sl < nsl; ++sl) v += rhs_part[(long)sl * n + e];
    dcam[e] = -(gc[e] - v);
  }
}

// diagonal block [c0, c0 + nb) of S -> its Cholesky factor in L.  One wave, row r of the block in the registers of lane
// r; per column the scaled column goes through 48 doubles of LDS and every lane folds it into its row (broadcast reads):
// no workgroup barrier in the 48 steps (round 3's 256-thread form spent 20-50 us per block on three barriers a column).
__global__ __launch_bounds__(64) void ba_chol_diag(int n, int c0, int nb, const double* __restrict__ S,
                                                   double* __restrict__ L, BaState* st) {
  __shared__ double col[64];
  const int r = threadIdx.x;
  if (st->done != 0.0 || st->chol_fail != 0.0) return;
  double a[BA_NB];
  const bool mine = r < nb;
  const double* srow = S + (long)(c0 + (mine ? r : 0)) * n + c0;
#pragma unroll
  for (int q = 0; q < BA_NB; ++q) a[q] = srow[q < nb ? q : nb - 1];          // unconditional: 48 loads in flight
#pragma unroll
  for (int q = 0; q < BA_NB; ++q) a[q] = (mine && q <= r && q < nb) ? a[q] : 0.0;
  bool fail = false;
#pragma unroll
  for (int c = 0; c < BA_NB; ++c) {
    if (c < nb && !fail) {          // uniform
      col[r] = a[c];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const double d = col[c];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      if (!(d > 0.0)) {
        fail = true;
      } else {
        const double sd = sqrt(d), ip = 1.0 / sd;
        const double l = (r == c) ? sd : a[c] * ip;
        a[c] = l;
        col[r] = l;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int q = c + 1; q < BA_NB; ++q) a[q] -= l * col[q];     // (entries above a lane's diagonal are never used)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    }
  }
  if (fail) {
    if (r == 0) st->chol_fail = 1.0;
    return;
  }
  if (mine) {
    double* lrow = L + (long)(c0 + r) * n + c0;
#pragma unroll
    for (int q = 0; q < BA_NB; ++q)
      if (q <= r && q < nb) lrow[q] = a[q];
  }
}

// rows below the diagonal block: L[r][c0 + j] = (S[r][c0 + j] - sum_{k<j} L[r][c0 + k] L11[j][k]) / L11[j][j].
// One thread per row with the row in registers, the diagonal block in LDS (broadcast reads at compile-time offsets), the
// 1 128 multiply-adds straight-line code.  (Round 3 kept the rows in LDS as well: two dependent LDS reads per
// multiply-add, 19-47 us per panel; scalar loads of the block from global memory serialise on their latency: 33 us.)
// The right-hand side rides along as row n of the matrix (L y = rhs is then solved by the factorisation itself: the
// panel step gives y's segment, the trailing update folds it into the rest of rhs): row index n reads and writes `rhs`.
__global__ __launch_bounds__(64) void ba_chol_panel(int n, int c0, int nb, const double* __restrict__ S,
                                                    const double* __restrict__ Ldiag, double* __restrict__ L,
                                                    double* __restrict__ rhs, const BaState* st) {
  __shared__ double l11[BA_NB][BA_NB];
  if (st->done != 0.0 || st->chol_fail != 0.0) return;
  // (a predicated load followed by its LDS store costs one global round trip per element - hipcc puts the wait in front
  // of every store: 36 round trips were 30 of this kernel's 37 us)
  {
    double tmp[BA_NB * BA_NB / 64];
#pragma unroll
    for (int q = 0; q < BA_NB * BA_NB / 64; ++q) {       // unconditional loads at clamped indices: in flight together
      const int e = q * 64 + threadIdx.x, i = e / BA_NB, j = e - i * BA_NB;
      const int ii = i < nb ? i : nb - 1, jj = j < ii ? j : ii;
      tmp[q] = Ldiag[(long)(c0 + ii) * n + c0 + jj];
    }
#pragma unroll
    for (int q = 0; q < BA_NB * BA_NB / 64; ++q) {
      const int e = q * 64 + threadIdx.x, i = e / BA_NB, j = e - i * BA_NB;
      l11[i][j] = (i < nb && j <= i) ? tmp[q] : 0.0;
    }
  }
  __syncthreads();
  const int r = c0 + nb + blockIdx.x * 64 + threadIdx.x;
  if (r > n) return;
  double x[BA_NB];
  const double* srow = r < n ? S + (long)r * n + c0 : rhs + c0;
#pragma unroll
  for (int j = 0; j < BA_NB; ++j) x[j] = srow[j < nb ? j : nb - 1];
#pragma unroll
  for (int j = 0; j < BA_NB; ++j) {
    if (j < nb) {        // uniform
      // two partial sums (even / odd k): half the length of the dependent multiply-add chain of a row
      double v0 = x[j], v1 = 0.0;
#pragma unroll
      for (int k = 0; k + 1 < j; k += 2) {
        v0 -= x[k] * l11[j][k];                                // same address in every lane: a broadcast read
        v1 -= x[k + 1] * l11[j][k + 1];
      }
      if (j & 1) v0 -= x[j - 1] * l11[j][j - 1];
      x[j] = (v0 + v1) / l11[j][j];
    }
  }
  double* lrow = r < n ? L + (long)r * n + c0 : rhs + c0;
#pragma unroll
  for (int j = 0; j < BA_NB; ++j)
    if (j < nb) lrow[j] = x[j];
}

// trailing update S[r][q] -= sum_k L[r][c0 + k] L[q][c0 + k] for c0 + nb <= q <= r: one 48 x 48 tile per workgroup
__global__ __launch_bounds__(256) void ba_chol_update(int n, int c0, int nb, double* __restrict__ S,
                                                      const double* __restrict__ L, double* __restrict__ rhs,
                                                      const BaState* st) {
  __shared__ double lr[BA_NB][BA_NB + 1], lq[BA_NB][BA_NB + 1];
  const int tid = threadIdx.x;
  if (blockIdx.x > blockIdx.y) return;       // tiles above the diagonal
  if (st->done != 0.0 || st->chol_fail != 0.0) return;
  const int t0 = c0 + nb, r0 = t0 + blockIdx.y * BA_NB, q0 = t0 + blockIdx.x * BA_NB;
  if (blockIdx.y == gridDim.y - 1) {         // the rhs row: rhs[q] -= sum_k y[c0 + k] L[q][c0 + k]
    const int q = q0 + tid;
    if (tid < BA_NB && q < n) {
      const double* lrow = L + (long)q * n + c0;
      double acc = 0.0;
      for (int k = 0; k < nb; ++k) acc += rhs[c0 + k] * lrow[k];
      rhs[q] -= acc;
    }
    return;
  }
  {
    double tr[BA_NB * BA_NB / 256], tq[BA_NB * BA_NB / 256];
#pragma unroll
    for (int q = 0; q < BA_NB * BA_NB / 256; ++q) {      // unconditional loads at clamped indices (see ba_chol_panel)
      const int e = q * 256 + tid, i = e / BA_NB, k = e - i * BA_NB, kk = k < nb ? k : nb - 1;
      const int ri = r0 + i < n ? r0 + i : n - 1, qi = q0 + i < n ? q0 + i : n - 1;
      tr[q] = L[(long)ri * n + c0 + kk];
      tq[q] = L[(long)qi * n + c0 + kk];
    }
#pragma unroll
    for (int q = 0; q < BA_NB * BA_NB / 256; ++q) {
      const int e = q * 256 + tid, i = e / BA_NB, k = e - i * BA_NB;
      lr[i][k] = (k < nb && r0 + i < n) ? tr[q] : 0.0;
      lq[i][k] = (k < nb && q0 + i < n) ? tq[q] : 0.0;
    }
  }
  __syncthreads();
  for (int e = tid; e < BA_NB * BA_NB; e += 256) {
    const int i = e / BA_NB, j = e - i * BA_NB;
    const int r = r0 + i, q = q0 + j;
    if (r >= n || q > r) continue;
    double acc = 0.0;
    for (int k = 0; k < nb; ++k) acc += lr[i][k] * lq[j][k];
    S[(long)r * n + q] -= acc;
  }
}

// L^T x = y, panel by panel (y = L^-1 rhs came out of the factorisation, see ba_chol_panel); the solution lives in LDS,
// the diagonal blocks pass through LDS too
__global__ __launch_bounds__(1024) void ba_chol_solve(int n, const double* __restrict__ L, double* __restrict__ dcam,
                                                      const BaState* st) {
  __shared__ double x[6 * BA_MAXN];
  __shared__ double blk[BA_NB][BA_NB + 1];
  const int tid = threadIdx.x;
  if (st->done != 0.0) return;
  if (st->chol_fail != 0.0) {
    for (int e = tid; e < n; e += 1024) dcam[e] = 0.0;
    return;
  }
  for (int e = tid; e < n; e += 1024) x[e] = dcam[e];
  __syncthreads();
  for (int c0 = ((n - 1) / BA_NB) * BA_NB; c0 >= 0; c0 -= BA_NB) {   // backward
    const int nb = min(BA_NB, n - c0);
    {
      double tb[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = q * 1024 + tid, ec = e < BA_NB * BA_NB ? e : 0, i = ec / BA_NB, j = ec - i * BA_NB;
        const int ii = i < nb ? i : nb - 1, jj = j < ii ? j : ii;
        tb[q] = L[(long)(c0 + ii) * n + c0 + jj];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int e = q * 1024 + tid, i = e / BA_NB, j = e - i * BA_NB;
        if (e < BA_NB * BA_NB) blk[i][j] = (i < nb && j <= i) ? tb[q] : 0.0;
      }
    }
    __syncthreads();
    if (tid < 64) {
      for (int r = nb - 1; r >= 0; --r) {
        double acc = (tid > r && tid < nb) ? blk[tid][r] * x[c0 + tid] : 0.0;
        acc = wave_sum_f64(acc);
        if (tid == 0) x[c0 + r] = (x[c0 + r] - acc) / blk[r][r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
    }
    __syncthreads();
    for (int r = tid; r < c0; r += 1024) {
      double acc = 0.0;
      for (int k = 0; k < nb; ++k) acc += L[(long)(c0 + k) * n + r] * x[c0 + k];
      x[r] -= acc;
    }
    __syncthreads();
  }
  for (int e = tid; e < n; e += 1024) dcam[e] = x[e];
}

// ---- pass 5: back-substitution for the points and candidate parameters; model decrease partials
__global__ __launch_bounds__(256) void ba_backsub_points(BaProblem pb, const double* __restrict__ pts,
                                                         const double* __restrict__ poses,
                                                         const double* __restrict__ Cblk, const double* __restrict__ gp,
                                                         const double* __restrict__ dcam, double* __restrict__ pts_new,
                                                         double* __restrict__ model_part, const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double model = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    double Ci[6], D[3];
    double dx[3] = {0, 0, 0};
    BaFrame fr;
    if (pb.homogeneous) ba_point_frame(X, fr);
    if (st->chol_fail == 0.0 && ba_point_inverse(Cblk + 6 * i, st->radius, Ci, D)) {
      double v[3] = {gp[3 * i], gp[3 * i + 1], gp[3 * i + 2]};   // g_i + sum_t E_it^T dc_t
      for (int t = 0; t < N; ++t) {
        const long o = ((long)s * N + t) * K + k;
        if (!pb.valid[o]) continue;
        double r[2], Jc[2][6], Jp[2][3], w;
        if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp,
                        pb.homogeneous ? fr.T : nullptr))
          continue;
        huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
        double jd0 = 0.0, jd1 = 0.0;      // Jc dc
#pragma unroll
        for (int a = 0; a < 6; ++a) { jd0 += Jc[0][a] * dcam[6 * t + a]; jd1 += Jc[1][a] * dcam[6 * t + a]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] += w * (Jp[0][c] * jd0 + Jp[1][c] * jd1);
      }
      dx[0] = -(Ci[0] * v[0] + Ci[1] * v[1] + Ci[2] * v[2]);
      dx[1] = -(Ci[1] * v[0] + Ci[3] * v[1] + Ci[4] * v[2]);
      dx[2] = -(Ci[2] * v[0] + Ci[4] * v[1] + Ci[5] * v[2]);
      // model decrease = -1/2 g^T d + 1/2 d^T D d
#pragma unroll
      for (int c = 0; c < 3; ++c) model += -0.5 * gp[3 * i + c] * dx[c] + 0.5 * D[c] * dx[c] * dx[c];
    }
    if (pb.homogeneous) {
      double Xn[3];
      ba_point_plus(X, fr, dx, Xn);
#pragma unroll
      for (int c = 0; c < 3; ++c) pts_new[3 * i + c] = Xn[c];
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) pts_new[3 * i + c] = X[c] + dx[c];
    }
  }
  const double tot = ba_block_sum<256>(model, red, tid);
  if (tid == 0) model_part[blockIdx.x] = tot;
}

__global__ void ba_update_cameras(int N, const double* __restrict__ poses, const double* __restrict__ dcam,
                                  const double* __restrict__ gc, const double* __restrict__ Dcam,
                                  double* __restrict__ poses_new, double* __restrict__ model_cam, const BaState* st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (st->done != 0.0 || t >= N) return;
  const double* d = dcam + 6 * t;
  const double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double a, b;      // exp([w]x) = I + a [w]x + b [w]x^2
  if (th < 1e-8) { a = 1.0 - th * th / 6.0; b = 0.5 - th * th / 24.0; }
  else { a = sin(th) / th; b = (1.0 - cos(th)) / (th * th); }
  const double wx = d[0], wy = d[1], wz = d[2];
  const double E[9] = {1.0 - b * (wy * wy + wz * wz), -a * wz + b * wx * wy, a * wy + b * wx * wz,
                       a * wz + b * wx * wy, 1.0 - b * (wx * wx + wz * wz), -a * wx + b * wy * wz,
                       -a * wy + b * wx * wz, a * wx + b * wy * wz, 1.0 - b * (wx * wx + wy * wy)};
  const double* R = poses + 12 * t;
  for (int p = 0; p < 3; ++p)
    for (int c = 0; c < 3; ++c)
      poses_new[12 * t + 3 * p + c] = E[3 * p] * R[c] + E[3 * p + 1] * R[3 + c] + E[3 * p + 2] * R[6 + c];
  for (int c = 0; c < 3; ++c) poses_new[12 * t + 9 + c] = poses[12 * t + 9 + c] + d[3 + c];
  double m = 0.0;
  for (int a6 = 0; a6 < 6; ++a6) m += -0.5 * gc[6 * t + a6] * d[a6] + 0.5 * Dcam[6 * t + a6] * d[a6] * d[a6];
  model_cam[t] = m;
}

// ---- pass 6: reductions in a fixed order + Ceres-style trust-region update + commit (single workgroup)
__global__ __launch_bounds__(256) void ba_decide(int nblk, int N, long n_pts3, const double* __restrict__ cost_part,
                                                 const double* __restrict__ prior_cost, const double* __restrict__ model_part,
                                                 const double* __restrict__ model_cam, int first, double* __restrict__ pts,
                                                 const double* __restrict__ pts_new, double* __restrict__ poses,
                                                 const double* __restrict__ poses_new, int max_iters, BaState* st) {
  __shared__ int commit, was_done;
  const int tid = threadIdx.x;
  // thread 0 may SET st->done further down: a wave that read it only after that store would leave before the commit
  // copy and tear the last accepted step.  One read, shared through LDS, decides for the whole workgroup.
  if (tid == 0) was_done = st->done != 0.0;
  __syncthreads();
  if (was_done) return;
  if (tid == 0) {
    double c = 0.0;
    for (int b = 0; b < nblk; ++b) c += cost_part[b];
    for (int t = 0; t < N; ++t) c += prior_cost[t];
    commit = 0;
    if (first) {            // cost at the starting point
      st->cost = c; st->initial_cost = c; st->radius = 1e4; st->decrease = 2.0; st->iters = 0.0;
      st->accepted_steps = 0.0; st->chol_fail = 0.0;
    } else {
      double model = 0.0;
      for (int b = 0; b < nblk; ++b) model += model_part[b];
      for (int t = 0; t < N; ++t) model += model_cam[t];
      st->cost_new = c; st->model_change = model;
      st->iters += 1.0;
      const double rho = (model > 0.0 && st->chol_fail == 0.0) ? (st->cost - c) / model : -1.0;
      if (rho > 1e-3 && isfinite(c)) {
        const double t3 = 2.0 * rho - 1.0;
        st->radius = fmin(st->radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3), 1e16);
        st->decrease = 2.0;
        const double rel = fabs(st->cost - c) / fmax(st->cost, 1e-300);
        st->cost = c;
        st->accepted_steps += 1.0;
        commit = 1;
        if (rel < 1e-6) st->done = 1.0;           // function tolerance
      } else {
        st->radius = st->radius / st->decrease;
        st->decrease *= 2.0;
        if (st->radius < 1e-32) st->done = 1.0;
      }
      if (st->iters >= (double)max_iters) st->done = 1.0;
    }
    st->accept = (double)commit;
  }
  __syncthreads();
  if (commit) {
    for (long e = tid; e < n_pts3; e += 256) pts[e] = pts_new[e];
    for (int e = tid; e < 12 * N; e += 256) poses[e] = poses_new[e];
  }
}

// ---- outlier tracks: SetOutlierTracksToUnestimated(tracks, max_reprojection_error_px, min_triangulation_angle_deg)
// restated from the TheiaSfM semantics: a track is unestimated if any observation is behind its camera or off by more
// than max px, or if no pair of viewing rays subtends more than the minimum angle.
__global__ __launch_bounds__(256) void ba_outlier_tracks(BaProblem pb, const double* __restrict__ pts,
                                                         const double* __restrict__ poses, double max_px,
                                                         double cos_min_angle, uint8_t* __restrict__ estimated) {
  const int N = pb.N, K = pb.K;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * K) return;
  const int s = (int)(i / K), k = (int)(i - (long)s * K);
  const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  bool ok = true;
  int nobs = 0;
  for (int t = 0; t < N && ok; ++t) {
    const long o = ((long)s * N + t) * K + k;
    if (!pb.valid[o]) continue;
    double r[2], Jc[2][6], Jp[2][3];
    if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp)) ok = false;
    else if (r[0] * r[0] + r[1] * r[1] > max_px * max_px) ok = false;
    ++nobs;
  }
  if (ok) {   // sufficient triangulation angle: some pair of rays with cos(angle) < cos(min angle)
    bool wide = false;
    for (int t = 0; t < N && !wide; ++t) {
      if (!pb.valid[((long)s * N + t) * K + k]) continue;
      double a0 = X[0] - poses[12 * t + 9], a1 = X[1] - poses[12 * t + 10], a2 = X[2] - poses[12 * t + 11];
      const double ia = 1.0 / sqrt(a0 * a0 + a1 * a1 + a2 * a2);
      a0 *= ia; a1 *= ia; a2 *= ia;
      for (int u = t + 1; u < N; ++u) {
        if (!pb.valid[((long)s * N + u) * K + k]) continue;
        double b0 = X[0] - poses[12 * u + 9], b1 = X[1] - poses[12 * u + 10], b2 = X[2] - poses[12 * u + 11];
        const double ib = 1.0 / sqrt(b0 * b0 + b1 * b1 + b2 * b2);
        if ((a0 * b0 + a1 * b1 + a2 * b2) * ib < cos_min_angle) { wide = true; break; }
      }
    }
    ok = wide;
  }
  estimated[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------- C ABI
// ---- inverse depth: kernels (thread per track unless stated)
__device__ __forceinline__ void ba_id_bearing(const BaProblem& pb, int s, int k, double b[3]) {
  const long o = ((long)s * pb.N + s) * pb.K + k;
  const double* in4 = pb.intr + 4 * s;
  b[0] = ((double)pb.uv[2 * o] - in4[2]) / in4[0];
  b[1] = ((double)pb.uv[2 * o + 1] - in4[3]) / in4[1];
  b[2] = 1.0;
}
// X = C_s + R_s^T (b / rho): rho == nullptr -> the depth of pts in the reference view (InitializeInverseDepth: snap)
__global__ __launch_bounds__(256) void ba_id_points(BaProblem pb, const double* __restrict__ poses,
                                                    const double* __restrict__ rho, const double* __restrict__ pts_in,
                                                    double* __restrict__ pts_out, const BaState* st) {
  const int N = pb.N, K = pb.K;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if ((st && st->done != 0.0) || i >= (long)N * K) return;
  const int s = (int)(i / K), k = (int)(i - (long)s * K);
  const double* P = poses + 12 * s;
  if (!pb.valid[((long)s * N + s) * K + k]) {       // no reference observation: the track is left alone
    if (pts_in != pts_out)
      for (int c = 0; c < 3; ++c) pts_out[3 * i + c] = pts_in[3 * i + c];
    return;
  }
  double b[3];
  ba_id_bearing(pb, s, k, b);
  double z;
  if (rho) z = 1.0 / rho[i];
  else z = P[6] * (pts_in[3 * i] - P[9]) + P[7] * (pts_in[3 * i + 1] - P[10]) + P[8] * (pts_in[3 * i + 2] - P[11]);
  const double p[3] = {b[0] * z, b[1] * z, z};
#pragma unroll
  for (int c = 0; c < 3; ++c) pts_out[3 * i + c] = P[9 + c] + P[c] * p[0] + P[3 + c] * p[1] + P[6 + c] * p[2];
}

// per track: c = sum w jr.jr, g = sum w jr.r, cost, the E row (observing cameras + the reference camera), ok flags
__global__ __launch_bounds__(256) void ba_id_linearize(BaProblem pb, const double* __restrict__ pts,
                                                       const double* __restrict__ poses, double* __restrict__ Cblk,
                                                       double* __restrict__ gp, double* __restrict__ Eblk,
                                                       uint8_t* __restrict__ ok, double* __restrict__ cost_part,
                                                       const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double cost = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const bool has_ref = pb.valid[((long)s * N + s) * K + k] != 0;
    double c = 0.0, g = 0.0, Ea[6] = {0, 0, 0, 0, 0, 0};
    bool any = false;
    for (int t = 0; t < N; ++t) {
      const long o = ((long)s * N + t) * K + k, oT = i * N + t;
      uint8_t good = 0;
      if (t != s && has_ref && pb.valid[o]) {
        double r[2], Jc[2][6], Jp[2][3], Ja[2][6], jr[2];
        if (ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp)) {
          ba_id_chain(poses + 12 * s, X, Jp, Ja, jr);
          double w;
          cost += huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
          c += w * (jr[0] * jr[0] + jr[1] * jr[1]);
          g += w * (jr[0] * r[0] + jr[1] * r[1]);
          double* E = Eblk + 18 * oT;
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            E[3 * a] = w * (Jc[0][a] * jr[0] + Jc[1][a] * jr[1]);
            E[3 * a + 1] = 0.0;
            E[3 * a + 2] = 0.0;
            Ea[a] += w * (Ja[0][a] * jr[0] + Ja[1][a] * jr[1]);
          }
          good = 1;
          any = true;
        }
      }
      if (t != s) ok[oT] = good;
    }
    {
      double* E = Eblk + 18 * (i * N + s);
#pragma unroll
      for (int a = 0; a < 6; ++a) { E[3 * a] = Ea[a]; E[3 * a + 1] = 0.0; E[3 * a + 2] = 0.0; }
      ok[i * N + s] = any ? 1 : 0;
    }
    Cblk[6 * i] = c; Cblk[6 * i + 1] = 0.0; Cblk[6 * i + 2] = 0.0; Cblk[6 * i + 3] = 1.0; Cblk[6 * i + 4] = 0.0; Cblk[6 * i + 5] = 1.0;
    gp[3 * i] = g; gp[3 * i + 1] = 0.0; gp[3 * i + 2] = 0.0;
  }
  const double tot = ba_block_sum<256>(cost, red, tid);
  if (tid == 0) cost_part[blockIdx.x] = 0.5 * tot;
}

// cost of candidate / start parameters without the reference-view observations (they are met by construction, but only
// up to rounding: leaving them out keeps the cost exactly the oracle's sum)
__global__ __launch_bounds__(256) void ba_id_cost(BaProblem pb, const double* __restrict__ pts,
                                                  const double* __restrict__ poses, double* __restrict__ cost_part,
                                                  const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double cost = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K), k = (int)(i - (long)s * K);
    const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (pb.valid[((long)s * N + s) * K + k])
      for (int t = 0; t < N; ++t) {
        const long o = ((long)s * N + t) * K + k;
        if (t == s || !pb.valid[o]) continue;
        double r[2], Jc[2][6], Jp[2][3];
        if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp))
          continue;
        double w;
        cost += huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
      }
  }
  const double tot = ba_block_sum<256>(cost, red, tid);
  if (tid == 0) cost_part[blockIdx.x] = 0.5 * tot;
}

// direct camera-camera blocks: X[s][t] = sum_k w Ja(s, k, t)^T Jc(s, k, t)  (6 x 6; rows: reference camera s).  One wave per (s, t).
__global__ __launch_bounds__(64) void ba_id_cross_blocks(BaProblem pb, const double* __restrict__ pts,
                                                         const double* __restrict__ poses, double* __restrict__ Xblk,
                                                         const BaState* st) {
  const int N = pb.N, K = pb.K, s = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
  if (st->done != 0.0) return;
  double a[36];
#pragma unroll
  for (int c = 0; c < 36; ++c) a[c] = 0.0;
  if (s != t)
    for (int k = lane; k < K; k += 64) {
      const long i = (long)s * K + k, o = ((long)s * N + t) * K + k;
      if (!pb.valid[o] || !pb.valid[((long)s * N + s) * K + k]) continue;
      const double X[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
      double r[2], Jc[2][6], Jp[2][3], Ja[2][6], jr[2];
      if (!ba_project(poses + 12 * t, pb.intr + 4 * t, X, (double)pb.uv[2 * o], (double)pb.uv[2 * o + 1], r, Jc, Jp)) continue;
      ba_id_chain(poses + 12 * s, X, Jp, Ja, jr);
      double w;
      huber_rho(r[0] * r[0] + r[1] * r[1], pb.huber, w);
#pragma unroll
      for (int p = 0; p < 6; ++p)
#pragma unroll
        for (int c = 0; c < 6; ++c) a[6 * p + c] += w * (Ja[0][p] * Jc[0][c] + Ja[1][p] * Jc[1][c]);
    }
#pragma unroll
  for (int c = 0; c < 36; ++c) {
    const double v = wave_sum_f64(a[c]);
    if (lane == 0) Xblk[((long)s * N + t) * 36 + c] = v;
  }
}
// S (lower triangle) += the cross blocks: H[tr][tc] = X[tr][tc] + X[tc][tr]^T
__global__ __launch_bounds__(256) void ba_id_add_cross(int N, const double* __restrict__ Xblk, double* __restrict__ S,
                                                       const BaState* st) {
  if (st->done != 0.0) return;
  const int n = 6 * N;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)n * n) return;
  const int r = (int)(e / n), c = (int)(e - (long)r * n);
  const int tr = r / 6, tc = c / 6;
  if (tr <= tc) return;
  const int a = r - 6 * tr, b = c - 6 * tc;
  S[e] += Xblk[((long)tr * N + tc) * 36 + 6 * a + b] + Xblk[((long)tc * N + tr) * 36 + 6 * b + a];
}

// d rho = -(g + sum_t E_it . dc_t) / (c + D): candidate inverse depths and the model decrease
__global__ __launch_bounds__(256) void ba_id_backsub(BaProblem pb, const double* __restrict__ pts,
                                                     const double* __restrict__ poses, const double* __restrict__ Cblk,
                                                     const double* __restrict__ gp, const double* __restrict__ Eblk,
                                                     const uint8_t* __restrict__ ok, const double* __restrict__ dcam,
                                                     double* __restrict__ rho_new, double* __restrict__ model_part,
                                                     const BaState* st) {
  __shared__ double red[4];
  const int N = pb.N, K = pb.K, tid = threadIdx.x;
  const long i = (long)blockIdx.x * 256 + tid;
  double model = 0.0;
  if (st->done == 0.0 && i < (long)N * K) {
    const int s = (int)(i / K);
    const double* P = poses + 12 * s;
    const double z = P[6] * (pts[3 * i] - P[9]) + P[7] * (pts[3 * i + 1] - P[10]) + P[8] * (pts[3 * i + 2] - P[11]);
    double rho = 1.0 / z, Ci[6], D[3], dr = 0.0;
    if (st->chol_fail == 0.0 && ba_point_inverse(Cblk + 6 * i, st->radius, Ci, D)) {
      double v = gp[3 * i];
      for (int t = 0; t < N; ++t) {
        if (!ok[i * N + t]) continue;
        const double* E = Eblk + 18 * (i * N + t);
#pragma unroll
        for (int a = 0; a < 6; ++a) v += E[3 * a] * dcam[6 * t + a];
      }
      dr = -Ci[0] * v;
      model = -0.5 * gp[3 * i] * dr + 0.5 * D[0] * dr * dr;
    }
    rho_new[i] = rho + dr;
  }
  const double tot = ba_block_sum<256>(model, red, tid);
  if (tid == 0) model_part[blockIdx.x] = tot;
}

static inline int ba_nblk(int N, int K) { return (int)(((long)N * K + 255) / 256); }

extern "C" long pi3_ba_workspace_doubles(int N, int K) {
  const long n6 = 6L * N, nk = (long)N * K;
  return 16 /*state*/ + 6 * nk + 3 * nk + 3 * nk /*pts_new*/ + 12L * N /*poses_new*/ + 21L * N + 6L * N /*gc*/ +
         6L * N /*dcam*/ + 6L * N /*Dcam*/ + BA_SLICES * n6 * n6 + BA_SLICES * n6 + 2 * n6 * n6 + 18 * nk * N + (nk * N + 7) / 8 + 3L * ba_nblk(N, K) +
         3L * N + 64 + nk /*inverse depth: rho_new*/ + 36L * N * N /*cross blocks*/ + 6 * nk /*damped point inverses*/ + 18 * nk * N /*Y = E Cinv*/ + 256 /*tile reads past the last track of Y*/ + 144 /*the same past Eblk*/;
}

static int ba_run(double* points, double* poses, const double* intr, const float* uv, const unsigned char* valid,
                  const float* uvT, const unsigned char* validT, int N, int K, double huber_width, int max_iters,
                  const double* prior_R, const double* prior_C, const unsigned char* prior_flag, double sqrt_info_rot,
                  double sqrt_info_pos, int mode /* 0 Euclidean, 1 homogeneous, 2 inverse depth */, double* summary_dev,
                  double* workspace, long workspace_doubles, void* stream) {
  if (!points || !poses || !intr || !uv || !valid || !uvT || !validT || !summary_dev || !workspace || N <= 0 ||
      N > BA_MAXN || K <= 0 || max_iters < 0 || !(huber_width > 0.0) ||
      ((prior_flag != nullptr) && (!prior_R || !prior_C))) {
    pi3_set_error("pi3_bundle_adjust: bad arguments N=%d (<= %d) K=%d", N, BA_MAXN, K);
    return PI3_ERR_ARG;
  }
  if (workspace_doubles < pi3_ba_workspace_doubles(N, K)) {
    pi3_set_error("pi3_bundle_adjust: workspace of %ld doubles, need %ld", workspace_doubles, pi3_ba_workspace_doubles(N, K));
    return PI3_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const long n6 = 6L * N, nk = (long)N * K;
  const int nblk = ba_nblk(N, K);
  double* w = workspace;
  BaState* state = (BaState*)w; w += 16;
  double* Cblk = w; w += 6 * nk;
  double* gp = w; w += 3 * nk;
  double* pts_new = w; w += 3 * nk;
  double* poses_new = w; w += 12L * N;
  double* Bblk = w; w += 21L * N;
  double* gc = w; w += 6L * N;
  double* dcam = w; w += 6L * N;
  double* Dcam = w; w += 6L * N;
  double* S_part = w; w += BA_SLICES * n6 * n6;
  double* rhs_part = w; w += BA_SLICES * n6;
  double* S = w; w += n6 * n6;
  double* Lfac = w; w += n6 * n6;
  double* Eblk = w; w += 18 * nk * N + 144;   // ba_schur_tiles prefetches up to 126 doubles past the last track's run
  uint8_t* obs_ok = (uint8_t*)w; w += (nk * N + 7) / 8;
  double* cost_part = w; w += nblk;
  double* model_part = w; w += nblk;
  double* cost_part2 = w; w += nblk;
  double* prior_cost = w; w += N;
  double* model_cam = w; w += N;
  double* prior_cost2 = w; w += N;
  w += 64;      // (the spare doubles of the size formula)
  double* rho_new = w; w += nk;
  double* Xblk = w; w += 36L * N * N;
  double* Cinv = w; w += 6 * nk;
  double* Yblk = w; w += 18 * nk * N;
  const bool invd = mode == 2;
  const bool schur_rows = (int)PI3_KNOB("ba_schur_rows", 0) != 0;     // 1: the round-3 row kernel (A/B)
  // track slices (partial sums added in slice order by ba_assemble_cameras).  The tile kernel runs two single-wave
  // workgroups per SIMD (216 registers): the slice count that puts every (tile, slice) wave on the chip at once, in one
  // round - 22 for 100 cameras (91 tiles) instead of 32, which needed a second, half-empty round
  const int nb8 = (N + 7) / 8, ntile = nb8 * (nb8 + 1) / 2;
  int nsl = BA_SLICES;
  if (!schur_rows) {
    nsl = 2048 / ntile;
    nsl = nsl < 1 ? 1 : (nsl > BA_SLICES ? BA_SLICES : nsl);
    nsl = nsl > N ? N : nsl;
  }
  // the tile kernel writes the lower-triangle tiles only; the rest of S_part is read by ba_assemble_cameras and never
  // used by the factorisation: defined once
  if (!schur_rows && hipMemsetAsync(S_part, 0, sizeof(double) * nsl * n6 * n6, st) != hipSuccess) {   // only nsl are read
    pi3_set_error("pi3_bundle_adjust: hipMemsetAsync failed");
    return PI3_ERR_LAUNCH;
  }
  if (hipMemsetAsync(state, 0, 16 * sizeof(double), st) != hipSuccess) {
    pi3_set_error("pi3_bundle_adjust: hipMemsetAsync failed");
    return PI3_ERR_LAUNCH;
  }
  BaProblem pb;
  pb.uv = uv; pb.valid = valid; pb.uvT = uvT; pb.validT = validT; pb.intr = intr;
  pb.prior_R = prior_R; pb.prior_C = prior_C; pb.prior_flag = prior_flag;
  pb.sqrt_info_rot = sqrt_info_rot; pb.sqrt_info_pos = sqrt_info_pos; pb.huber = huber_width; pb.N = N; pb.K = K;
  pb.homogeneous = mode == 1 ? 1 : 0;
  pb.invdepth = invd ? 1 : 0;
  if (invd)   // InitializeInverseDepth: every track onto the ray of its reference keypoint, at its current depth there
    hipLaunchKernelGGL(ba_id_points, dim3(nblk), dim3(256), 0, st, pb, poses, (const double*)nullptr, points, points,
                       (const BaState*)nullptr);
  // cost at the start
  if (invd) hipLaunchKernelGGL(ba_id_cost, dim3(nblk), dim3(256), 0, st, pb, points, poses, cost_part, state);
  else hipLaunchKernelGGL(ba_cost_points, dim3(nblk), dim3(256), 0, st, pb, points, poses, cost_part, state);
  hipLaunchKernelGGL(ba_prior_cost, dim3((N + 63) / 64), dim3(64), 0, st, pb, poses, prior_cost, state);
  hipLaunchKernelGGL(ba_decide, dim3(1), dim3(256), 0, st, nblk, N, 3 * nk, cost_part, prior_cost, model_part, model_cam, 1,
                     points, pts_new, poses, poses_new, max_iters, state);
  for (int it = 0; it < max_iters; ++it) {
    if (invd) {
      hipLaunchKernelGGL(ba_id_linearize, dim3(nblk), dim3(256), 0, st, pb, points, poses, Cblk, gp, Eblk, obs_ok, cost_part,
                         state);
      hipLaunchKernelGGL(ba_id_cross_blocks, dim3(N, N), dim3(64), 0, st, pb, points, poses, Xblk, state);
    } else {
      hipLaunchKernelGGL(ba_linearize_points, dim3(nblk), dim3(256), 0, st, pb, points, poses, Cblk, gp, cost_part, state);
      hipLaunchKernelGGL(ba_obs_blocks, dim3((unsigned)((nk * N + 255) / 256)), dim3(256), 0, st, pb, points, poses, Eblk,
                         obs_ok, state);
    }
    hipLaunchKernelGGL(ba_camera_blocks, dim3(N), dim3(256), 0, st, pb, points, poses, Bblk, gc, prior_cost, state);
    if (schur_rows) {
      hipLaunchKernelGGL(ba_schur_rows, dim3(N, BA_SLICES), dim3(BA_MAXN), 0, st, pb, Eblk, obs_ok, Cblk, gp, S_part,
                         rhs_part, state);
    } else {
      hipLaunchKernelGGL(ba_point_inverses, dim3(nblk), dim3(256), 0, st, nk, Cblk, Cinv, state);
      hipLaunchKernelGGL(ba_y_blocks, dim3((unsigned)((6 * nk * N + 255) / 256)), dim3(256), 0, st, nk * N, N, Eblk, obs_ok,
                         Cinv, Yblk, state);
      hipLaunchKernelGGL(ba_schur_tiles, dim3(ntile, nsl), dim3(64), 0, st, pb, Eblk, Yblk, gp, S_part, rhs_part, nsl, state);
    }
    const int n = (int)n6;
    hipLaunchKernelGGL(ba_assemble_cameras, dim3((unsigned)((n6 * n6 + 255) / 256)), dim3(256), 0, st, N, S_part, rhs_part,
                       Bblk, gc, S, dcam, Dcam, nsl, state);
    if (invd)
      hipLaunchKernelGGL(ba_id_add_cross, dim3((unsigned)((n6 * n6 + 255) / 256)), dim3(256), 0, st, N, Xblk, S, state);
    for (int c0 = 0; c0 < n; c0 += BA_NB) {
      const int nb = n - c0 < BA_NB ? n - c0 : BA_NB;
      hipLaunchKernelGGL(ba_chol_diag, dim3(1), dim3(64), 0, st, n, c0, nb, S, Lfac, state);
      const int below = n - c0 - nb;      // (+ 1: the right-hand side as row n)
      hipLaunchKernelGGL(ba_chol_panel, dim3((below + 1 + 63) / 64), dim3(64), 0, st, n, c0, nb, S, (const double*)Lfac, Lfac,
                         dcam, state);
      if (below > 0) {
        const int nt = (below + BA_NB - 1) / BA_NB;
        hipLaunchKernelGGL(ba_chol_update, dim3(nt, nt + 1), dim3(256), 0, st, n, c0, nb, S, Lfac, dcam, state);
      }
    }
    hipLaunchKernelGGL(ba_chol_solve, dim3(1), dim3(1024), 0, st, n, Lfac, dcam, state);
    if (invd)
      hipLaunchKernelGGL(ba_id_backsub, dim3(nblk), dim3(256), 0, st, pb, points, poses, Cblk, gp, Eblk, obs_ok, dcam, rho_new,
                         model_part, state);
    else
      hipLaunchKernelGGL(ba_backsub_points, dim3(nblk), dim3(256), 0, st, pb, points, poses, Cblk, gp, dcam, pts_new,
                         model_part, state);
    hipLaunchKernelGGL(ba_update_cameras, dim3((N + 63) / 64), dim3(64), 0, st, N, poses, dcam, gc, Dcam, poses_new,
                       model_cam, state);
    if (invd) {      // candidate points from the candidate reference poses and inverse depths
      hipLaunchKernelGGL(ba_id_points, dim3(nblk), dim3(256), 0, st, pb, poses_new, rho_new, points, pts_new, state);
      hipLaunchKernelGGL(ba_id_cost, dim3(nblk), dim3(256), 0, st, pb, pts_new, poses_new, cost_part2, state);
    } else {
      hipLaunchKernelGGL(ba_cost_points, dim3(nblk), dim3(256), 0, st, pb, pts_new, poses_new, cost_part2, state);
    }
    hipLaunchKernelGGL(ba_prior_cost, dim3((N + 63) / 64), dim3(64), 0, st, pb, poses_new, prior_cost2, state);
    hipLaunchKernelGGL(ba_decide, dim3(1), dim3(256), 0, st, nblk, N, 3 * nk, cost_part2, prior_cost2, model_part, model_cam,
                       0, points, pts_new, poses, poses_new, max_iters, state);
  }
  // summary: initial cost, final cost, iterations, accepted steps, final radius
  if (hipMemcpyAsync(summary_dev, state, sizeof(BaState), hipMemcpyDeviceToDevice, st) != hipSuccess) {
    pi3_set_error("pi3_bundle_adjust: summary copy failed");
    return PI3_ERR_LAUNCH;
  }
  return pi3_check_launch("bundle_adjust");
}

extern "C" int pi3_bundle_adjust(double* points, double* poses, const double* intr, const float* uv,
                                 const unsigned char* valid, const float* uvT, const unsigned char* validT, int N, int K,
                                 double huber_width, int max_iters, const double* prior_R, const double* prior_C,
                                 const unsigned char* prior_flag, double sqrt_info_rot, double sqrt_info_pos,
                                 double* summary_dev, double* workspace, long workspace_doubles, void* stream) {
  return ba_run(points, poses, intr, uv, valid, uvT, validT, N, K, huber_width, max_iters, prior_R, prior_C, prior_flag,
                sqrt_info_rot, sqrt_info_pos, 0, summary_dev, workspace, workspace_doubles, stream);
}

// The same adjustment with Theia's default point parametrization (see ba_point_frame): what the reference's calls use.
extern "C" int pi3_bundle_adjust_homogeneous(double* points, double* poses, const double* intr, const float* uv,
                                             const unsigned char* valid, const float* uvT, const unsigned char* validT,
                                             int N, int K, double huber_width, int max_iters, const double* prior_R,
                                             const double* prior_C, const unsigned char* prior_flag, double sqrt_info_rot,
                                             double sqrt_info_pos, double* summary_dev, double* workspace,
                                             long workspace_doubles, void* stream) {
  return ba_run(points, poses, intr, uv, valid, uvT, validT, N, K, huber_width, max_iters, prior_R, prior_C, prior_flag,
                sqrt_info_rot, sqrt_info_pos, 1, summary_dev, workspace, workspace_doubles, stream);
}

// The reference's --use-inverse-depth (see "inverse depth" above): one inverse depth per track along the bearing of its
// keypoint in its own frame.  `points` are snapped onto those rays first and stay Euclidean for the caller.
extern "C" int pi3_bundle_adjust_inverse_depth(double* points, double* poses, const double* intr, const float* uv,
                                               const unsigned char* valid, const float* uvT, const unsigned char* validT,
                                               int N, int K, double huber_width, int max_iters, const double* prior_R,
                                               const double* prior_C, const unsigned char* prior_flag,
                                               double sqrt_info_rot, double sqrt_info_pos, double* summary_dev,
                                               double* workspace, long workspace_doubles, void* stream) {
  return ba_run(points, poses, intr, uv, valid, uvT, validT, N, K, huber_width, max_iters, prior_R, prior_C, prior_flag,
                sqrt_info_rot, sqrt_info_pos, 2, summary_dev, workspace, workspace_doubles, stream);
}

extern "C" int pi3_ba_outlier_tracks(const double* points, const double* poses, const double* intr, const float* uv,
                                     const unsigned char* valid, int N, int K, double max_reprojection_px,
                                     double min_triangulation_angle_deg, unsigned char* estimated, void* stream) {
  if (!points || !poses || !intr || !uv || !valid || !estimated || N <= 0 || K <= 0) {
    pi3_set_error("pi3_ba_outlier_tracks: bad arguments");
    return PI3_ERR_ARG;
  }
  BaProblem pb;
  pb.uv = uv; pb.valid = valid; pb.uvT = nullptr; pb.validT = nullptr; pb.intr = intr;
  pb.prior_R = nullptr; pb.prior_C = nullptr; pb.prior_flag = nullptr;
  pb.sqrt_info_rot = 0; pb.sqrt_info_pos = 0; pb.huber = 1.0; pb.N = N; pb.K = K; pb.homogeneous = 0; pb.invdepth = 0;
  hipLaunchKernelGGL(ba_outlier_tracks, dim3(ba_nblk(N, K)), dim3(256), 0, (hipStream_t)stream, pb, points, poses,
                     max_reprojection_px, cos(min_triangulation_angle_deg * 3.14159265358979323846 / 180.0), estimated);
  return pi3_check_launch("ba_outlier_tracks");
}
