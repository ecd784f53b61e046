// Overlap-based Sim(3) chunk alignment: steps 1-3 of align_and_refine_reconstructions
// (utils/reconstruction_alignment.py:74-105) as wavefront-reduction kernels, fp64 accumulation.
//
//   1. common tracks      FindCommonTracksByFeatureInReconstructions (:74): a track pair = the same keypoint pixel in
//                         the same image seen from both chunks' overlap views (view pairs from
//                         create_view_graph_matches :16-37)  ->  pi3_sim3_match_keypoints (integer, bit-exact)
//   2. near-half filter   keep pairs whose ref point is closer to the LAST ref camera than the median distance,
//                         strict '<', np.median semantics (:78-86)
//   3. closed form Sim(3) OptimizeAlignmentSim3(points_qry, points_ref, perform_optimization=False) (:89-97): the
//                         Umeyama (1991) similarity qry -> ref;  Sim3d.exp(params).matrix() (:104) is its 4x4
//   4. apply              TransformReconstruction4 (:105): X' = s R X + t, camera centre likewise, R_cw' = R R_cw
// The arithmetic of steps 1, 3, 4 lives in pytheia 0.2.9 (C++, not vendored, not installable offline): it is restated
// from the published algorithm and the reference's call sites (parity unpinned for this stage).
// Inputs are exactly what a chunk file stores: fp16 keypoints / points (offline_chunk_creator.py:231-241), fp32 poses.
#include "common.h"
#include "rot3.h"
#include <hip/hip_fp16.h>

// ---- 1. keypoint matching: for overlap pair v and qry keypoint j, index of the ref keypoint with identical (x, y)
// bits in ref overlap frame v, or -1.  kp_* : [ov][K][2] fp16.
__global__ __launch_bounds__(256) void match_keypoints_kernel(const uint32_t* __restrict__ kp_ref,
                                                              const uint32_t* __restrict__ kp_qry, int ov, int K,
                                                              int* __restrict__ idx) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ov * K) return;
  const int v = i / K;
  const uint32_t q = kp_qry[i];
  // the FIRST reference keypoint with these bits.  (Rounds 1-3 took kp_ref[i] == q as a shortcut for identical grids in
  // identical order: with two keypoints on one pixel inside a view - possible for detector keypoints, 23 % of random
  // 257-keypoint fp16 sets, found by tests/test_fuzz_gpu.py - the later one then paired with itself instead of the first.)
  int found = -1;
  for (int j = 0; j < K; ++j)
    if (kp_ref[v * K + j] == q) { found = j; break; }
  idx[i] = found;
}

extern "C" int pi3_sim3_match_keypoints(const void* kp_ref, const void* kp_qry, int ov, int K, int* idx,
                                        void* stream) {
  if (!kp_ref || !kp_qry || !idx || ov <= 0 || K <= 0) {
    pi3_set_error("pi3_sim3_match_keypoints: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(match_keypoints_kernel, dim3((ov * K + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     (const uint32_t*)kp_ref, (const uint32_t*)kp_qry, ov, K, idx);
  return pi3_check_launch("sim3_match_keypoints");
}

// ---- 2 + 3. filter + Umeyama, one 1024-thread workgroup (M = ov*K <= ~8 k pairs: latency-bound by design)
__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < 16; ++w) s += red[w];
  return s;
}

struct PairSrc {
  const void* pts_ref;    // [ov][K][3] f16 (chunk files) or f32 (bundle-adjusted chunks)
  const void* pts_qry;    // [ov][K][3]
  int f32;
  const int* idx;         // [ov][K] ref keypoint index per qry keypoint or -1
  const uint8_t* w_ref;   // optional validity [ov][K] (e.g. masks) or null
  const uint8_t* w_qry;
  const float* wf_ref;    // optional real-valued weights [ov][K] (e.g. mask * sigmoid(conf)) or null
  const float* wf_qry;
  int ov, K;
};

// One pair (qry keypoint i, its ref track): false when it does not take part (no common track, validity 0, weight
// <= 0 or not finite).  w = the pair's weight: 1 unless real-valued weights were given (then w_ref[r] * w_qry[i]).
__device__ __forceinline__ bool pair_get(const PairSrc& s, int i, double x[3], double y[3], double& w) {
  const int j = s.idx[i];
  if (j < 0) return false;
  const int v = i / s.K;
  const int r = v * s.K + j;
  if (s.w_qry && !s.w_qry[i]) return false;
  if (s.w_ref && !s.w_ref[r]) return false;
  w = 1.0;
  if (s.wf_qry) w *= (double)s.wf_qry[i];
  if (s.wf_ref) w *= (double)s.wf_ref[r];
  if (!(w > 0.0) || !(w < INFINITY)) return false;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (s.f32) {
      x[c] = (double)((const float*)s.pts_qry)[3 * i + c];
      y[c] = (double)((const float*)s.pts_ref)[3 * r + c];
    } else {
      x[c] = (double)__half2float(((const __half*)s.pts_qry)[3 * i + c]);
      y[c] = (double)__half2float(((const __half*)s.pts_ref)[3 * r + c]);
    }
  }
  return true;
}

// Distance of a ref point from the last ref camera, with the reference's rounding points: np.linalg.norm(x, axis=1)
// (reconstruction_alignment.py:80,83) is sqrt(add.reduce(x * x, axis=1)) - three rounded squares, then the sequential
// sum (x0^2 + x1^2) + x2^2, no fused multiply-add.  The library is built with -ffp-contract=fast, which turned this into
// fma(dz, dz, fma(dy, dy, dx * dx)): distances, and with them the median, differed from numpy's in the last bit on
// about one pair in a few hundred (found by the shipping-size tests, round 4: 4 of 32 medians off by one ulp) - and a
// strict '<' against the median can then keep or drop a different pair.
__device__ __forceinline__ double ref_cam_dist(const double y[3], const double cam[3]) {
  const double dx = y[0] - cam[0], dy = y[1] - cam[1], dz = y[2] - cam[2];
  // HIP's __dmul_rn / __dadd_rn are plain operators and get contracted again, and so does a local `#pragma clang fp
  // contract(off)` once inlined into a contract=fast caller (checked in the ISA): the products pass through an empty asm,
  // which the optimiser cannot look through, so each is rounded on its own
  double a = dx * dx, b = dy * dy, c = dz * dz;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c));
  double s = a + b;
  asm volatile("" : "+v"(s));
  return sqrt(s + c);
}

// k-th smallest (0-based) of the non-negative doubles dist(i), by 8-pass radix select on their bit patterns
__device__ double select_rank(const PairSrc& s, const double cam[3], int M, unsigned k, unsigned* hist,
                              unsigned long long* s_prefix, unsigned* s_k, int tid) {
  if (tid == 0) { *s_prefix = 0ull; *s_k = k; }
  unsigned long long pmask = 0ull;
  for (int pass = 7; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = *s_prefix;
    for (int i = tid; i < M; i += 1024) {
      double x[3], y[3], w;
      if (!pair_get(s, i, x, y, w)) continue;
      const unsigned long long key = (unsigned long long)__double_as_longlong(ref_cam_dist(y, cam));
      if ((key & pmask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255ull], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned kk = *s_k, cum = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > kk) break;
        cum += hist[b];
      }
      *s_k = kk - cum;
      *s_prefix = prefix | ((unsigned long long)b << (8 * pass));
    }
    pmask |= 0xffull << (8 * pass);
    __syncthreads();
  }
  return __longlong_as_double((long long)*s_prefix);
}

// out (doubles): [0] s, [1..9] R row-major, [10..12] t, [13..28] 4x4 row-major, [29] pairs used, [30] common pairs,
//                [31] median distance, [32] (weighted) rms alignment error over the used pairs
// With weights w_i: W = sum w, mx = sum w x / W, my = sum w y / W,
// Sigma = sum w (y - my)(x - mx)^T / W, var_x = sum w |x - mx|^2 / W; the near-half filter stays the reference's
// (unweighted median over the pairs that take part, strict '<').  Unweighted: w = 1 and every expression below is the
// plain one bit for bit (x * 1.0 is exact).
__global__ __launch_bounds__(1024) void sim3_umeyama_kernel(PairSrc s, const float* __restrict__ last_ref_pose,
                                                            int use_filter, double* __restrict__ out) {
  __shared__ double red[16];
  __shared__ unsigned hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ unsigned s_k;
  __shared__ double sh[16];
  const int tid = threadIdx.x;
  const int M = s.ov * s.K;
  const double cam[3] = {(double)last_ref_pose[3], (double)last_ref_pose[7], (double)last_ref_pose[11]};

  double cnt = 0.0;
  for (int i = tid; i < M; i += 1024) {
    double x[3], y[3], w;
    if (pair_get(s, i, x, y, w)) cnt += 1.0;
  }
  const double ncommon = block_sum(cnt, red, tid);
  double med = INFINITY;
  if (use_filter && ncommon >= 1.0) {
    const unsigned n = (unsigned)ncommon;
    const double lo = select_rank(s, cam, M, (n - 1) / 2, hist, &s_prefix, &s_k, tid);
    const double hi = (n & 1) ? lo : select_rank(s, cam, M, n / 2, hist, &s_prefix, &s_k, tid);
    med = 0.5 * (lo + hi);  // np.median
  }

  // pass A: count, weight sum + weighted means over kept pairs
  double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < M; i += 1024) {
    double x[3], y[3], w;
    if (!pair_get(s, i, x, y, w)) continue;
    if (use_filter && !(ref_cam_dist(y, cam) < med)) continue;
    a[0] += 1.0;
    a[7] += w;
    for (int c = 0; c < 3; ++c) { a[1 + c] += w * x[c]; a[4 + c] += w * y[c]; }
  }
  for (int c = 0; c < 8; ++c) {
    const double v = block_sum(a[c], red, tid);
    if (tid == 0) sh[c] = v;
  }
  __syncthreads();
  const double npairs = sh[0];
  const double n = sh[7];          // W = sum of weights (= the pair count when unweighted)
  if (npairs < 3.0) {  // not enough pairs: report failure through the count, identity transform
    if (tid == 0) {
      for (int i = 0; i < 33; ++i) out[i] = 0.0;
      out[0] = 1.0; out[1] = out[5] = out[9] = 1.0;
      out[13] = out[18] = out[23] = out[28] = 1.0;
      out[29] = npairs; out[30] = ncommon; out[31] = med;
    }
    return;
  }
  const double mx[3] = {sh[1] / n, sh[2] / n, sh[3] / n}, my[3] = {sh[4] / n, sh[5] / n, sh[6] / n};
  __syncthreads();
  // pass B: centred second moments  Sigma = 1/n sum (y - my)(x - mx)^T,  var_x = 1/n sum |x - mx|^2
  double cm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < M; i += 1024) {
    double x[3], y[3], w;
    if (!pair_get(s, i, x, y, w)) continue;
    if (use_filter && !(ref_cam_dist(y, cam) < med)) continue;
    const double xc[3] = {x[0] - mx[0], x[1] - mx[1], x[2] - mx[2]};
    const double yc[3] = {y[0] - my[0], y[1] - my[1], y[2] - my[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) cm[3 * r + c] += w * yc[r] * xc[c];
    cm[9] += w * (xc[0] * xc[0] + xc[1] * xc[1] + xc[2] * xc[2]);
  }
  for (int c = 0; c < 10; ++c) {
    const double v = block_sum(cm[c], red, tid);
    if (tid == 0) sh[c] = v / n;
  }
  __syncthreads();
  if (tid == 0) {
    double Sg[9], R[9];
    for (int c = 0; c < 9; ++c) Sg[c] = sh[c];
    const double varx = sh[9];
    // R = U diag(1,1,det(UV^T)) V^T of Sigma = U D V^T  ==  argmax_R tr(R^T Sigma)  (rot3.h)
    nearest_rotation_d(Sg, R);
    double trc = 0.0;  // tr(D S) = tr(R^T Sigma)
    for (int c = 0; c < 9; ++c) trc += R[c] * Sg[c];
    const double sc = varx > 0.0 ? trc / varx : 1.0;
    double t[3];
    for (int r = 0; r < 3; ++r)
      t[r] = my[r] - sc * (R[3 * r] * mx[0] + R[3 * r + 1] * mx[1] + R[3 * r + 2] * mx[2]);
    out[0] = sc;
    for (int c = 0; c < 9; ++c) out[1 + c] = R[c];
    for (int c = 0; c < 3; ++c) out[10 + c] = t[c];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) out[13 + 4 * r + c] = sc * R[3 * r + c];
      out[13 + 4 * r + 3] = t[r];
    }
    out[25] = 0.0; out[26] = 0.0; out[27] = 0.0; out[28] = 1.0;
    out[29] = npairs; out[30] = ncommon; out[31] = med;
    sh[10] = sc;
    for (int c = 0; c < 9; ++c) sh[c] = R[c];  // reuse for the residual pass
    sh[11] = t[0]; sh[12] = t[1]; sh[13] = t[2];
  }
  __syncthreads();
  double err = 0.0;
  for (int i = tid; i < M; i += 1024) {
    double x[3], y[3], w;
    if (!pair_get(s, i, x, y, w)) continue;
    if (use_filter && !(ref_cam_dist(y, cam) < med)) continue;
    double e2 = 0.0;
    for (int r = 0; r < 3; ++r) {
      const double p = sh[10] * (sh[3 * r] * x[0] + sh[3 * r + 1] * x[1] + sh[3 * r + 2] * x[2]) + sh[11 + r] - y[r];
      e2 += p * p;
    }
    err += w * e2;
  }
  const double tot = block_sum(err, red, tid);
  if (tid == 0) out[32] = sqrt(tot / n);
}

static int launch_sim3(const char* who, const void* pts_ref, const void* pts_qry, const int* idx,
                       const unsigned char* w_ref, const unsigned char* w_qry, const float* wf_ref, const float* wf_qry,
                       int ov, int K, const float* last_ref_pose, int use_filter, double* out33, void* stream) {
  if (!pts_ref || !pts_qry || !idx || !last_ref_pose || !out33 || ov <= 0 || K <= 0 || (use_filter & ~3)) {
    pi3_set_error(who);
    return PI3_ERR_ARG;
  }
  PairSrc s;
  s.pts_ref = pts_ref; s.pts_qry = pts_qry; s.idx = idx;
  s.f32 = (use_filter & 2) ? 1 : 0;
  use_filter &= 1;
  s.w_ref = w_ref; s.w_qry = w_qry; s.wf_ref = wf_ref; s.wf_qry = wf_qry; s.ov = ov; s.K = K;
  hipLaunchKernelGGL(sim3_umeyama_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, s, last_ref_pose, use_filter,
                     out33);
  return pi3_check_launch("sim3_umeyama");
}

extern "C" int pi3_sim3_umeyama(const void* pts_ref, const void* pts_qry, const int* idx, const unsigned char* w_ref,
                                const unsigned char* w_qry, int ov, int K, const float* last_ref_pose,
                                int use_filter, double* out33, void* stream) {
  return launch_sim3("pi3_sim3_umeyama: bad arguments", pts_ref, pts_qry, idx, w_ref, w_qry, nullptr, nullptr, ov, K,
                     last_ref_pose, use_filter, out33, stream);
}

// The weighted Umeyama of SURVEY.md §7 step 7: real-valued per-keypoint weights (e.g. mask * sigmoid(conf)); the pair
// weight is w_ref[ref track] * w_qry[qry keypoint], pairs with a weight that is not in (0, inf) do not take part.
extern "C" int pi3_sim3_umeyama_weighted(const void* pts_ref, const void* pts_qry, const int* idx, const float* w_ref,
                                         const float* w_qry, int ov, int K, const float* last_ref_pose, int use_filter,
                                         double* out33, void* stream) {
  if (!w_ref && !w_qry) {
    pi3_set_error("pi3_sim3_umeyama_weighted: no weights given (use pi3_sim3_umeyama)");
    return PI3_ERR_ARG;
  }
  return launch_sim3("pi3_sim3_umeyama_weighted: bad arguments", pts_ref, pts_qry, idx, nullptr, nullptr, w_ref, w_qry,
                     ov, K, last_ref_pose, use_filter, out33, stream);
}

// ---- 4. apply a 4x4 similarity (row-major doubles, device memory) to fp32 points [n][3] and cam->world poses [F][16]
__global__ __launch_bounds__(256) void sim3_apply_kernel(const double* __restrict__ M4, float* pts, long n,
                                                         float* poses, int F) {
  double A[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) A[i] = M4[i];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (long j = i; j < n; j += (long)gridDim.x * 256) {
    const double x = pts[3 * j], y = pts[3 * j + 1], z = pts[3 * j + 2];
    pts[3 * j + 0] = (float)(A[0] * x + A[1] * y + A[2] * z + A[3]);
    pts[3 * j + 1] = (float)(A[4] * x + A[5] * y + A[6] * z + A[7]);
    pts[3 * j + 2] = (float)(A[8] * x + A[9] * y + A[10] * z + A[11]);
  }
  if (i < F) {
    const double sc = sqrt(A[0] * A[0] + A[4] * A[4] + A[8] * A[8]);  // column norm of sR
    float* P = poses + 16 * i;
    double Pn[12];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c)
        Pn[4 * r + c] = (A[4 * r] * P[c] + A[4 * r + 1] * P[4 + c] + A[4 * r + 2] * P[8 + c]) / sc;  // R . R_cw
      Pn[4 * r + 3] = A[4 * r] * P[3] + A[4 * r + 1] * P[7] + A[4 * r + 2] * P[11] + A[4 * r + 3];    // sR c + t
    }
    for (int k = 0; k < 12; ++k) P[k] = (float)Pn[k];
  }
}

extern "C" int pi3_sim3_apply(const double* M4_dev, float* pts, long n, float* poses, int F, void* stream) {
  if (!M4_dev || (n > 0 && !pts) || (F > 0 && !poses) || n < 0 || F < 0 || (n == 0 && F == 0)) {
    pi3_set_error("pi3_sim3_apply: bad arguments");
    return PI3_ERR_ARG;
  }
  long work = n > F ? n : F;
  long blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks * 256 < F) blocks = (F + 255) / 256;
  hipLaunchKernelGGL(sim3_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, M4_dev, pts, n,
                     poses, F);
  return pi3_check_launch("sim3_apply");
}

// ---- prefix composition G_c = G_{c-1} . T_c of 4x4 similarities (chunk-parallel alignment: every rank composes the
// all-gathered relative transforms locally).  T, G: [n][16] row-major doubles; G_0 = T_0.
__global__ void sim3_compose_kernel(const double* __restrict__ T, double* __restrict__ G, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double cur[16];
  for (int i = 0; i < 16; ++i) cur[i] = T[i];
  for (int i = 0; i < 16; ++i) G[i] = cur[i];
  for (int c = 1; c < n; ++c) {
    double nx[16];
    for (int r = 0; r < 4; ++r)
      for (int k = 0; k < 4; ++k) {
        double acc = 0.0;
        for (int j = 0; j < 4; ++j) acc += cur[4 * r + j] * T[16 * c + 4 * j + k];
        nx[4 * r + k] = acc;
      }
    for (int i = 0; i < 16; ++i) { cur[i] = nx[i]; G[16 * c + i] = nx[i]; }
  }
}

extern "C" int pi3_sim3_compose_prefix(const double* T, double* G, int n, void* stream) {
  if (!T || !G || n <= 0) {
    pi3_set_error("pi3_sim3_compose_prefix: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(sim3_compose_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, T, G, n);
  return pi3_check_launch("sim3_compose_prefix");
}

// ---------------------------------------------------------------------------------------------------------------
// Observation projection of ChunkPTRecon.create_recon_from_chunk (utils/chunk_reconstruction.py:162-185, 445-509),
// every frame's K keypoint world points are projected into all EARLIER frames and into the next
// `max_after` (= max_observations_per_track // 2) frames:  x = K_t . ((T_t^-1 X)[:3] / z);  an observation is kept when
// 0 <= u < W and 0 <= v < H (no z > 0 test in the reference).  The reference is a python triple loop with one
// np.linalg.inv per (source, target) pair; here one workgroup per (source, target) pair, fp64 math like the
// reference's float64 promotion, the 4x4 inverse by Gauss-Jordan with partial pivoting.
// points: f16 [N][K][3]; poses: f32 [N][16] cam->world; intr: f32 [N][9].  uv: f32 [N][N][K][2] indexed
// [source][target]; valid: uint8 [N][N][K] (0 for pairs that are not projected at all).
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void inv4x4_d(const double* A, double* Ainv) {
  double M[4][8];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) { M[r][c] = A[4 * r + c]; M[r][4 + c] = (r == c) ? 1.0 : 0.0; }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (fabs(M[r][col]) > fabs(M[piv][col])) piv = r;
    if (piv != col)
      for (int c = 0; c < 8; ++c) { const double t = M[col][c]; M[col][c] = M[piv][c]; M[piv][c] = t; }
    const double d = 1.0 / M[col][col];
    for (int c = 0; c < 8; ++c) M[col][c] *= d;
    for (int r = 0; r < 4; ++r)
      if (r != col) {
        const double f = M[r][col];
        for (int c = 0; c < 8; ++c) M[r][c] -= f * M[col][c];
      }
  }
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) Ainv[4 * r + c] = M[r][4 + c];
}

__global__ __launch_bounds__(128) void project_obs_kernel(const __half* __restrict__ points,
                                                          const float* __restrict__ poses,
                                                          const float* __restrict__ intr, int N, int K, int W, int H,
                                                          int max_after, float* __restrict__ uv,
                                                          uint8_t* __restrict__ valid) {
  const int src = blockIdx.y, tgt = blockIdx.x;
  const bool active = (tgt < src) || (tgt > src && tgt <= src + max_after);
  const long base = ((long)src * N + tgt) * K;
  __shared__ double Winv[16];
  __shared__ double Kt[9];
  if (!active) {
    for (int k = threadIdx.x; k < K; k += 128) valid[base + k] = 0;
    return;
  }
  if (threadIdx.x == 0) {
    double P[16];
    for (int i = 0; i < 16; ++i) P[i] = (double)poses[16 * tgt + i];
    inv4x4_d(P, Winv);
    for (int i = 0; i < 9; ++i) Kt[i] = (double)intr[9 * tgt + i];
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 128) {
    const __half* p = points + ((long)src * K + k) * 3;
    const double X = (double)__half2float(p[0]), Y = (double)__half2float(p[1]), Z = (double)__half2float(p[2]);
    const double xc = Winv[0] * X + Winv[1] * Y + Winv[2] * Z + Winv[3];
    const double yc = Winv[4] * X + Winv[5] * Y + Winv[6] * Z + Winv[7];
    const double zc = Winv[8] * X + Winv[9] * Y + Winv[10] * Z + Winv[11];
    const double xn = xc / zc, yn = yc / zc, zn = zc / zc;
    const double u = Kt[0] * xn + Kt[1] * yn + Kt[2] * zn;
    const double v = Kt[3] * xn + Kt[4] * yn + Kt[5] * zn;
    uv[2 * (base + k)] = (float)u;
    uv[2 * (base + k) + 1] = (float)v;
    valid[base + k] = (u >= 0.0 && u < (double)W && v >= 0.0 && v < (double)H) ? 1 : 0;
  }
}

extern "C" int pi3_project_observations(const void* points, const float* poses, const float* intrinsics, int N, int K,
                                        int W, int H, int max_after, float* uv, unsigned char* valid, void* stream) {
  if (!points || !poses || !intrinsics || !uv || !valid || N <= 0 || K <= 0 || W <= 0 || H <= 0 || max_after < 0) {
    pi3_set_error("pi3_project_observations: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(project_obs_kernel, dim3(N, N), dim3(128), 0, (hipStream_t)stream, (const __half*)points, poses,
                     intrinsics, N, K, W, H, max_after, uv, valid);
  return pi3_check_launch("project_observations");
}
