// Output heads of the pi3 network after the fp32 head GEMMs: unpatchify + exp/xy*z + pose transform, and the
// camera-head tail (token mean, MLP, fc_t / fc_rot, SO(3) projection).
#include "common.h"
#include "rot3.h"

// ---------------------------------------------------------------------------------------------------------------
// pfeat [F*P][ldp] fp32 = LinearPts3d.proj output for the point head (588 used columns), cfeat [F*P][ldc] for the
// conf head (196 used columns).  transformer_head.py:70-80: transpose -> view(B, c*196, h, w) -> pixel_shuffle(14)
// -> permute  ==>  out[f, y, x, c] = feat[f*P + (y/14)*pw + x/14][c*196 + (y%14)*14 + x%14]   (SURVEY.md §8 a7).
// pi3.py:195-198: z = exp(z); local = (x*z, y*z, z).  pi3.py:209: points = (pose @ [local, 1])[:3].
// Round 5 (LDS-staged tiles, as the north star prescribes): a workgroup owns one row of up to 32 patches of a frame.
// Its feature rows - 588 (+196) contiguous floats per patch - are read as whole 16-byte-per-lane rows and staged in LDS
// (row stride 592 floats: a pixel row's 14-float runs of consecutive patches then fall on disjoint banks); the 14 pixel
// rows of the patch row are then produced pixel by pixel from LDS, so that consecutive lanes write consecutive pixels.
// (The first form - one thread per pixel reading its three values straight from the feature rows, 56-byte runs 2 352
// bytes apart - ran at 4.1 TB/s of algorithmic bytes, 0.51 of the HBM peak.)
// ---------------------------------------------------------------------------------------------------------------
#define UNP_SEG 8          // patches per workgroup (25 KB of LDS: six workgroups per CU; 32 patches = 100 KB ran one workgroup per CU at 2.8 TB/s)
#define UNP_PSTRIDE 592    // floats per staged point-feature row (588 used)
#define UNP_CSTRIDE 208    // floats per staged conf-feature row (196 used; 208 = 13 x 16: disjoint banks again)
__global__ __launch_bounds__(256) void unpatchify_points_kernel(const float* __restrict__ pfeat, long ldp,
                                                                const float* __restrict__ cfeat, long ldc,
                                                                const float* __restrict__ poses, int F, int H, int W,
                                                                int T, int tok_off, float* __restrict__ local_points,
                                                                float* __restrict__ points, float* __restrict__ conf) {
  extern __shared__ __attribute__((aligned(16))) float unp_lds[];
  float* lp_ = unp_lds;                               // [UNP_SEG][UNP_PSTRIDE]
  float* lc_ = unp_lds + UNP_SEG * UNP_PSTRIDE;       // [UNP_SEG][UNP_CSTRIDE]
  const int pw = W / 14, ph = H / 14;
  const int nseg = (pw + UNP_SEG - 1) / UNP_SEG;
  const int seg = blockIdx.x % nseg;
  const int py = (blockIdx.x / nseg) % ph;
  const int f = blockIdx.x / (nseg * ph);
  const int px0 = seg * UNP_SEG, np = min(UNP_SEG, pw - px0);
  const long tok0 = (long)f * T + tok_off + py * pw + px0;
  const bool vecp = (ldp & 3) == 0 && ((uintptr_t)pfeat & 15) == 0, vecc = (ldc & 3) == 0 && ((uintptr_t)cfeat & 15) == 0;
  // ---- stage: 147 float4 per point row, 49 per conf row
  for (int i = threadIdx.x; i < np * 147; i += 256) {
    const int p = i / 147, q = i - p * 147;
    const float* src = pfeat + (tok0 + p) * ldp + 4 * q;
    f32x4 v;
    if (vecp) v = *(const f32x4*)src;
    else v = (f32x4){src[0], src[1], src[2], src[3]};
    *(f32x4*)(lp_ + p * UNP_PSTRIDE + 4 * q) = v;
  }
  for (int i = threadIdx.x; i < np * 49; i += 256) {
    const int p = i / 49, q = i - p * 49;
    const float* src = cfeat + (tok0 + p) * ldc + 4 * q;
    f32x4 v;
    if (vecc) v = *(const f32x4*)src;
    else v = (f32x4){src[0], src[1], src[2], src[3]};
    *(f32x4*)(lc_ + p * UNP_CSTRIDE + 4 * q) = v;
  }
  __syncthreads();
  const float* Tm = poses + (long)f * 16;
  float tm[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) tm[r] = Tm[r];
  const int wseg = np * 14;                           // pixels per row in this segment
  for (int i = threadIdx.x; i < 14 * wseg; i += 256) {
    const int ky = i / wseg, xs = i - ky * wseg;
    const int p = xs / 14, kx = xs - p * 14;
    const int sub = ky * 14 + kx;
    const float* pr = lp_ + p * UNP_PSTRIDE + sub;
    const float vx = pr[0], vy = pr[196], vz = pr[392];
    const float z = expf(vz);
    const float lx = vx * z, ly = vy * z;
    const long pix = ((long)f * H + py * 14 + ky) * W + px0 * 14 + xs;
    local_points[3 * pix + 0] = lx;
    local_points[3 * pix + 1] = ly;
    local_points[3 * pix + 2] = z;
    conf[pix] = lc_[p * UNP_CSTRIDE + sub];
#pragma unroll
    for (int r = 0; r < 3; ++r) points[3 * pix + r] = ((tm[4 * r] * lx + tm[4 * r + 1] * ly) + tm[4 * r + 2] * z) + tm[4 * r + 3];
  }
}

// Feature rows are addressed as f*T + tok_off + patch, so the head GEMMs may run over all T tokens of a frame
// (tok_off = 5 skips the register tokens, pi3.py:195 `[:, self.patch_start_idx:]`) or over patch rows only (T = P).
extern "C" int pi3_unpatchify_points(const float* pfeat, long ldp, const float* cfeat, long ldc, const float* poses,
                                     int F, int H, int W, int T, int tok_off, float* local_points, float* points,
                                     float* conf, void* stream) {
  if (!pfeat || !cfeat || !poses || !local_points || !points || !conf || F <= 0 || H <= 0 || W <= 0 || (H % 14) ||
      (W % 14) || ldp < 588 || ldc < 196 || tok_off < 0 || T < tok_off + (H / 14) * (W / 14)) {
    pi3_set_error("pi3_unpatchify_points: bad arguments F=%d H=%d W=%d", F, H, W);
    return PI3_ERR_ARG;
  }
  const int pw = W / 14, ph = H / 14, nseg = (pw + UNP_SEG - 1) / UNP_SEG;
  const long nwg = (long)F * ph * nseg;
  constexpr int lds_bytes = UNP_SEG * (UNP_PSTRIDE + UNP_CSTRIDE) * 4;      // 100 KB
  static unsigned long long optin = 0;
  if (int rc = pi3_lds_optin((const void*)unpatchify_points_kernel, lds_bytes, &optin, "unpatchify_points")) return rc;
  if (nwg > 0x7fffffffL) {
    pi3_set_error("pi3_unpatchify_points: grid too large");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(unpatchify_points_kernel, dim3((unsigned)nwg), dim3(256), lds_bytes, (hipStream_t)stream, pfeat, ldp,
                     cfeat, ldc, poses, F, H, W, T, tok_off, local_points, points, conf);
  return pi3_check_launch("unpatchify_points");
}

// ---------------------------------------------------------------------------------------------------------------
// Camera-head tail (camera_head.py:55-72): feat [F*P][ld] fp32 is the output of the two ResConvBlocks (run as fp32
// GEMMs).  Per frame: mean over the P patch tokens (AdaptiveAvgPool2d(1)), two Linear+ReLU (D x D), fc_t (3 x D),
// fc_rot (9 x D), rows of the 3x3 normalised (F.normalize, eps 1e-12), SO(3) projection (rot3.h), 4x4 pose.
// One 1024-thread workgroup per frame.  The kernel is a chain of dependent loads, so its time is set by how many are in
// flight: the pool splits the rows over 8 phases with four float4 loads outstanding per thread, and a wave computes four
// output neurons at a time (coalesced weight rows, 32 loads in flight) - 16 waves share a layer's neurons.
// ---------------------------------------------------------------------------------------------------------------
#define CAM_MAXD 1024
#define CAM_WAVES 16
__device__ __forceinline__ void cam_linear(const float* __restrict__ Wt, const float* __restrict__ bias,
                                           const float* vin, float* vout, int nout, int D, bool relu, int wave,
                                           int lane) {
  for (int j0 = 4 * wave; j0 < nout; j0 += 4 * CAM_WAVES) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < D; k += 64) {
      const float x = vin[k];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < nout) acc[u] += Wt[(long)(j0 + u) * D + k] * x;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float a = wave_sum(acc[u]);
      if (lane == 0 && j0 + u < nout) {
        const float v = a + bias[j0 + u];
        vout[j0 + u] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

__global__ __launch_bounds__(1024) void camera_tail_kernel(const float* __restrict__ feat, long ld, long fstride,
                                                           int P, int D,
                                                           const float* w1, const float* b1, const float* w2,
                                                           const float* b2, const float* wt, const float* bt,
                                                           const float* wr, const float* br, float* __restrict__ poses) {
  __shared__ float va[CAM_MAXD], vb[CAM_MAXD], tr[12];
  __shared__ __attribute__((aligned(16))) float part[8][CAM_MAXD];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // AdaptiveAvgPool2d(1): thread = (column group of 4, row phase); partial sums per phase, added in phase order
  const int ng = D >> 2;                       // float4 column groups (D % 4 == 0, checked on the host)
  const int np = min(8, 1024 / ng);            // row phases
  const int g = tid % ng, ph = tid / ng;
  if (ph < np) {
    f32x4 a4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a4[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* col = feat + (long)f * fstride + 4 * g;
    int p = ph;
    for (; p + 3 * np < P; p += 4 * np) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a4[u] += *(const f32x4*)(col + (long)(p + u * np) * ld);
    }
    for (; p < P; p += np) a4[0] += *(const f32x4*)(col + (long)p * ld);
    *(f32x4*)&part[ph][4 * g] = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  }
  __syncthreads();
  for (int c = tid; c < D; c += 1024) {
    float v = 0.f;
    for (int q = 0; q < np; ++q) v += part[q][c];
    va[c] = v / (float)P;
  }
  __syncthreads();
  cam_linear(w1, b1, va, vb, D, D, true, wave, lane);
  __syncthreads();
  cam_linear(w2, b2, vb, va, D, D, true, wave, lane);
  __syncthreads();
  if (wave < 8) cam_linear(wt, bt, va, tr, 3, D, false, wave, lane);
  else cam_linear(wr, br, va, tr + 3, 9, D, false, wave - 8, lane);
  __syncthreads();
  if (tid == 0) {
    double A[9], R[9];
    for (int r = 0; r < 3; ++r) {
      const float a = tr[3 + 3 * r], b = tr[4 + 3 * r], c = tr[5 + 3 * r];
      const float nrm = fmaxf(sqrtf(a * a + b * b + c * c), 1e-12f);
      A[3 * r] = a / nrm;
      A[3 * r + 1] = b / nrm;
      A[3 * r + 2] = c / nrm;
    }
    nearest_rotation_d(A, R);
    float* T = poses + (long)f * 16;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T[4 * r + c] = (float)R[3 * r + c];
      T[4 * r + 3] = tr[r];
    }
    T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
  }
}

// feat points at the first PATCH token of frame 0; fstride = elements between frames (T * ld when registers are kept).
extern "C" int pi3_camera_tail(const float* feat, long ld, long fstride, int F, int P, int D, const float* w1,
                               const float* b1,
                               const float* w2, const float* b2, const float* wt, const float* bt, const float* wr,
                               const float* br, float* poses, void* stream) {
  if (!feat || !w1 || !b1 || !w2 || !b2 || !wt || !bt || !wr || !br || !poses || F <= 0 || P <= 0 || D <= 0 ||
      D > CAM_MAXD || (D % 4) || (ld % 4) || (fstride % 4) || ((uintptr_t)feat & 15)) {
    pi3_set_error("pi3_camera_tail: bad arguments F=%d P=%d D=%d (D <= %d, D %% 4 == 0, 16-byte aligned rows)", F, P, D, CAM_MAXD);
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(camera_tail_kernel, dim3(F), dim3(1024), 0, (hipStream_t)stream, feat, ld, fstride, P, D, w1, b1,
                     w2, b2, wt, bt, wr, br, poses);
  return pi3_check_launch("camera_tail");
}
