// Per-chunk post-processing between the network and the chunk file: validity masks, MoGe metric scale (masked ratio
// median), scale application, keypoint gather + fp16 pack.  All HBM-bound or latency-bound; they exist so that the
// 350 MB of dense maps never leave the device (the reference copies them to the host and samples there:
// slam/offline_chunk_creator.py:204-213, 228).
#include "common.h"
#include <hip/hip_fp16.h>
#include <float.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------------------
// masks = sigmoid(conf) > thr  &  ~depth_edge(z, rtol)         (offline_chunk_creator.py:114-119)
// depth_edge (pi3/utils/geometry.py:347-375): diff = maxpool3(z) + maxpool3(-z) with max_pool2d's implicit -inf
// padding (borders use the valid neighbours only); edge = nan_to_num(diff / z) > rtol.
// ---------------------------------------------------------------------------------------------------------------
// One workgroup = a strip of MK_ROWS image rows of one frame (14: 308 = 22 x 14, 280 = 20 x 14; halo re-read 2 / 14).
// Phase 1 stages the z channel of the strip and its two halo rows in LDS from the interleaved (x, y, z) rows with
// 16-byte loads: a float4 at strip offset s = 3 pix + r holds the z of pixel pix at lane element (2 - r) mod 3 (and a
// second one, of pix + 1, when r == 2), and a thread's next float4 is 1024 floats on = 341 pixels + 1 float, so the
// walk needs no division (round 2 read one float per instruction and did `% 3` and `/ 3` per element: 1.1 TB/s).
// An aligned 16-byte load that holds one valid float cannot cross a page, so the ragged first / last float4 are safe.
// The LDS image has one extra column on either side holding a copy of the row's first / last pixel, and rows are
// addressed with a clamp: max_pool2d's implicit -inf padding means "ignore what is outside", and a replicated edge
// value is a value the window contains anyway - so phase 2 has NO bounds tests (they were 36 of ~100 instructions per
// pixel; the kernel is VALU-bound, not memory-bound: 9 window reads, NaN tracking, an IEEE division, a sigmoid).
// Phase 2 walks the strip as ONE contiguous pixel range (full rows of a [F][H][W] array are contiguous), consecutive
// lanes on consecutive pixels.
#define MK_ROWS 14
#define MK_MAXW 1024

// zr0 / zr1 / zr2: LDS rows above / at / below the pixel (clamped), already offset so that index x is the pixel's left
// neighbour (the image is stored from column 1).
// estar: the largest e >= 0 with fl(1 / fl(1 + e)) > thr (found on the host, see pi3_compute_masks): sigmoid(c) > thr
// <=> expf(-c) <= estar, decision for decision (the outer 1 / (1 + e) is monotone in e and correctly rounded on both
// sides), without the add and the IEEE division per pixel.
__device__ __forceinline__ uint8_t mask_pixel(const float* zr0, const float* zr1, const float* zr2, int x, float c,
                                              float estar, float rtol) {
  const float a0 = zr0[x], a1 = zr0[x + 1], a2 = zr0[x + 2];
  const float b0 = zr1[x], z = zr1[x + 1], b2 = zr1[x + 2];
  const float c0 = zr2[x], c1 = zr2[x + 1], c2 = zr2[x + 2];
  // max_pool2d propagates NaN (a NaN in the window wins), fmaxf would drop it
  const bool anynan = isnan(a0) | isnan(a1) | isnan(a2) | isnan(b0) | isnan(z) | isnan(b2) | isnan(c0) | isnan(c1) | isnan(c2);
  const float mx = fmaxf(fmaxf(fmaxf(a0, a1), fmaxf(a2, b0)), fmaxf(fmaxf(z, b2), fmaxf(fmaxf(c0, c1), c2)));
  const float mn = fminf(fminf(fminf(a0, a1), fminf(a2, b0)), fminf(fminf(z, b2), fminf(fminf(c0, c1), c2)));
  float ratio = anynan ? __uint_as_float(0x7fc00000u) : (mx + (-mn)) / z;
  if (isnan(ratio)) ratio = 0.f;
  else if (isinf(ratio)) ratio = ratio > 0.f ? FLT_MAX : -FLT_MAX;
  const bool edge = ratio > rtol;
  return (expf(-c) <= estar && !edge) ? 1 : 0;
}

__global__ __launch_bounds__(256) void masks_kernel(const float* __restrict__ conf, const float* __restrict__ lp,
                                                    int F, int H, int W, float thr, float rtol,
                                                    uint8_t* __restrict__ out) {   // thr: the e-threshold (estar)
  extern __shared__ __attribute__((aligned(16))) float zs[];          // (MK_ROWS + 2) rows of W + 2 floats
  const int strips = (H + MK_ROWS - 1) / MK_ROWS;
  const int f = blockIdx.x / strips, y0 = (blockIdx.x % strips) * MK_ROWS;
  const int tid = threadIdx.x;
  const int ylo = max(y0 - 1, 0), yhi = min(y0 + MK_ROWS + 1, H);       // staged rows [ylo, yhi)
  const int P = W + 2, nst = yhi - ylo;
  {
    const float* first = lp + ((long)f * H + ylo) * W * 3;
    const uintptr_t addr = (uintptr_t)first, aligned = addr & ~(uintptr_t)15;
    const int lead = (int)((addr - aligned) >> 2);                      // floats in front of the strip in float4 0
    const float4* src = (const float4*)aligned;
    const int npix = nst * W;
    const int nq = (lead + 3 * npix + 3) >> 2;
    const int s3 = 4 * tid - lead + 3;                                  // strip offset of element 0, biased by +3 (>= 0)
    int pix = s3 / 3 - 1, r = s3 - 3 * (s3 / 3);
    // (row, x) of pixel `pix`, kept incrementally; pix = -1 (only possible for the very first float4) maps to (0, -1)
    int row = pix < 0 ? 0 : pix / W, x = pix < 0 ? -1 : pix - row * W;
    const bool wide = W >= 342;                                         // one conditional subtraction per step is enough
    for (int q = tid; q < nq; q += 256) {
      const float4 v = src[q];
      const float za = r == 0 ? v.z : (r == 1 ? v.y : v.x);
      if (pix >= 0 && pix < npix) zs[row * P + 1 + x] = za;
      if (r == 2 && pix + 1 < npix) {
        const bool wrap = x + 1 >= W;
        zs[(wrap ? row + 1 : row) * P + 1 + (wrap ? 0 : x + 1)] = v.w;
      }
      pix += 341;                                                       // 256 threads x 4 floats = 1024 = 3 * 341 + 1
      x += 341;
      if (++r == 3) { r = 0; ++pix; ++x; }
      if (wide) {
        if (x >= W) { x -= W; ++row; }
      } else {
        row = pix / W;
        x = pix - row * W;
      }
    }
  }
  __syncthreads();
  if (tid < 2 * nst) {                                                  // replicated edge columns
    const int rr = tid >> 1;
    if (tid & 1) zs[rr * P + W + 1] = zs[rr * P + W];
    else zs[rr * P] = zs[rr * P + 1];
  }
  __syncthreads();
  const int rows = min(MK_ROWS, H - y0);
  const int npx = rows * W;
  const long p0 = ((long)f * H + y0) * W;                               // first pixel of the strip, frame-linear
  const int yoff = y0 - ylo;                                            // staged row of the strip's first row (0 or 1)
  // consecutive lanes take consecutive pixels: the nine window reads of a wave then walk consecutive LDS words (no bank
  // conflict; four pixels per lane made every read 4-way conflicted), confidence loads and mask stores are coalesced
  int ry = tid / W, x = tid - ry * W;
  const bool wide = W >= 256;
  for (int e = tid; e < npx; e += 256) {
    const int yc = yoff + ry;
    const float* zr1 = zs + yc * P;
    const float* zr0 = zs + max(yc - 1, 0) * P;
    const float* zr2 = zs + min(yc + 1, nst - 1) * P;
    out[p0 + e] = mask_pixel(zr0, zr1, zr2, x, conf[p0 + e], thr, rtol);
    x += 256;
    if (wide) {
      if (x >= W) { x -= W; ++ry; }
    } else {
      const int e2 = e + 256;
      ry = e2 / W;
      x = e2 - ry * W;
    }
  }
}

extern "C" int pi3_compute_masks(const float* conf, const float* local_points, int F, int H, int W, float conf_thr,
                                 float rtol, unsigned char* masks, void* stream) {
  if (!conf || !local_points || !masks || F <= 0 || H <= 0 || W <= 0 || W > MK_MAXW ||
      ((uintptr_t)local_points & 3) != 0) {
    pi3_set_error("pi3_compute_masks: bad arguments (W <= %d, 4-byte aligned maps)", MK_MAXW);
    return PI3_ERR_ARG;
  }
  // sigmoid(c) > conf_thr, i.e. fl(1 / fl(1 + e)) > conf_thr with e = expf(-c): the set of such e is an interval
  // [0, estar]; bisect estar over the bit patterns of the non-negative floats (they order like the values) with the
  // same two correctly rounded fp32 operations the kernel used to do per pixel
  float estar = -1.0f;
  {
    auto pass = [&](uint32_t bits) {
      float e;
      memcpy(&e, &bits, 4);
      volatile float d = 1.0f + e;
      volatile float sg = 1.0f / d;
      return sg > conf_thr;
    };
    if (pass(0u)) {
      uint32_t lo = 0u, hi = 0x7f800000u;          // pass(lo) holds; +inf fails unless conf_thr < 0
      if (pass(hi)) lo = hi;
      while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (pass(mid)) lo = mid; else hi = mid;
      }
      memcpy(&estar, &lo, 4);
    }
  }
  const long nwg = (long)F * ((H + MK_ROWS - 1) / MK_ROWS);
  hipLaunchKernelGGL(masks_kernel, dim3((unsigned)nwg), dim3(256), (size_t)(MK_ROWS + 2) * (W + 2) * sizeof(float),
                     (hipStream_t)stream, conf, local_points, F, H, W, estar, rtol, masks);
  return pi3_check_launch("compute_masks");
}

// ---------------------------------------------------------------------------------------------------------------
// scale = median((num / den)[mask])   (offline_chunk_creator.py:121-127; torch.median = LOWER median, i.e. the
// element of rank (n-1)/2).  Exact selection by a 4-pass 8-bit radix select over the order-preserving integer image
// of the fp32 ratios, one 1024-thread workgroup (n <= 125 k values at the north-star size).  NaN in -> NaN out, as
// torch.  out[0] = median, out[1] = number of selected values (0 -> median is NaN; torch would raise).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f32_orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_orderable(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(1024) void ratio_median_kernel(const float* __restrict__ num,
                                                            const float* __restrict__ den, long dstride,
                                                            const uint8_t* __restrict__ mask, long n,
                                                            float* __restrict__ out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_cnt, s_nan, s_prefix, s_k;
  const int tid = threadIdx.x;
  if (tid == 0) { s_cnt = 0; s_nan = 0; }
  __syncthreads();
  unsigned c = 0, cn = 0;
  for (long i = tid; i < n; i += 1024)
    if (mask[i]) {
      ++c;
      const float r = num[i] / den[i * dstride];
      if (isnan(r)) ++cn;
    }
  atomicAdd(&s_cnt, c);
  atomicAdd(&s_nan, cn);
  __syncthreads();
  const unsigned total = s_cnt;
  if (total == 0 || s_nan > 0) {
    if (tid == 0) { out[0] = __uint_as_float(0x7fc00000u); out[1] = (float)total; }
    return;
  }
  if (tid == 0) { s_prefix = 0; s_k = (total - 1) / 2; }
  uint32_t pmask = 0;
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    for (long i = tid; i < n; i += 1024)
      if (mask[i]) {
        const uint32_t key = f32_orderable(num[i] / den[i * dstride]);
        if ((key & pmask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1u);
      }
    __syncthreads();
    if (tid == 0) {
      unsigned k = s_k, cum = 0;
      int b = 0;
      for (; b < 256; ++b) {
        if (cum + hist[b] > k) break;
        cum += hist[b];
      }
      s_k = k - cum;
      s_prefix = prefix | ((uint32_t)b << (8 * pass));
    }
    pmask |= 0xffu << (8 * pass);
    __syncthreads();
  }
  if (tid == 0) { out[0] = f32_from_orderable(s_prefix); out[1] = (float)total; }
}

extern "C" int pi3_masked_ratio_median(const float* num, const float* den, long den_stride,
                                       const unsigned char* mask, long n, float* out2, void* stream) {
  if (!num || !den || !mask || !out2 || n <= 0 || den_stride <= 0) {
    pi3_set_error("pi3_masked_ratio_median: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(ratio_median_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, num, den, den_stride, mask, n,
                     out2);
  return pi3_check_launch("masked_ratio_median");
}

// local_points *= s, points *= s, camera_poses[:, :3, 3] *= s with s read from device memory (no host sync)
// (offline_chunk_creator.py:189-191).
__global__ __launch_bounds__(256) void apply_scale_kernel(const float* __restrict__ scale, float* lp, float* pts,
                                                          long n3, float* poses, int F) {
  float s = scale[0];
  // an empty / NaN / non-positive / infinite median (no valid pixel in frame 0) leaves the chunk unscaled: the
  // reference would raise on the empty median (offline_chunk_creator.py:126) and lose the chunk; we degrade instead
  if (!(s > 0.f) || isinf(s)) s = 1.0f;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long stride = (long)gridDim.x * 256;
  for (long j = i; j < n3; j += stride) {
    lp[j] *= s;
    pts[j] *= s;
  }
  if (i < 3L * F) {
    const int f = (int)(i / 3), r = (int)(i % 3);
    poses[16 * f + 4 * r + 3] *= s;
  }
}

extern "C" int pi3_apply_scale(const float* scale_dev, float* local_points, float* points, long n3, float* poses,
                               int F, void* stream) {
  if (!scale_dev || !local_points || !points || !poses || n3 <= 0 || F <= 0) {
    pi3_set_error("pi3_apply_scale: bad arguments");
    return PI3_ERR_ARG;
  }
  long blocks = (n3 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(apply_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, scale_dev,
                     local_points, points, n3, poses, F);
  return pi3_check_launch("apply_scale");
}

// ---------------------------------------------------------------------------------------------------------------
// Keypoint gather (offline_chunk_creator.py:129-159 + keypoint_extraction.py:203-229) with the fp16 pack of
// :231-241.  The reference normalises with x/(W-1)*2-1 and then calls grid_sample(align_corners=False,
// padding_mode='border'), whose CPU kernel un-normalises with (g+1)*(W/2)-0.5 and clips to [0, W-1]
// (ATen GridSamplerKernel.cpp): effective index x*W/(W-1)-0.5.  points / local_points / colours: bilinear
// (nw = s*e, ne = s*w, sw = n*e, se = n*w); conf / mask: nearest = rint (half to even).
// Rounding order = the vectorised ATen CPU kernel as built for x86 with FMA (what the reference executes: the gather
// runs on CPU tensors, offline_chunk_creator.py:204-228, keypoint_extraction.py:222): gcc contracts
// (g + 1) * (W/2) - 0.5 into one fma and nw_val*nw + ne_val*ne + sw_val*sw + se_val*se into
// fma(se_val, se, fma(sw_val, sw, fma(ne_val, ne, nw_val*nw))).  Restated with explicit fmaf here (contraction off for
// everything else), this reproduces F.grid_sample bit for bit (tests/test_post_gpu.py gates 0 ulp / 0 LSB).
// Colours: uint8 truncation of 255*bilinear, then fp16 (so 0..255 values).  One thread per (frame, keypoint).
// ---------------------------------------------------------------------------------------------------------------
struct Bil { int x0, y0; float nw, ne, sw, se; int xn, yn; };

__device__ __forceinline__ Bil bil_setup(float kx, float ky, int H, int W) {
#pragma clang fp contract(off)
  Bil b;
  const float gx = (kx / (float)(W - 1)) * 2.0f - 1.0f;
  const float gy = (ky / (float)(H - 1)) * 2.0f - 1.0f;
  float x = __builtin_fmaf(gx + 1.0f, (float)W / 2.0f, -0.5f);
  float y = __builtin_fmaf(gy + 1.0f, (float)H / 2.0f, -0.5f);
  x = fminf((float)(W - 1), fmaxf(x, 0.0f));
  y = fminf((float)(H - 1), fmaxf(y, 0.0f));
  const float xw = floorf(x), yn = floorf(y);
  const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
  b.nw = s * e; b.ne = s * w; b.sw = n * e; b.se = n * w;
  b.x0 = (int)xw; b.y0 = (int)yn;
  b.xn = (int)rintf(x); b.yn = (int)rintf(y);
  return b;
}

__device__ __forceinline__ float bil_sample(const float* __restrict__ base, long sx, long sy, const Bil& b, int H,
                                            int W) {
#pragma clang fp contract(off)
  const bool xe = b.x0 + 1 < W, ys = b.y0 + 1 < H;
  const float* p = base + b.y0 * sy + b.x0 * sx;
  const float vnw = p[0];
  const float vne = xe ? p[sx] : 0.f;
  const float vsw = ys ? p[sy] : 0.f;
  const float vse = (xe && ys) ? p[sy + sx] : 0.f;
  return __builtin_fmaf(vse, b.se, __builtin_fmaf(vsw, b.sw, __builtin_fmaf(vne, b.ne, vnw * b.nw)));
}

__global__ __launch_bounds__(256) void gather_keypoints_kernel(
    const float* __restrict__ points, const float* __restrict__ local_points, const float* __restrict__ conf,
    const uint8_t* __restrict__ masks, const float* __restrict__ images, const float* __restrict__ kps, int F, int H,
    int W, int K, __half* __restrict__ o_points, __half* __restrict__ o_local, __half* __restrict__ o_conf,
    uint8_t* __restrict__ o_mask, __half* __restrict__ o_colors, __half* __restrict__ o_kps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)F * K) return;
  const int f = (int)(i / K);
  const float kx = kps[2 * i], ky = kps[2 * i + 1];
  const Bil b = bil_setup(kx, ky, H, W);
  const long fo = (long)f * H * W;
  // The fp32 value must exist before it is rounded to fp16, as in the reference (grid_sample returns fp32, .to(float16)
  // rounds it a second time).  Left to itself LLVM folds fptrunc(fma) into v_fma_mixlo_f16, which rounds the exact
  // fma result ONCE: a different fp16 value whenever the fp32 result is an fp16 tie (1 element in ~60 000; found by
  // the full-chunk comparison with the oracle, tests/test_fullsize_gpu.py).  The empty asm pins the fp32 rounding.
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float vp = bil_sample(points + 3 * fo + c, 3, 3L * W, b, H, W);
    float vl = bil_sample(local_points + 3 * fo + c, 3, 3L * W, b, H, W);
    asm volatile("" : "+v"(vp), "+v"(vl));
    o_points[3 * i + c] = __float2half_rn(vp);
    o_local[3 * i + c] = __float2half_rn(vl);
  }
  const long pn = fo + (long)b.yn * W + b.xn;
  o_conf[i] = __float2half_rn(conf[pn]);
  o_mask[i] = masks[pn] ? 1 : 0;
  if (images) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float vb = bil_sample(images + ((long)f * 3 + c) * H * W, 1, W, b, H, W);
      asm volatile("" : "+v"(vb));     // the fp32 sample, then * 255 as its own rounding (keypoint_extraction.py:229)
      const float v = vb * 255.0f;
      const int u = (int)v;  // .to(torch.uint8): truncation
      o_colors[3 * i + c] = __float2half_rn((float)(u & 255));
    }
  }
  o_kps[2 * i] = __float2half_rn(kx);
  o_kps[2 * i + 1] = __float2half_rn(ky);
}

extern "C" int pi3_gather_keypoints(const float* points, const float* local_points, const float* conf,
                                    const unsigned char* masks, const float* images, const float* keypoints, int F,
                                    int H, int W, int K, void* o_points, void* o_local, void* o_conf,
                                    unsigned char* o_mask, void* o_colors, void* o_kps, void* stream) {
  if (!points || !local_points || !conf || !masks || !keypoints || !o_points || !o_local || !o_conf || !o_mask ||
      !o_kps || (images && !o_colors) || F <= 0 || H < 2 || W < 2 || K <= 0) {
    pi3_set_error("pi3_gather_keypoints: bad arguments F=%d H=%d W=%d K=%d", F, H, W, K);
    return PI3_ERR_ARG;
  }
  const long n = (long)F * K;
  hipLaunchKernelGGL(gather_keypoints_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     points, local_points, conf, masks, images, keypoints, F, H, W, K, (__half*)o_points,
                     (__half*)o_local, (__half*)o_conf, o_mask, (__half*)o_colors, (__half*)o_kps);
  return pi3_check_launch("gather_keypoints");
}

// ---------------------------------------------------------------------------------------------------------------
// Per-frame focal / shift recovery = estimate_camera_parameters (utils/camera_estimation.py:12-70) ->
// recover_focal_shift (utils/geometry_torch.py:114-169) -> solve_optimal_focal_shift (utils/geometry_numpy.py:79-96).
//   * nearest 64x64 down-sampling of local_points, uv and mask = sigmoid(conf) > 0.1 (F.interpolate 'nearest':
//     src = floor(dst * in/out) in fp32);
//   * scipy.optimize.least_squares(fn, x0=0, ftol=1e-3, method='lm') on the z-shift, i.e. MINPACK lmdif with n = 1,
//     forward-difference Jacobian (epsfcn = DBL_EPSILON), diag = 1 (mode 2), factor = 100, xtol = gtol = 1e-8,
//     maxfev = 200.  With n = 1 the QR factorisation collapses to r = |J|, q^T f = J.f / |J|, so the whole solver is
//     a handful of fp64 reductions over <= 4096 points; residual fn(shift) = f * xy/(z+shift) - uv with the closed
//     form f = sum(xyp.uv) / sum(xyp^2).  The 100 frames of a chunk are 100 independent workgroups instead of the
//     reference's serial python loop (geometry_torch.py:149-161).
//   * fx, fy, cx, cy and the 3x3 matrix (camera_estimation.py:47-57) in fp32 with the reference's operation order.
// uvx [W], uvy [H]: the fp32 torch.linspace tables of normalized_view_plane_uv (geometry_torch.py:39-51), host built.
// ---------------------------------------------------------------------------------------------------------------
#define FS_N 64
struct FsPoint { float x, y, z, u, v; };

__device__ __forceinline__ double fs_block_sum(double v, double* red, int tid) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// sums needed for the closed-form focal at a given shift
__device__ __forceinline__ void fs_focal(const FsPoint* pts, int n, double shift, double* red, int tid, double& f) {
  double a = 0.0, b = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double inv = 1.0 / ((double)pts[i].z + shift);
    const double px = pts[i].x * inv, py = pts[i].y * inv;
    a += px * (double)pts[i].u + py * (double)pts[i].v;
    b += px * px + py * py;
  }
  a = fs_block_sum(a, red, tid);
  b = fs_block_sum(b, red, tid);
  f = a / b;
}

__device__ __forceinline__ double fs_resnorm2(const FsPoint* pts, int n, double shift, double f, double* red,
                                              int tid) {
  double s = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double inv = 1.0 / ((double)pts[i].z + shift);
    const double ex = f * pts[i].x * inv - (double)pts[i].u, ey = f * pts[i].y * inv - (double)pts[i].v;
    s += ex * ex + ey * ey;
  }
  return fs_block_sum(s, red, tid);
}

__global__ __launch_bounds__(256) void focal_shift_kernel(const float* __restrict__ local_points,
                                                          const float* __restrict__ conf,
                                                          const uint8_t* __restrict__ mask8,
                                                          const float* __restrict__ uvx,
                                                          const float* __restrict__ uvy, int H, int W, float conf_thr,
                                                          float* __restrict__ o_focal, float* __restrict__ o_shift,
                                                          float* __restrict__ o_fxfycxcy, float* __restrict__ o_K) {
  __shared__ FsPoint pts[FS_N * FS_N];
  __shared__ double red[4];
  __shared__ int s_n;
  __shared__ int woff[5];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long fo = (long)f * H * W;
  const float sy = (float)H / (float)FS_N, sx = (float)W / (float)FS_N;
  // ordered compaction (row-major like numpy boolean indexing) so the point order, hence the summation order inside a
  // thread's strided subset, is deterministic
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int base = 0; base < FS_N * FS_N; base += 256) {
    const int k = base + tid;
    const int i = k / FS_N, j = k - i * FS_N;
    const int yy = min((int)floorf(i * sy), H - 1), xx = min((int)floorf(j * sx), W - 1);
    const long p = fo + (long)yy * W + xx;
    const bool ok = mask8 ? (mask8[p] != 0) : ((1.0f / (1.0f + expf(-conf[p]))) > conf_thr);
    const unsigned long long bal = __ballot(ok);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) woff[wave + 1] = __popcll(bal);
    __syncthreads();
    if (tid == 0) {
      woff[0] = s_n;
      for (int w = 1; w <= 4; ++w) woff[w] += woff[w - 1];
      s_n = woff[4];
    }
    __syncthreads();
    if (ok) {
      FsPoint q;
      q.x = local_points[3 * p]; q.y = local_points[3 * p + 1]; q.z = local_points[3 * p + 2];
      q.u = uvx[xx]; q.v = uvy[yy];
      pts[woff[wave] + before] = q;
    }
    __syncthreads();
  }
  const int n = s_n;
  double x = 0.0, focal = 1.0;
  if (n >= 2) {
    const double epsmch = 2.220446049250313e-16, ftol = 1e-3, xtol = 1e-8, gtol = 1e-8, factor = 100.0;
    const double p1 = 0.1, p5 = 0.5, p25 = 0.25, p75 = 0.75, p0001 = 1e-4, dwarf = 2.2250738585072014e-308;
    double fx_; fs_focal(pts, n, x, red, tid, fx_);
    double fnorm = sqrt(fs_resnorm2(pts, n, x, fx_, red, tid));
    int nfev = 1, iter = 1, info = 0;
    double par = 0.0, delta = 0.0, xnorm = 0.0;
    const int maxfev = 200;
    while (info == 0) {
      // forward-difference Jacobian column J = (fn(x+h) - fn(x)) / h
      const double eps = sqrt(epsmch);
      double h = eps * fabs(x);
      if (h == 0.0) h = eps;
      double fh; fs_focal(pts, n, x + h, red, tid, fh);
      ++nfev;
      double jj = 0.0, jf = 0.0;
      for (int i = tid; i < n; i += 256) {
        const double inv0 = 1.0 / ((double)pts[i].z + x), inv1 = 1.0 / ((double)pts[i].z + (x + h));
        const double e0x = fx_ * pts[i].x * inv0 - (double)pts[i].u, e0y = fx_ * pts[i].y * inv0 - (double)pts[i].v;
        const double e1x = fh * pts[i].x * inv1 - (double)pts[i].u, e1y = fh * pts[i].y * inv1 - (double)pts[i].v;
        const double jx = (e1x - e0x) / h, jy = (e1y - e0y) / h;
        jj += jx * jx + jy * jy;
        jf += jx * e0x + jy * e0y;
      }
      jj = fs_block_sum(jj, red, tid);
      jf = fs_block_sum(jf, red, tid);
      const double r = sqrt(jj);                       // |R| of the 1-column QR; acnorm
      const double qtf = r != 0.0 ? jf / r : 0.0;      // |q^T f|-signed so that qtf / r == J.f / |J|^2
      if (iter == 1) {
        xnorm = fabs(x);
        delta = factor * xnorm;
        if (delta == 0.0) delta = factor;
      }
      double gnorm = 0.0;
      if (fnorm != 0.0 && r != 0.0) gnorm = fabs(r * (qtf / fnorm) / r);
      if (gnorm <= gtol) { info = 4; break; }
      double ratio = 0.0;
      do {
        // ---- lmpar, n = 1, diag = 1
        double p = r != 0.0 ? qtf / r : 0.0;          // Gauss-Newton step (to be negated)
        {
          double dxnorm = fabs(p), fp = dxnorm - delta;
          int it = 0;
          if (fp <= p1 * delta) {
            par = 0.0;
          } else {
            double parl = 0.0;
            if (r != 0.0) { const double t = (1.0) / r; const double tn = fabs(t); parl = ((fp / delta) / tn) / tn; }
            const double gn = fabs(r * qtf);
            double paru = gn / delta;
            if (paru == 0.0) paru = dwarf / fmin(delta, p1);
            par = fmax(par, parl);
            par = fmin(par, paru);
            if (par == 0.0) par = gn / dxnorm;
            for (;;) {
              ++it;
              if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
              const double sd = sqrt(r * r + par);     // R of [r; sqrt(par)]
              p = r * qtf / (sd * sd);
              dxnorm = fabs(p);
              const double tmp = fp;
              fp = dxnorm - delta;
              if (fabs(fp) <= p1 * delta || (parl == 0.0 && fp <= tmp && tmp < 0.0) || it == 10) break;
              const double w = (1.0) / sd;             // diag * (wa2 / dxnorm) solved with R^T: |.| = 1 / sd
              const double parc = ((fp / delta) / w) / w;
              if (fp > 0.0) parl = fmax(parl, par);
              if (fp < 0.0) paru = fmin(paru, par);
              par = fmax(parl, par + parc);
            }
          }
        }
        const double step = -p;
        const double xnew = x + step;
        const double pnorm = fabs(step);
        if (iter == 1) delta = fmin(delta, pnorm);
        double fnew; fs_focal(pts, n, xnew, red, tid, fnew);
        const double fnorm1 = sqrt(fs_resnorm2(pts, n, xnew, fnew, red, tid));
        ++nfev;
        double actred = -1.0;
        if (p1 * fnorm1 < fnorm) { const double q = fnorm1 / fnorm; actred = 1.0 - q * q; }
        const double temp1 = fabs(r * step) / fnorm, temp2 = sqrt(par) * pnorm / fnorm;
        const double prered = temp1 * temp1 + temp2 * temp2 / p5;
        const double dirder = -(temp1 * temp1 + temp2 * temp2);
        ratio = prered != 0.0 ? actred / prered : 0.0;
        if (ratio <= p25) {
          double temp = actred >= 0.0 ? p5 : p5 * dirder / (dirder + p5 * actred);
          if (p1 * fnorm1 >= fnorm || temp < p1) temp = p1;
          delta = temp * fmin(delta, pnorm / p1);
          par = par / temp;
        } else if (par == 0.0 || ratio >= p75) {
          delta = pnorm / p5;
          par = p5 * par;
        }
        if (ratio >= p0001) {
          x = xnew; fx_ = fnew; xnorm = fabs(x); fnorm = fnorm1; ++iter;
        }
        if (fabs(actred) <= ftol && prered <= ftol && p5 * ratio <= 1.0) info = 1;
        if (delta <= xtol * xnorm) info = 2;
        if (fabs(actred) <= ftol && prered <= ftol && p5 * ratio <= 1.0 && info == 2) info = 3;
        if (info != 0) break;
        if (nfev >= maxfev) info = 5;
        if (fabs(actred) <= epsmch && prered <= epsmch && p5 * ratio <= 1.0) info = 6;
        if (delta <= epsmch * xnorm) info = 7;
        if (gnorm <= epsmch) info = 8;
        if (info != 0) break;
      } while (ratio < p0001);
    }
    // optim_shift is cast to float32 BEFORE the final focal evaluation (geometry_numpy.py:91-94)
    const float xs = (float)x;
    fs_focal(pts, n, (double)xs, red, tid, focal);
    x = (double)xs;
  } else {
    x = 0.0; focal = 1.0;  // geometry_torch.py:152-155
  }
  if (tid == 0) {
    const float fo32 = (float)focal;
    o_focal[f] = fo32;
    o_shift[f] = (float)x;
    const double ar = (double)W / (double)H;
    const float c1 = (float)sqrt(1.0 + ar * ar), arf = (float)ar;
    const float fxv = (((fo32 / 2.0f) * c1) / arf) * (float)W;   // camera_estimation.py:47
    const float fyv = ((fo32 / 2.0f) * c1) * (float)H;           // :48
    const float cxv = (float)(W / 2), cyv = (float)(H / 2);      // :51-52 (integer halves)
    o_fxfycxcy[4 * f + 0] = fxv; o_fxfycxcy[4 * f + 1] = fyv; o_fxfycxcy[4 * f + 2] = cxv; o_fxfycxcy[4 * f + 3] = cyv;
    float* K = o_K + 9 * f;
    K[0] = fxv; K[1] = 0.f; K[2] = cxv; K[3] = 0.f; K[4] = fyv; K[5] = cyv; K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
  }
}

// Validity of a pixel: mask8 != NULL ? mask8[p] != 0 : sigmoid(conf[p]) > conf_thr (MoGe passes its own binary mask,
// moge/model/v2.py:241; the pi3 path uses the confidence, utils/camera_estimation.py:37).
extern "C" int pi3_focal_shift(const float* local_points, const float* conf, const unsigned char* mask8,
                               const float* uvx, const float* uvy, int F, int H, int W, float conf_thr, float* focal,
                               float* shift, float* fxfycxcy, float* K33, void* stream) {
  if (!local_points || (!conf && !mask8) || !uvx || !uvy || !focal || !shift || !fxfycxcy || !K33 || F <= 0 || H <= 0 || W <= 0) {
    pi3_set_error("pi3_focal_shift: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(focal_shift_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, local_points, conf, mask8, uvx,
                     uvy, H, W, conf_thr, focal, shift, fxfycxcy, K33);
  return pi3_check_launch("focal_shift");
}
