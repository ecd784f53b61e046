// Per-chunk post-processing between the network and the chunk file: validity masks, MoGe metric scale (masked ratio
// median), scale application, keypoint gather + fp16 pack.  All HBM-bound or latency-bound; they exist so that the
// 350 MB of dense maps never leave the device (the reference copies them to the host and samples there:
// slam/offline_chunk_creator.py:204-213, 228).
#include "common.h"
#include <hip/hip_fp16.h>
#include <float.h>

// ---------------------------------------------------------------------------------------------------------------
// masks = sigmoid(conf) > thr  &  ~depth_edge(z, rtol)         (offline_chunk_creator.py:114-119)
// depth_edge (pi3/utils/geometry.py:347-375): diff = maxpool3(z) + maxpool3(-z) with max_pool2d's implicit -inf
// padding (borders use the valid neighbours only); edge = nan_to_num(diff / z) > rtol.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void masks_kernel(const float* __restrict__ conf, const float* __restrict__ lp,
                                                    int F, int H, int W, float thr, float rtol,
                                                    uint8_t* __restrict__ out) {
  const long n = (long)F * H * W;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W);
  const int y = (int)((i / W) % H);
  const float z = lp[3 * i + 2];
  float mx = z, mn = z;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const float v = lp[3 * (i + (long)dy * W + dx) + 2];
        mx = fmaxf(mx, v);
        mn = fminf(mn, v);
      }
    }
  float ratio = (mx + (-mn)) / z;
  if (isnan(ratio)) ratio = 0.f;
  else if (isinf(ratio)) ratio = ratio > 0.f ? FLT_MAX : -FLT_MAX;
  const bool edge = ratio > rtol;
  const float sg = 1.0f / (1.0f + expf(-conf[i]));
  out[i] = (sg > thr && !edge) ? 1 : 0;
}

extern "C" int pi3_compute_masks(const float* conf, const float* local_points, int F, int H, int W, float conf_thr,
                                 float rtol, unsigned char* masks, void* stream) {
  if (!conf || !local_points || !masks || F <= 0 || H <= 0 || W <= 0) {
    pi3_set_error("pi3_compute_masks: bad arguments");
    return PI3_ERR_ARG;
  }
  const long n = (long)F * H * W;
  hipLaunchKernelGGL(masks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, conf,
                     local_points, F, H, W, conf_thr, rtol, masks);
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
  const float s = scale[0];
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
// (nw = s*e, ne = s*w, sw = n*e, se = n*w; sum in that order); conf / mask: nearest = rint (half to even).
// Colours: uint8 truncation of 255*bilinear, then fp16 (so 0..255 values).  One thread per (frame, keypoint).
// ---------------------------------------------------------------------------------------------------------------
struct Bil { int x0, y0; float nw, ne, sw, se; int xn, yn; };

__device__ __forceinline__ Bil bil_setup(float kx, float ky, int H, int W) {
#pragma clang fp contract(off)
  Bil b;
  const float gx = (kx / (float)(W - 1)) * 2.0f - 1.0f;
  const float gy = (ky / (float)(H - 1)) * 2.0f - 1.0f;
  float x = (gx + 1.0f) * ((float)W / 2.0f) - 0.5f;
  float y = (gy + 1.0f) * ((float)H / 2.0f) - 0.5f;
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
  return ((vnw * b.nw + vne * b.ne) + vsw * b.sw) + vse * b.se;
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
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o_points[3 * i + c] = __float2half_rn(bil_sample(points + 3 * fo + c, 3, 3L * W, b, H, W));
    o_local[3 * i + c] = __float2half_rn(bil_sample(local_points + 3 * fo + c, 3, 3L * W, b, H, W));
  }
  const long pn = fo + (long)b.yn * W + b.xn;
  o_conf[i] = __float2half_rn(conf[pn]);
  o_mask[i] = masks[pn] ? 1 : 0;
  if (images) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = bil_sample(images + ((long)f * 3 + c) * H * W, 1, W, b, H, W) * 255.0f;
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
