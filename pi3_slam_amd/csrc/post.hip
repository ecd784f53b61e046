// Per-chunk post-processing between the network and the chunk file: validity masks, MoGe metric scale (masked ratio
// median), scale application, keypoint gather + fp16 pack.  All HBM-bound or latency-bound; they exist so that the
// 350 MB of dense maps never leave the device (the reference copies them to the host and samples there:
// slam/offline_chunk_creator.py:204-213, 228).
#include "common.h"
#include <hip/hip_fp16.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

// ---------------------------------------------------------------------------------------------------------------
// masks = sigmoid(conf) > thr  &  ~depth_edge(z, rtol)         (offline_chunk_creator.py:114-119)
// depth_edge (pi3/utils/geometry.py:347-375): diff = maxpool3(z) + maxpool3(-z) with max_pool2d's implicit -inf
// padding (borders use the valid neighbours only); edge = nan_to_num(diff / z) > rtol.
// ---------------------------------------------------------------------------------------------------------------
// One workgroup = a strip of R image rows of one frame (R chosen on the host so that five workgroups share a CU's LDS).
// Staging (round 3, third form): the strip's interleaved (x, y, z) rows with one halo row either side, and the strip's
// confidences, are two contiguous byte ranges of the [F][H][W] arrays; both go to LDS by LDS-DMA
// (global_load_lds_dwordx4, 16 bytes per lane, from the 16-byte aligned address at or below the range) - every byte of
// the workgroup is in flight at once, no registers, no address walk.  (Round 2 read one float per instruction with `%
// 3` and `/ 3` per element: 1.1 TB/s; the first round-3 form walked float4 loads through registers into a compacted z
// image, one memory latency per loop trip: 2.35 TB/s.)  An aligned 16-byte load that holds one valid float cannot cross
// a page, so the ragged first / last 16 bytes are safe; lanes beyond the range re-read its last 16 bytes.
// The window then reads z of pixel p of the staged rows at float `lead + 3 p + 2`: consecutive lanes on consecutive
// pixels are 3 words apart, which 64 banks serve without conflict.  max_pool2d's implicit -inf padding means "ignore
// what is outside", and a clamped coordinate names a value the window holds anyway, so borders are two clamps.
#define MK_MAXW 1024
#define MK_LDS_BUDGET (32 * 1024)       // 5 workgroups per CU (160 KB): 3 rows at W = 406 / 448, measured best of 2..8

struct MaskWin { float a0, a1, a2, b0, z, b2, c0, c1, c2; };

// estar: the largest e >= 0 with fl(1 / fl(1 + e)) > thr (found on the host, see pi3_compute_masks): sigmoid(c) > thr
// <=> expf(-c) <= estar, decision for decision (the outer 1 / (1 + e) is monotone in e and correctly rounded on both
// sides), without the add and the IEEE division per pixel.
__device__ __forceinline__ uint8_t mask_pixel(const MaskWin& w, float c, float estar, float rtol) {
  // max_pool2d propagates NaN (a NaN in the window wins), fmaxf would drop it
  const bool anynan = isnan(w.a0) | isnan(w.a1) | isnan(w.a2) | isnan(w.b0) | isnan(w.z) | isnan(w.b2) | isnan(w.c0) |
                      isnan(w.c1) | isnan(w.c2);
  const float mx = fmaxf(fmaxf(fmaxf(w.a0, w.a1), fmaxf(w.a2, w.b0)), fmaxf(fmaxf(w.z, w.b2), fmaxf(fmaxf(w.c0, w.c1), w.c2)));
  const float mn = fminf(fminf(fminf(w.a0, w.a1), fminf(w.a2, w.b0)), fminf(fminf(w.z, w.b2), fminf(fminf(w.c0, w.c1), w.c2)));
  float ratio = anynan ? __uint_as_float(0x7fc00000u) : (mx + (-mn)) / w.z;
  if (isnan(ratio)) ratio = 0.f;
  else if (isinf(ratio)) ratio = ratio > 0.f ? FLT_MAX : -FLT_MAX;
  const bool edge = ratio > rtol;
  return (expf(-c) <= estar && !edge) ? 1 : 0;
}

// The same decisions without the IEEE division and the expf in all but a sliver of the cases.  Preconditions,
// established per strip / on the host: no NaN among the staged z (fmaxf / fminf then ARE max_pool2d), 0 <= rtol <
// FLT_MAX (nan_to_num then never changes the outcome of `> rtol`: NaN -> 0 and NaN itself both compare false, +inf ->
// FLT_MAX and +inf itself both compare true).
//  edge: q = d * rcp(z) is within a few ulp of d / z for 2^-100 < |z| < 2^100; outside [rlo, rhi] = rtol (1 -+ 1e-5) the
//        correctly rounded quotient is on the same side of rtol; inside (or z out of range, or q NaN) divide exactly.
//  conf: expf(-c) <= estar holds for c > chi and fails for c < clo, where [clo, chi] brackets -ln(estar) by 1e-5 (1 +
//        |ln estar|), 40x what the expf's few ulp can move the crossing; inside the bracket (or c NaN) evaluate it.
struct MaskFast { float rlo, rhi, clo, chi; int ok; };

__device__ __forceinline__ uint8_t mask_pixel_fast(const MaskWin& w, float c, float estar, float rtol, const MaskFast& mf) {
  const float mx = fmaxf(fmaxf(fmaxf(w.a0, w.a1), w.a2), fmaxf(fmaxf(fmaxf(w.b0, w.z), w.b2), fmaxf(fmaxf(w.c0, w.c1), w.c2)));
  const float mn = fminf(fminf(fminf(w.a0, w.a1), w.a2), fminf(fminf(fminf(w.b0, w.z), w.b2), fminf(fminf(w.c0, w.c1), w.c2)));
  const float d = mx + (-mn);
  const float q = d * __builtin_amdgcn_rcpf(w.z);
  const float az = fabsf(w.z);
  bool edge = q > mf.rhi;
  if (!(az > 0x1p-100f && az < 0x1p100f && (edge || q < mf.rlo))) {
    float ratio = d / w.z;
    if (isnan(ratio)) ratio = 0.f;
    else if (isinf(ratio)) ratio = ratio > 0.f ? FLT_MAX : -FLT_MAX;
    edge = ratio > rtol;
  }
  bool cok = c > mf.chi;
  if (!cok && !(c < mf.clo)) cok = expf(-c) <= estar;
  return (cok && !edge) ? 1 : 0;
}

// floats of LDS that a range of n floats starting `lead` floats into its first 16 bytes occupies (whole 64-lane DMAs)
__host__ __device__ __forceinline__ int mk_lds_floats(int lead, int n) { return (((lead + n + 3) >> 2) + 63) / 64 * 256; }

__device__ __forceinline__ void mk_stage(const float* first, int n, float* lds, int tid, int& lead) {
  const uintptr_t addr = (uintptr_t)first, aligned = addr & ~(uintptr_t)15;
  lead = (int)((addr - aligned) >> 2);
  const float4* src = (const float4*)aligned;
  const int nq = (lead + n + 3) >> 2;
  const int wave = tid >> 6, lane = tid & 63;
  for (int c = wave; c * 64 < nq; c += 4)
    __builtin_amdgcn_global_load_lds(GLB_PTR(src + min(c * 64 + lane, nq - 1)), LDS_PTR(lds + c * 256), 16, 0, 0);
}

__global__ __launch_bounds__(256) void masks_kernel(const float* __restrict__ conf, const float* __restrict__ lp,
                                                    int F, int H, int W, int R, int zfloats, float estar, float rtol,
                                                    MaskFast mf, uint8_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float zs[];          // staged rows (zfloats), then the confidences
  const int strips = (H + R - 1) / R;
  // workgroups go round-robin to the 8 XCDs: give XCD k the k-th eighth of the strips, in order, so that the halo rows
  // two neighbouring strips share are fetched into one L2 once (the grid is padded to a multiple of 8)
  const int per_xcd = gridDim.x >> 3;
  const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (wg >= F * strips) return;
  const int f = wg / strips, y0 = (wg % strips) * R;
  const int tid = threadIdx.x;
  const int ylo = max(y0 - 1, 0), yhi = min(y0 + R + 1, H);             // staged rows [ylo, yhi)
  const int nst = yhi - ylo, rows = min(R, H - y0), npx = rows * W;
  const long p0 = ((long)f * H + y0) * W;                               // first pixel of the strip, frame-linear
  float* cs = zs + zfloats;
  int zlead, clead;
  mk_stage(lp + ((long)f * H + ylo) * W * 3, 3 * nst * W, zs, tid, zlead);
  mk_stage(conf + p0, npx, cs, tid, clead);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float* zz = zs + zlead + 2;                                     // z of staged pixel p: zz[3 p]
  const float* cc = cs + clead;
  int sawnan = 0;
  for (int p = tid; p < nst * W; p += 256) { const float v = zz[3 * p]; sawnan |= v != v; }
  const bool fast = !__syncthreads_or(sawnan) && mf.ok;                 // (the barrier is executed either way)
  const int yoff = y0 - ylo;                                            // staged row of the strip's first row (0 or 1)
  int ry = tid / W, x = tid - ry * W;
  const bool wide = W >= 256;
  for (int e = tid; e < npx; e += 256) {
    const int yc = yoff + ry;
    const float* r1 = zz + 3 * (yc * W);
    const float* r0 = zz + 3 * (max(yc - 1, 0) * W);
    const float* r2 = zz + 3 * (min(yc + 1, nst - 1) * W);
    const int xl = 3 * max(x - 1, 0), xc = 3 * x, xr = 3 * min(x + 1, W - 1);
    MaskWin w;
    w.a0 = r0[xl]; w.a1 = r0[xc]; w.a2 = r0[xr];
    w.b0 = r1[xl]; w.z = r1[xc]; w.b2 = r1[xr];
    w.c0 = r2[xl]; w.c1 = r2[xc]; w.c2 = r2[xr];
    const float cf = cc[e];
    out[p0 + e] = fast ? mask_pixel_fast(w, cf, estar, rtol, mf) : mask_pixel(w, cf, estar, rtol);
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
      ((uintptr_t)local_points & 3) != 0 || ((uintptr_t)conf & 3) != 0) {
    pi3_set_error("pi3_compute_masks: bad arguments (W <= %d, 4-byte aligned maps)", MK_MAXW);
    return PI3_ERR_ARG;
  }
  // environment knobs (A/B only; read once per process like the GEMM / attention knobs) and the per-threshold constants
  // (memoised for the last conf_thr: the chunk creator always passes the same one) - no getenv / bisection / log() on
  // the per-chunk launch path
  struct Knobs { int exact_only, rows; size_t budget; };
  static const Knobs knobs = [] {
    Knobs k;
    k.exact_only = getenv("PI3_MASKS_EXACT_ONLY") ? 1 : 0;
    const char* e = getenv("PI3_MASKS_ROWS");
    k.rows = (e && atoi(e) > 0) ? atoi(e) : 16;
    long kb = MK_LDS_BUDGET / 1024;
    if (const char* b = getenv("PI3_MASKS_LDS_KB")) kb = atol(b);
    kb = kb < 1 ? 1 : (kb > 64 ? 64 : kb);          // [1 KB, 64 KB]: the kernel does not opt into more dynamic LDS
    k.budget = (size_t)kb * 1024;
    return k;
  }();
  struct ThrConsts { float thr, estar, clo, chi; int valid; };
  static thread_local ThrConsts memo = {0.0f, 0.0f, 0.0f, 0.0f, 0};
  if (!memo.valid || memcmp(&memo.thr, &conf_thr, 4) != 0) {
    // sigmoid(c) > conf_thr, i.e. fl(1 / fl(1 + e)) > conf_thr with e = expf(-c): the set of such e is an interval
    // [0, estar]; bisect estar over the bit patterns of the non-negative floats (they order like the values) with the
    // same two correctly rounded fp32 operations the kernel used to do per pixel
    float es = -1.0f;
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
      memcpy(&es, &lo, 4);
    }
    memo.thr = conf_thr;
    memo.estar = es;
    memo.clo = -INFINITY;                          // "never decided without the expf" unless the bracket is sound
    memo.chi = INFINITY;
    if (es > 1e-30f && es < 1e30f) {
      const double c0 = -log((double)es), band = 1e-5 * (1.0 + fabs(c0));
      memo.clo = (float)(c0 - band);
      memo.chi = (float)(c0 + band);
    }
    memo.valid = 1;
  }
  const float estar = memo.estar;
  MaskFast mf;
  mf.ok = (rtol >= 0.0f && rtol < FLT_MAX && !knobs.exact_only) ? 1 : 0;
  mf.rlo = (float)((double)rtol * (1.0 - 1e-5));
  mf.rhi = (float)((double)rtol * (1.0 + 1e-5));
  mf.clo = memo.clo;
  mf.chi = memo.chi;
  // strip height: the tallest whose staged rows + confidences (worst-case 3 floats of lead each) fit the budget
  int R = knobs.rows < H ? knobs.rows : H;
  auto lds_bytes = [&](int r) { return (size_t)(mk_lds_floats(3, 3 * (r + 2) * W) + mk_lds_floats(3, r * W)) * sizeof(float); };
  while (R > 1 && lds_bytes(R) > knobs.budget) --R;
  if (lds_bytes(R) > 64 * 1024) {
    pi3_set_error("pi3_compute_masks: one row strip of W = %d needs %zu bytes of LDS (> 64 KB)", W, lds_bytes(R));
    return PI3_ERR_ARG;
  }
  const int zfloats = mk_lds_floats(3, 3 * (R + 2) * W);
  const long nwg = (long)F * ((H + R - 1) / R);
  hipLaunchKernelGGL(masks_kernel, dim3((unsigned)((nwg + 7) / 8 * 8)), dim3(256), lds_bytes(R), (hipStream_t)stream, conf, local_points,
                     F, H, W, R, zfloats, estar, rtol, mf, masks);
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
