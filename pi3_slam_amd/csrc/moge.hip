// MoGe-2 metric-scale forward (moge/model/v2.py:128-290, moge/model/modules.py:18-254): the kernels that are not
// shared with the pi3 transformer path.  Activations of the conv pyramid live in HBM as NHWC fp32 ("x") plus a bf16
// NHWC staging image per convolution input (channel stride padded to a multiple of 64, pad = 0) so that 3x3
// replicate-padded convolutions run as implicit GEMMs on the MFMA path (pi3_conv3x3 in gemm.hip) and 1x1 convolutions
// as plain pi3_gemm calls.  Everything here is HBM-bound row / pixel work.
#include "common.h"
#include <float.h>

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm statistics (nn.GroupNorm(G, C), modules.py:47-56: G = C/32 'group_norm' or G = 1 'layer_norm').
// x: f32 [B][HW][ldx]; stats: f64 [B][G][2] (sum, sum of squares).
// DETERMINISTIC by construction (no floating-point atomics: an order-dependent last bit here flips a bf16 rounding
// downstream once in a while, and chunk files must not depend on block scheduling):
//   pass 1  a wave walks pixels; lane l owns channels l, l+64, ...; fp32 partials per 64-pixel chunk, fp64 across
//           chunks; the block's four waves are combined in wave order through LDS and the block writes its
//           per-CHANNEL partials to the caller's workspace ws[b][block][2][C];
//   pass 2  one workgroup per sample: thread c sums channel c over the blocks in block order, then one wave per group
//           adds the group's channels in a fixed order (lane-strided, then a fixed shuffle tree).
// ---------------------------------------------------------------------------------------------------------------
#define GN_MAXJ 16  // C <= 1024
#define GN_MAXBLOCKS 512
// NJ = channels per lane (C <= 64 NJ): compile-time so that the pixel loop carries no per-channel branches and four
// pixels' loads are in flight per lane (the first version walked one pixel at a time under 16 predicated channel slots
// and was latency-bound: 185 us for 30 MB)
// HALF (NJ == 1, C <= 32: the 32-channel maps of the finest levels, ~900 k pixels): a wave's row load covers TWO pixels
// (lanes 0-31 an even one, lanes 32-63 the odd one after it) instead of leaving half of its lanes on a clamped
// duplicate; the two halves are added at the end (even + odd, a fixed order).
template <int NJ, bool HALF = false>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, long ldx, int HW, int C,
                                                              double* __restrict__ ws) {
  static_assert(!HALF || NJ == 1, "HALF is the one-channel-per-lane form");
  __shared__ double part[3][2][64 * NJ];   // waves 1..3 hand their per-channel sums to wave 0
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int PP = HALF ? 2 : 1;         // pixels per row load
  const int sub = HALF ? lane >> 5 : 0;
  double s[NJ], q[NJ];
  int ch[NJ];       // clamped channel: loads stay in bounds, lanes past C are dropped at the end
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    s[j] = 0.0; q[j] = 0.0;
    const int c = (HALF ? lane & 31 : lane) + 64 * j;
    ch[j] = c < C ? c : C - 1;
  }
  const int per_block = (HW + gridDim.x - 1) / gridDim.x;
  const int p0 = blockIdx.x * per_block, p1 = min(HW, p0 + per_block);
  const float* xb = x + (long)b * HW * ldx;
  for (int pc = p0 + wave * 64; pc < p1; pc += 256) {   // chunks of 64 pixels per wave: fp32 partials inside a chunk
    float fs[NJ], fq[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { fs[j] = 0.f; fq[j] = 0.f; }
    const int pe = min(p1, pc + 64);
    for (int p = pc; p < pe; p += 4 * PP) {
      float v[4][NJ];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = xb + (long)min(p + PP * u + sub, pe - 1) * ldx;
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[u][j] = row[ch[j]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool live = p + PP * u + sub < pe;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float t = live ? v[u][j] : 0.f;   // pixel by pixel, fp32 in the chunk
          fs[j] += t;
          fq[j] += t * t;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { s[j] += (double)fs[j]; q[j] += (double)fq[j]; }
  }
  if constexpr (HALF) {                    // even-pixel half + odd-pixel half
    s[0] += __shfl_down(s[0], 32, 64);
    q[0] += __shfl_down(q[0], 32, 64);
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      part[wave - 1][0][lane + 64 * j] = s[j];
      part[wave - 1][1][lane + 64 * j] = q[j];
    }
  }
  __syncthreads();
  if (wave == 0) {
    double* o = ws + ((long)b * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = lane + 64 * j;
      if (c < C && (!HALF || lane < 32)) {
        double ss = s[j], qq = q[j];
#pragma unroll
        for (int w = 0; w < 3; ++w) { ss += part[w][0][c]; qq += part[w][1][c]; }   // wave order 0, 1, 2, 3
        o[c] = ss;
        o[C + c] = qq;
      }
    }
  }
}

// One workgroup of 1024 threads per sample.  Column c of [sum | sumsq][channel] is summed over the blocks by P = 1024 /
// columns-per-pass threads (thread part p takes blocks p, p + P, ... in order), the P partial sums are then added in
// part order: a fixed order again, P times shorter than the first version's single chain over up to 512 blocks (100 us
// of one wave's latency per group norm at the finest level).
__global__ __launch_bounds__(1024) void groupnorm_finalize_kernel(const double* __restrict__ ws, int nblk, int C, int G,
                                                                  double* __restrict__ stats) {
  __shared__ double tot[2][64 * GN_MAXJ];
  __shared__ double partial[1024];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* w = ws + (long)b * nblk * 2 * C;
  const int ncol = 2 * C;
  int cp = 1;                                     // columns per pass: the largest power of two <= min(ncol, 1024)
  while (cp * 2 <= ncol && cp * 2 <= 1024) cp *= 2;
  const int P = 1024 / cp;
  for (int c0 = 0; c0 < ncol; c0 += cp) {
    const int c = c0 + (tid & (cp - 1)), part = tid / cp;
    double a = 0.0;
    if (c < ncol)
      for (int k = part; k < nblk; k += P) a += w[(long)k * 2 * C + c];
    partial[tid] = a;
    __syncthreads();
    if (tid < cp && c < ncol) {
      double t = 0.0;
      for (int pp = 0; pp < P; ++pp) t += partial[pp * cp + tid];
      tot[c >= C][c >= C ? c - C : c] = t;
    }
    __syncthreads();
  }
  const int cpg = C / G;
  for (int g = wave; g < G; g += 16) {
    double ss = 0.0, qq = 0.0;
    for (int c = lane; c < cpg; c += 64) { ss += tot[0][g * cpg + c]; qq += tot[1][g * cpg + c]; }
    ss = wave_sum_f64(ss);
    qq = wave_sum_f64(qq);
    if (lane == 0) {
      stats[((long)b * G + g) * 2] = ss;
      stats[((long)b * G + g) * 2 + 1] = qq;
    }
  }
}

static inline int gn_blocks(int HW) {
  int bx = (HW + 511) / 512;
  return bx > GN_MAXBLOCKS ? GN_MAXBLOCKS : bx;
}

extern "C" long pi3_groupnorm_ws_doubles(int B, int HW, int C) { return (long)B * gn_blocks(HW) * 2 * C; }

extern "C" int pi3_groupnorm_stats(const float* x, long ldx, int B, int HW, int C, int G, double* stats, double* ws,
                                   long ws_doubles, void* stream) {
  if (!x || !stats || !ws || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) || C > 64 * GN_MAXJ) {
    pi3_set_error("pi3_groupnorm_stats: bad arguments C=%d G=%d", C, G);
    return PI3_ERR_ARG;
  }
  if (ws_doubles < pi3_groupnorm_ws_doubles(B, HW, C)) {
    pi3_set_error("pi3_groupnorm_stats: workspace of %ld doubles, need %ld", ws_doubles, pi3_groupnorm_ws_doubles(B, HW, C));
    return PI3_ERR_WORKSPACE;
  }
  const int bx = gn_blocks(HW);
  const dim3 grid(bx, B), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 32) hipLaunchKernelGGL((groupnorm_stats_kernel<1, true>), grid, block, 0, st, x, ldx, HW, C, ws);
  else if (C <= 64) hipLaunchKernelGGL(groupnorm_stats_kernel<1>, grid, block, 0, st, x, ldx, HW, C, ws);
  else if (C <= 128) hipLaunchKernelGGL(groupnorm_stats_kernel<2>, grid, block, 0, st, x, ldx, HW, C, ws);
  else if (C <= 256) hipLaunchKernelGGL(groupnorm_stats_kernel<4>, grid, block, 0, st, x, ldx, HW, C, ws);
  else if (C <= 512) hipLaunchKernelGGL(groupnorm_stats_kernel<8>, grid, block, 0, st, x, ldx, HW, C, ws);
  else hipLaunchKernelGGL(groupnorm_stats_kernel<16>, grid, block, 0, st, x, ldx, HW, C, ws);
  hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3(B), dim3(1024), 0, st, ws, bx, C, G, stats);
  return pi3_check_launch("groupnorm_stats");
}

// y = act((x - mean_g) * rstd_g * gamma_c + beta_c) -> bf16 NHWC staging image [B][HW][ldo], channels [C, Cpad) = 0.
// act (modules.py:36-45): 0 none, 2 ReLU (same codes as pi3_gemm), 3 LeakyReLU(0.2), 4 SiLU, 5 ELU(1).
// eps = 1e-5 (nn.GroupNorm / nn.InstanceNorm2d default).  gamma / beta may be null (InstanceNorm2d has no affine
// parameters: G = C); G == 0 means no normalisation at all (in_norm 'none': y = act(x), stats unused).
__device__ __forceinline__ float moge_act(float v, int act) {
  switch (act) {
    case 2: return fmaxf(v, 0.f);
    case 3: return v >= 0.f ? v : 0.2f * v;
    case 4: return v / (1.0f + __expf(-v));
    case 5: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}

// Round 3: the first version did three 64-bit divisions and an fp64 division + sqrt per ELEMENT (208 us for a 32-channel
// map of 905 k pixels, 0.8 TB/s).  Now: mean / rstd per group once per workgroup (same fp64 expressions, same float
// results), four channels per thread (16-byte load, 8-byte store), (pixel, quad) walked with a carry; the grid is sized
// so that a thread's quad never changes and its per-channel constants stay in registers.
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const float* __restrict__ x, long ldx, int HW, int C,
                                                              int Cpad, int G, const double* __restrict__ stats,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, int act,
                                                              bf16_t* __restrict__ out, long ldo, int f16) {
  __shared__ float gmean[64 * GN_MAXJ], grstd[64 * GN_MAXJ];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpg = G > 0 ? C / G : C;
  if (G > 0) {
    const double n = (double)cpg * (double)HW;
    for (int g = tid; g < G; g += 256) {
      const double mean = stats[((long)b * G + g) * 2] / n;
      const double var = stats[((long)b * G + g) * 2 + 1] / n - mean * mean;
      gmean[g] = (float)mean;
      grstd[g] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
    }
  }
  __syncthreads();
  const int q = Cpad >> 2;
  const long stride = (long)gridDim.x * 256;
  const long i0 = (long)blockIdx.x * 256 + tid;
  int pix = (int)(i0 / q), quad = (int)(i0 - (long)pix * q);
  const int dp = (int)(stride / q), dq = (int)(stride - (long)dp * q);      // dq == 0 by the host's choice of grid
  const float* xb = x + (long)b * HW * ldx;
  bf16_t* ob = out + (long)b * HW * ldo;
  const int c = quad * 4;
  const bool vec = c + 3 < C && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0;
  float mu[4], rs[4], ga[4], be[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = c + e;
    const bool live = cc < C;
    mu[e] = (G > 0 && live) ? gmean[cc / cpg] : 0.f;
    rs[e] = (G > 0 && live) ? grstd[cc / cpg] : 1.f;
    ga[e] = (gamma && G > 0 && live) ? gamma[cc] : 1.f;
    be[e] = (gamma && G > 0 && live) ? beta[cc] : 0.f;
  }
  const bool affine = gamma != nullptr && G > 0;
#pragma unroll 4
  for (; pix < HW; pix += dp) {
    const float* row = xb + (long)pix * ldx + c;
    float v[4];
    if (vec) {
      const f32x4 t = *(const f32x4*)row;
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = c + e < C ? row[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (G > 0) {
        t = (t - mu[e]) * rs[e];
        if (affine) t = t * ga[e] + be[e];
      }
      v[e] = c + e < C ? moge_act(t, act) : 0.f;
    }
    u32x2 o;
    o[0] = pack16x2(v[0], v[1], f16);
    o[1] = pack16x2(v[2], v[3], f16);
    *(u32x2*)(ob + (long)pix * ldo + c) = o;
  }
  (void)dq;
}

// x[r][c] += y[r][c] on fp32 maps (ConvStack with an identity input block: x = x + feature, modules.py:245-249)
__global__ __launch_bounds__(256) void add_rows_kernel(float* __restrict__ x, long ldx, const float* __restrict__ y,
                                                       long ldy, long rows, int C) {
  const long total = rows * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    x[r * ldx + c] += y[r * ldy + c];
  }
}

extern "C" int pi3_add_rows(float* x, long ldx, const float* y, long ldy, long rows, int C, void* stream) {
  if (!x || !y || rows <= 0 || C <= 0) {
    pi3_set_error("pi3_add_rows: bad arguments");
    return PI3_ERR_ARG;
  }
  long blocks = (rows * C + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, C);
  return pi3_check_launch("add_rows");
}

extern "C" int pi3_groupnorm_apply(const float* x, long ldx, int B, int HW, int C, int Cpad, int G,
                                   const double* stats, const float* gamma, const float* beta, float eps, int act,
                                   void* out, long ldo, int out_dtype, void* stream) {
  if ((out_dtype != 0 && out_dtype != 2) || !x || (G > 0 && !stats) || ((gamma != nullptr) != (beta != nullptr)) || !out || C <= 0 || Cpad < C ||
      (Cpad % 4) || (ldo % 4) || ((uintptr_t)out & 7) || G < 0 || (G > 0 && (C % G)) || G > 64 * GN_MAXJ || act < 0 ||
      act > 5 || act == 1 || B <= 0 || HW <= 0) {
    pi3_set_error("pi3_groupnorm_apply: bad arguments C=%d Cpad=%d G=%d act=%d (Cpad, ldo multiples of 4)", C, Cpad, G, act);
    return PI3_ERR_ARG;
  }
  // threads of the grid = a multiple of the quads per pixel: every thread keeps its quad
  const int q = Cpad / 4;
  int a = 256, bq = q;
  while (bq) { const int t = a % bq; a = bq; bq = t; }      // a = gcd(256, q)
  const long unit = q / a;                                   // blocks must be a multiple of this
  long blocks = ((long)HW * q + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  blocks = (blocks + unit - 1) / unit * unit;
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3((unsigned)blocks, (unsigned)B), dim3(256), 0, (hipStream_t)stream, x,
                     ldx, HW, C, Cpad, G, stats, gamma, beta, eps, act, (bf16_t*)out, ldo, out_dtype == 2);
  return pi3_check_launch("groupnorm_apply");
}

// ConvTranspose2d(k=2, s=2) second half (modules.py:160-163): the GEMM produced g[pixel][(dy*2+dx)*Cs + co] (f32, bias
// included); scatter to the 2x up-sampled bf16 NHWC staging image out[b][2y+dy][2x+dx][co], pad channels = 0.
// One workgroup per output row, four channels per thread (16-byte load, 8-byte store), 32-bit index arithmetic (the first
// version: one element pair per thread behind five 64-bit divisions).
__global__ __launch_bounds__(256) void convt_scatter_kernel(const float* __restrict__ g, long ldg, int H, int W,
                                                            int Cout, int Cs, int Cpad, bf16_t* __restrict__ out,
                                                            long ldo, int f16) {
  const int orow = blockIdx.x;                 // over [B][2H]
  const int oy = orow % (2 * H), b = orow / (2 * H);
  const int q = Cpad >> 2, n = 2 * W * q;
  const float* grow = g + ((long)b * H + (oy >> 1)) * W * ldg + (long)(oy & 1) * 2 * Cs;
  bf16_t* orowp = out + (long)orow * 2 * W * ldo;
  const bool vec = (ldg & 3) == 0 && (Cs & 3) == 0 && ((uintptr_t)g & 15) == 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int ox = i / q, c = (i - ox * q) * 4;
    const float* src = grow + (long)(ox >> 1) * ldg + (ox & 1) * Cs + c;
    float v[4];
    if (vec && c + 3 < Cout) {
      const f32x4 t = *(const f32x4*)src;
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = c + e < Cout ? src[e] : 0.f;
    }
    u32x2 o;
    o[0] = pack16x2(v[0], v[1], f16);
    o[1] = pack16x2(v[2], v[3], f16);
    *(u32x2*)(orowp + (long)ox * ldo + c) = o;
  }
}

extern "C" int pi3_convt_scatter(const float* g, long ldg, int B, int H, int W, int Cout, int Cs, int Cpad, void* out,
                                 long ldo, int out_dtype, void* stream) {
  if ((out_dtype != 0 && out_dtype != 2) || !g || !out || B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cs < Cout || Cpad < Cout || (Cpad % 4) || (ldo % 4) ||
      ((uintptr_t)out & 7) || (long)B * 2 * H > 0x7fffffffL || (long)2 * W * (Cpad / 4) > 0x7fffffffL) {
    pi3_set_error("pi3_convt_scatter: bad arguments (Cpad, ldo multiples of 4)");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(convt_scatter_kernel, dim3((unsigned)(B * 2 * H)), dim3(256), 0, (hipStream_t)stream, g, ldg, H, W,
                     Cout, Cs, Cpad, (bf16_t*)out, ldo, out_dtype == 2);
  return pi3_check_launch("convt_scatter");
}

// 1x1 convolution of the 2-channel UV map (v2.py:141-147 concat + modules.py:214 input block), fused:
// x[b][y][x][c] (+)= w[c][wofs] * u[x] + w[c][wofs+1] * v[y] + bias[c];  w row stride ldw.  accumulate != 0 adds.
// blockIdx.x = image row (v[y] is then a workgroup constant), blockIdx.y = a segment of 1024 float4 of the row: four
// independent float4 per thread (the loads of an accumulate pass in flight together), 32-bit index arithmetic, the
// thread's four channels and their weights fixed across its iterations whenever C / 4 divides 256.
#define UV_SEG 1024
__global__ __launch_bounds__(256) void uv_affine_kernel(float* __restrict__ x, long ldx, int H, int W, int C,
                                                        const float* __restrict__ w, long ldw, int wofs,
                                                        const float* __restrict__ bias, const float* __restrict__ uvx,
                                                        const float* __restrict__ uvy, int accumulate, int vec) {
  const int row = blockIdx.x;                  // over [B][H]
  const float vy = uvy[row % H];
  float* xr = x + (long)row * W * ldx;
  if (vec) {
    const int q = C >> 2, n4 = W * q;
    const int i0 = blockIdx.y * UV_SEG + threadIdx.x, i1 = min(n4, (int)(blockIdx.y + 1) * UV_SEG);
    const bool fixed = (256 % q) == 0;
    f32x4 wa, wb, bs;
    auto load_w = [&](int c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wa[e] = w[(long)(c + e) * ldw + wofs];
        wb[e] = w[(long)(c + e) * ldw + wofs + 1];
        bs[e] = bias ? bias[c + e] : 0.f;
      }
    };
    if (fixed) load_w((i0 % q) * 4);
#pragma unroll
    for (int k = 0; k < UV_SEG / 256; ++k) {
      const int i = i0 + k * 256;
      if (i < i1) {
        const int px = i / q, c = (i - px * q) * 4;
        if (!fixed) load_w(c);
        const float ux = uvx[px];
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = wa[e] * ux + wb[e] * vy;
          if (bias) v[e] += bs[e];
        }
        f32x4* dst = (f32x4*)(xr + (long)px * ldx + c);
        if (accumulate) {
          const f32x4 o = *dst;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[e] + v[e];
        }
        *dst = v;
      }
    }
    return;
  }
  const int n = W * C;
  for (int i = blockIdx.y * (4 * UV_SEG) + threadIdx.x; i < min(n, (int)(blockIdx.y + 1) * 4 * UV_SEG); i += 256) {
    const int px = i / C, c = i - px * C;
    float v = w[(long)c * ldw + wofs] * uvx[px] + w[(long)c * ldw + wofs + 1] * vy;
    if (bias) v += bias[c];
    float* dst = xr + (long)px * ldx + c;
    *dst = accumulate ? (*dst + v) : v;
  }
}

extern "C" int pi3_uv_affine(float* x, long ldx, int B, int H, int W, int C, const float* w, long ldw, int wofs,
                             const float* bias, const float* uvx, const float* uvy, int accumulate, void* stream) {
  if (!x || !w || !uvx || !uvy || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (long)B * H > 0x7fffffffL ||
      (long)W * C > 0x3fffffffL) {
    pi3_set_error("pi3_uv_affine: bad arguments");
    return PI3_ERR_ARG;
  }
  const int vec = (C % 4) == 0 && (ldx % 4) == 0 && ((uintptr_t)x & 15) == 0;
  const int segs = (W * C + 4 * UV_SEG - 1) / (4 * UV_SEG);     // 4 UV_SEG floats per segment in either form
  hipLaunchKernelGGL(uv_affine_kernel, dim3((unsigned)(B * H), (unsigned)segs), dim3(256), 0, (hipStream_t)stream, x, ldx, H,
                     W, C, w, ldw, wofs, bias, uvx, uvy, accumulate, vec);
  return pi3_check_launch("uv_affine");
}

// Separable resize with host-built tap tables (F.interpolate bilinear, antialias or not: modules.py:121, v2.py:157).
// src element (c, y, x) at src[c*sc + y*sy + x*sx]; dst likewise.  Table per output index: start, count, weights[MT].
#define RS_MT 8
__global__ __launch_bounds__(256) void resize_taps_kernel(const float* __restrict__ src, long sc, long sy, long sx,
                                                          int C, const int* __restrict__ ys,
                                                          const float* __restrict__ yw, const int* __restrict__ xs,
                                                          const float* __restrict__ xw, int oh, int ow,
                                                          float* __restrict__ dst, long dc, long dy, long dx,
                                                          long total) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % ow);
    const int oy = (int)((i / ow) % oh);
    const int c = (int)(i / ((long)ow * oh));
    const int y0 = ys[2 * oy], yn = ys[2 * oy + 1], x0 = xs[2 * ox], xn = xs[2 * ox + 1];
    float acc = 0.f;
    for (int a = 0; a < yn; ++a) {
      float rowacc = 0.f;
      const float* row = src + (long)c * sc + (long)(y0 + a) * sy;
      for (int b = 0; b < xn; ++b) rowacc += xw[ox * RS_MT + b] * row[(long)(x0 + b) * sx];
      acc += yw[oy * RS_MT + a] * rowacc;
    }
    dst[(long)c * dc + (long)oy * dy + (long)ox * dx] = acc;
  }
}

extern "C" int pi3_resize_taps(const float* src, long sc, long sy, long sx, int C, const int* ys, const float* yw,
                               const int* xs, const float* xw, int oh, int ow, float* dst, long dc, long dy, long dx,
                               void* stream) {
  if (!src || !ys || !yw || !xs || !xw || !dst || C <= 0 || oh <= 0 || ow <= 0) {
    pi3_set_error("pi3_resize_taps: bad arguments");
    return PI3_ERR_ARG;
  }
  const long total = (long)C * oh * ow;
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(resize_taps_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, sc, sy, sx, C,
                     ys, yw, xs, xw, oh, ow, dst, dc, dy, dx, total);
  return pi3_check_launch("resize_taps");
}

// y = act(W x + b) for one vector (scale_head MLP on the class token, modules.py:184-192).  One workgroup.
__global__ __launch_bounds__(256) void dense_vec_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                        const float* __restrict__ b, int K, int N, int act,
                                                        float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = wave; j < N; j += 4) {
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += Wt[(long)j * K + k] * x[k];
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = acc + (b ? b[j] : 0.f);
      y[j] = act == 2 ? fmaxf(v, 0.f) : v;
    }
  }
}

extern "C" int pi3_dense_vec(const float* x, const float* Wt, const float* b, int K, int N, int act, float* y,
                             void* stream) {
  if (!x || !Wt || !y || K <= 0 || N <= 0) {
    pi3_set_error("pi3_dense_vec: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(dense_vec_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, Wt, b, K, N, act, y);
  return pi3_check_launch("dense_vec");
}

// v2.py:160-170 on the resized maps: points (HW x 3 f32, in place) remapped ('linear' 0, 'exp' 1, 'sinh' 2,
// 'sinh_exp' 3), mask logit -> sigmoid -> binary (> 0.5) uint8.
__global__ __launch_bounds__(256) void moge_remap_kernel(float* __restrict__ pts, const float* __restrict__ mlogit,
                                                         long n, int remap, uint8_t* __restrict__ mask) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  if (remap == 1) { z = expf(z); x *= z; y *= z; }
  else if (remap == 2) { x = sinhf(x); y = sinhf(y); z = sinhf(z); }
  else if (remap == 3) { x = sinhf(x); y = sinhf(y); z = expf(z); }
  pts[3 * i] = x; pts[3 * i + 1] = y; pts[3 * i + 2] = z;
  if (mask) mask[i] = mlogit ? ((1.0f / (1.0f + expf(-mlogit[i]))) > 0.5f ? 1 : 0) : 1;
}

extern "C" int pi3_moge_remap(float* pts, const float* mask_logit, long n, int remap, unsigned char* mask,
                              void* stream) {
  if (!pts || n <= 0 || remap < 0 || remap > 3) {
    pi3_set_error("pi3_moge_remap: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(moge_remap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
                     mask_logit, n, remap, mask);
  return pi3_check_launch("moge_remap");
}

// v2.py:255-274: depth = (z + shift); mask &= depth > 0; depth *= metric_scale; depth = mask ? depth : +inf.
// shift, log_scale: device scalars (log_scale = scale_head output before exp, or NULL).
__global__ __launch_bounds__(256) void moge_depth_kernel(const float* __restrict__ pts, const float* __restrict__ shift,
                                                         const float* __restrict__ log_scale, uint8_t* __restrict__ mask,
                                                         long n, float* __restrict__ depth) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d = pts[3 * i + 2] + shift[0];
  const bool ok = mask[i] && d > 0.f;
  mask[i] = ok ? 1 : 0;
  if (log_scale) d *= expf(log_scale[0]);
  depth[i] = ok ? d : INFINITY;
}

extern "C" int pi3_moge_depth(const float* pts, const float* shift, const float* log_scale, unsigned char* mask, long n,
                              float* depth, void* stream) {
  if (!pts || !shift || !mask || !depth || n <= 0) {
    pi3_set_error("pi3_moge_depth: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(moge_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
                     shift, log_scale, mask, n, depth);
  return pi3_check_launch("moge_depth");
}
