// HBM-bound row kernels of the transformer blocks: LayerNorm, per-head q/k LayerNorm + RoPE-2D, dtype casts,
// patch gather (im2col) and the positional-embedding resample.  All are one-pass, 16-byte-per-lane vectorised.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (eps inside the sqrt), fp32 in, bf16 or fp32 out.  One wave per row.
// reference: nn.LayerNorm(eps=1e-6) at pi3/models/layers/block.py:282,296 and dinov2/layers/block.py:63,75,
// final encoder norm pi3/models/dinov2/models/vision_transformer.py:271.
// Optional "special rows": rows whose (row % T) < nspecial are replaced by special[row % T][:] (fp32 out only) —
// that is Pi3.decode's register-token concat (pi3/models/pi3.py:140-144) fused with the encoder's final norm.
// ---------------------------------------------------------------------------------------------------------------
template <bool OUT_BF16, int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long ldx, int rows, int D,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float eps, void* out, long ldo, int T, int nspecial,
                                                        const float* __restrict__ special, int f16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 2;  // float4 count; VPL = ceil(nv / 64) vectors per lane, loads are clamped not branched
  if (nspecial > 0) {
    const int t = row % T;
    if (t < nspecial) {
      for (int c = lane; c < nv; c += 64)
        *(f32x4*)((float*)out + (long)row * ldo + 4 * c) = *(const f32x4*)(special + (long)t * D + 4 * c);
      return;
    }
  }
  const float* xr = x + (long)row * ldx;
  f32x4 v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = min(lane + 64 * i, nv - 1);
    v[i] = *(const f32x4*)(xr + 4 * c);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float part = (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    s += (lane + 64 * i < nv) ? part : 0.f;
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    v[i] -= mean;
    const float part = (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    q += (lane + 64 * i < nv) ? part : 0.f;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    const int cc = min(c, nv - 1);
    const f32x4 ww = *(const f32x4*)(w + 4 * cc);
    const f32x4 bb = *(const f32x4*)(b + 4 * cc);
    const f32x4 y = v[i] * rstd * ww + bb;
    if (c < nv) {
      if constexpr (OUT_BF16) {
        u32x2 o;
        o[0] = pack16x2(y[0], y[1], f16);
        o[1] = pack16x2(y[2], y[3], f16);
        *(u32x2*)((bf16_t*)out + (long)row * ldo + 4 * c) = o;
      } else {
        *(f32x4*)((float*)out + (long)row * ldo + 4 * c) = y;
      }
    }
  }
}

extern "C" int pi3_layernorm(const float* x, long ldx, int rows, int D, const float* w, const float* b, float eps,
                             void* out, long ldo, int out_dtype, int T, int nspecial, const float* special,
                             void* stream) {
  if (!x || !w || !b || !out || rows <= 0 || D <= 0 || (D % 4) || D > 2048 || (ldx % 4) || (ldo % 4) ||
      (nspecial > 0 && (out_dtype != 1 || !special || T <= 0)) || out_dtype < 0 || out_dtype > 2) {
    pi3_set_error("pi3_layernorm: bad arguments rows=%d D=%d (D %% 4 == 0, D <= 2048) out_dtype=%d", rows, D, out_dtype);
    return PI3_ERR_ARG;
  }
  const int f16 = out_dtype == 2;      // IEEE-half output: the bf16 instance with the other conversion
  if (f16) out_dtype = 0;
  const dim3 grid((rows + 3) / 4), block(256);
  const int vpl = (D / 4 + 63) / 64;
#define LN_LAUNCH(OB, V)                                                                                         \
  hipLaunchKernelGGL((layernorm_kernel<OB, V>), grid, block, 0, (hipStream_t)stream, x, ldx, rows, D, w, b, eps, \
                     out, ldo, T, nspecial, special, f16)
  if (out_dtype == 0) {
    if (vpl <= 1) LN_LAUNCH(true, 1); else if (vpl <= 2) LN_LAUNCH(true, 2);
    else if (vpl <= 4) LN_LAUNCH(true, 4); else LN_LAUNCH(true, 8);
  } else {
    if (vpl <= 1) LN_LAUNCH(false, 1); else if (vpl <= 2) LN_LAUNCH(false, 2);
    else if (vpl <= 4) LN_LAUNCH(false, 4); else LN_LAUNCH(false, 8);
  }
#undef LN_LAUNCH
  return pi3_check_launch("layernorm");
}

// ---------------------------------------------------------------------------------------------------------------
// In-place per-head LayerNorm(64) on q and k (optional) + RoPE-2D + softmax-scale fold on q, on a packed
// [rows][3][H][64] bf16 qkv buffer.
// reference: FlashAttentionRope.forward pi3/models/layers/attention.py:330-334; the rotation follows
// pi3/models/curope/curope.cpp:11-47 == RoPE2D.forward pi3/models/layers/pos_embed.py:142-159 evaluated in fp32:
// first 32 dims rotate by the token's y position, last 32 by x; inside a 32-dim half, dim j pairs with j+16 and
// both use inv_freq[j % 16] = base^(-(j % 16)/16).  cs is the host-built fp32 table [npos][16][2] = (cos, sin).
// 8 lanes per 64-vector (8 dims each): LayerNorm reduces over 8 lanes; the RoPE partner (dim +-16) is lane ^ 2.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qknorm_rope_kernel(bf16_t* qkv, long rows, int H, int T,
                                                          const int* __restrict__ pos,      // [T][2] (y, x)
                                                          const float* __restrict__ cs,     // [npos][16][2]
                                                          const float* __restrict__ qw, const float* __restrict__ qb,
                                                          const float* __restrict__ kw, const float* __restrict__ kb,
                                                          float eps, float qscale, int do_rope) {
  const int sub = threadIdx.x & 7;
  const long nvec = rows * 2 * H;  // q and k vectors
  const long stride = (long)gridDim.x * 32;
  for (long vi = (long)blockIdx.x * 32 + (threadIdx.x >> 3); vi < nvec; vi += stride) {
    const long row = vi / (2 * H);
    const int rem = (int)(vi - row * 2 * H);
    const int part = rem / H;  // 0 = q, 1 = k
    bf16_t* ptr = qkv + row * (3L * H * 64) + (long)rem * 64 + sub * 8;
    const u32x4 raw = *(const u32x4*)ptr;
    float x[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      x[2 * e] = __uint_as_float(raw[e] << 16);
      x[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u);
    }
    const float* nw = part ? kw : qw;
    const float* nb = part ? kb : qb;
    if (nw) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += x[e];
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      const float mean = s * (1.0f / 64.0f);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        x[e] -= mean;
        q += x[e] * x[e];
      }
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      q += __shfl_xor(q, 4, 64);
      const float rstd = rsqrtf(q * (1.0f / 64.0f) + eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = x[e] * rstd * nw[sub * 8 + e] + nb[sub * 8 + e];
    }
    if (do_rope) {
      const int t = (int)(row % T);
      const int pp = pos[2 * t + (sub >> 2)];      // y for dims 0..31, x for dims 32..63
      const float* tab = cs + ((long)pp * 16 + 8 * (sub & 1)) * 2;
      const float sgn = (sub & 2) ? 1.f : -1.f;    // j < 16: x*cos - partner*sin ; j >= 16: x*cos + partner*sin
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float partner = __shfl_xor(x[e], 2, 64);
        const float c = tab[2 * e], sn = tab[2 * e + 1];
        x[e] = x[e] * c + sgn * partner * sn;
      }
    }
    const float sc = part ? 1.0f : qscale;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack_bf16x2(x[2 * e] * sc, x[2 * e + 1] * sc);
    *(u32x4*)ptr = o;
  }
}

int pi3_qknorm_rope_launch(void* qkv, long rows, int H, int T, const int* pos, const float* cs, const float* qw,
                           const float* qb, const float* kw, const float* kb, float eps, float qscale, int do_rope,
                           hipStream_t stream) {
  const long nvec = rows * 2 * H;
  long blocks = (nvec + 31) / 32;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(qknorm_rope_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (bf16_t*)qkv, rows, H, T, pos,
                     cs, qw, qb, kw, kb, eps, qscale, do_rope);
  return pi3_check_launch("qknorm_rope");
}

extern "C" int pi3_qknorm_rope(void* qkv, long rows, int H, int T, const int* pos, const float* cs, const float* qw,
                               const float* qb, const float* kw, const float* kb, float eps, float qscale,
                               int do_rope, void* stream) {
  if (!qkv || rows <= 0 || H <= 0 || T <= 0 || (do_rope && (!pos || !cs)) || ((uintptr_t)qkv & 15) ||
      ((qw != nullptr) != (kw != nullptr))) {
    pi3_set_error("pi3_qknorm_rope: bad arguments rows=%ld H=%d T=%d", rows, H, T);
    return PI3_ERR_ARG;
  }
  return pi3_qknorm_rope_launch(qkv, rows, H, T, pos, cs, qw, qb, kw, kb, eps, qscale, do_rope, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// pi3_rope_2d: the reference's ONE native FFI entry, `curope.rope_2d(tokens, positions, base, fwd)`
// (pi3/models/curope/curope.cpp:49-68; device kernel kernels.cu:17-82; launch checks :85-108), with its contract:
// in place on tokens (B, N, H, D), element (b, n, h, d) at tok[b*stride_b + n*stride_n + h*stride_h + d].  The reference's
// kernel walks a PackedTensorAccessor (all four strides free) but its launch check demands stride(3) == 1 and
// stride(2) == D (kernels.cu:91) - true for q / k slices of a packed qkv, NOT for the transposed view of a contiguous
// (B, heads, N, D) tensor, which is what `cuRoPE2D.forward` passes after pi3's q_norm / k_norm (attention.py:330-334):
// there the CUDA original raises.  This entry takes the three outer strides as they are (stride(3) == 1 only), so it
// is a superset of what the reference accepts.  positions int64 (B, N, 2) =
// (y, x) contiguous, D % 4 == 0.  One token vector = [u_Y (Q) | v_Y (Q) | u_X (Q) | v_X (Q)], Q = D/4:
//   freq = pos * (fwd / base^(d/Q));  u' = u cos - v sin;  v' = v cos + u sin        (fp32 arithmetic, any storage)
// fwd = F0 forward, -F0 for the backward pass (curope2d.py:24-29), i.e. the inverse rotation.
// One thread per (token, axis, d): sincos once, then the H heads (kernels.cu:64-79 walks the heads the same way);
// a wave covers 64 / (D/2) consecutive tokens, so with contiguous tokens its loads are whole 128-byte lines.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope2d_kernel(T* __restrict__ tok, const long* __restrict__ pos, long ntok, int N,
                                                     int H, int D, long stride_b, long stride_n, long stride_h, float base,
                                                     float fwd) {
  const int half = D >> 1, Q = D >> 2;
  const long g = (long)blockIdx.x * 256 + threadIdx.x;
  const long token = g / half;
  if (token >= ntok) return;
  const int pr = (int)(g - token * half), X = pr / Q, d = pr - X * Q;
  const long b = token / N, n = token - b * N;
  const float inv_freq = fwd / powf(base, (float)d / (float)Q);          // kernels.cu:44
  const float freq = (float)pos[token * 2 + X] * inv_freq;               // kernels.cu:52
  float sn, cs;
  sincosf(freq, &sn, &cs);
  T* p = tok + b * stride_b + n * stride_n + X * 2 * Q + d;
  for (int h = 0; h < H; ++h, p += stride_h) {
    const float u = (float)p[0], v = (float)p[Q];
    p[0] = (T)(u * cs - v * sn);
    p[Q] = (T)(v * cs + u * sn);
  }
}

extern "C" int pi3_rope_2d(void* tokens, const long* positions, int B, int N, int H, int D, long stride_b, long stride_n,
                           long stride_h, float base, float fwd, int dtype, void* stream) {
  if (stride_h == 0) stride_h = D;
  if (stride_n == 0) stride_n = (long)H * stride_h;
  if (stride_b == 0) stride_b = (long)N * stride_n;
  // the reference's TORCH_CHECKs (curope.cpp:54-59, kernels.cu:91-94) as argument errors
  if (!tokens || !positions || B <= 0 || N <= 0 || H <= 0 || D <= 0 || (D & 3) || stride_h < D || stride_n <= 0 ||
      stride_b <= 0 || !(base > 0.0f) || dtype < 0 || dtype > 2) {
    pi3_set_error("pi3_rope_2d: bad arguments B=%d N=%d H=%d D=%d (token dim must be multiple of 4) stride_b=%ld "
                  "stride_n=%ld stride_h=%ld base=%g dtype=%d", B, N, H, D, stride_b, stride_n, stride_h, (double)base, dtype);
    return PI3_ERR_ARG;
  }
  const long ntok = (long)B * N, work = ntok * (D >> 1);
  const unsigned blocks = (unsigned)((work + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(rope2d_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (bf16_t*)tokens, positions, ntok, N, H, D,
                       stride_b, stride_n, stride_h, base, fwd);
  else if (dtype == 1)
    hipLaunchKernelGGL(rope2d_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)tokens, positions, ntok, N, H, D,
                       stride_b, stride_n, stride_h, base, fwd);
  else
    hipLaunchKernelGGL(rope2d_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, (_Float16*)tokens, positions, ntok, N, H,
                       D, stride_b, stride_n, stride_h, base, fwd);
  return pi3_check_launch("rope_2d");
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 -> bf16 / fp32 strided row copy (concat of the last two decoder block outputs, pi3/models/pi3.py:168-171,
// and weight conversion).  cols % 4 == 0.
// ---------------------------------------------------------------------------------------------------------------
// Walks (row, quad) without a 64-bit division per element (the first version did `i / nv` on longs per float4 and ran
// at 1.7-2.6 TB/s): one division per thread, then row += dr, quad += dq with a carry.  in_cols <= cols: columns at and
// beyond in_cols are written as zeros (K padding of a following GEMM) and are not read.
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ in, long ldi, int in_cols, void* out,
                                                        long ldo, long rows, int cols, int f16) {
  const int nv = cols >> 2;
  const long stride = (long)gridDim.x * 256;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  long r = i0 / nv;
  int c = (int)(i0 - r * nv);
  const long dr = stride / nv;
  const int dq = (int)(stride - dr * nv);
#pragma unroll 4
  for (; r < rows;) {
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (4 * c < in_cols) v = *(const f32x4*)(in + r * ldi + 4 * c);
    if constexpr (OUT_BF16) {
      u32x2 o;
      o[0] = pack16x2(v[0], v[1], f16);
      o[1] = pack16x2(v[2], v[3], f16);
      *(u32x2*)((bf16_t*)out + r * ldo + 4 * c) = o;
    } else {
      *(f32x4*)((float*)out + r * ldo + 4 * c) = v;
    }
    r += dr;
    c += dq;
    if (c >= nv) { c -= nv; ++r; }
  }
}

extern "C" int pi3_cast_rows_pad(const float* in, long ldi, int in_cols, void* out, long ldo, long rows, int cols,
                                 int out_dtype, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || in_cols <= 0 || in_cols > cols || (cols % 4) || (in_cols % 4) ||
      (ldi % 4) || (ldo % 4) || ((uintptr_t)in & 15) || ((uintptr_t)out & (out_dtype != 1 ? 7 : 15)) || out_dtype < 0 ||
      out_dtype > 2) {
    pi3_set_error("pi3_cast_rows: bad arguments rows=%ld cols=%d in_cols=%d (multiples of 4, 16-byte aligned)", rows, cols,
                  in_cols);
    return PI3_ERR_ARG;
  }
  long blocks = (rows * (cols >> 2) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (out_dtype != 1)      // 0 bf16, 2 IEEE half
    hipLaunchKernelGGL(cast_rows_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldi,
                       in_cols, out, ldo, rows, cols, out_dtype == 2);
  else
    hipLaunchKernelGGL(cast_rows_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldi,
                       in_cols, out, ldo, rows, cols, 0);
  return pi3_check_launch("cast_rows");
}

extern "C" int pi3_cast_rows(const float* in, long ldi, void* out, long ldo, long rows, int cols, int out_dtype,
                             void* stream) {
  return pi3_cast_rows_pad(in, ldi, cols, out, ldo, rows, cols, out_dtype, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Patch gather: frames fp32 [F][3][H][W] in [0,1] -> ImageNet-normalised bf16 patch rows [F*P][KP] (KP >= 588, zero
// padded), column c*196 + ky*14 + kx == the flattened Conv2d(3,1024,14,14) weight index, so the conv of
// pi3/models/dinov2/layers/patch_embed.py:65,75 becomes pi3_gemm on these rows.  Normalisation = pi3.py:174.
// Round 5 (LDS-staged tiles, as the north star prescribes): a workgroup owns one row of up to 32 patches of a frame:
// the 3 x 14 image rows of the segment are read as whole coalesced rows into LDS, normalised there, and every patch's
// 640-element output row is then written as one contiguous 1 280-byte run (4 bytes per lane).
// (The first form - one thread per (patch, channel, ky), 56-byte reads and 28-byte writes - ran at 3.3 TB/s.)
// ---------------------------------------------------------------------------------------------------------------
#define PG_SEG 8
__global__ __launch_bounds__(256) void patch_gather_kernel(const float* __restrict__ img, int F, int H, int W,
                                                           bf16_t* __restrict__ out, int KP, float m0, float m1,
                                                           float m2, float is0, float is1, float is2, int f16) {
  extern __shared__ __attribute__((aligned(16))) float pg_lds[];      // [c][ky][x in segment], 18 KB
  const int pw = W / 14, ph = H / 14;
  const int nseg = (pw + PG_SEG - 1) / PG_SEG;
  const int seg = blockIdx.x % nseg;
  const int py = (blockIdx.x / nseg) % ph;
  const int f = blockIdx.x / (nseg * ph);
  const int px0 = seg * PG_SEG, np = min(PG_SEG, pw - px0);
  const int wseg = np * 14, w2 = wseg >> 1;          // 14 np is even: the segment's rows are read as float2 (rows are 8-byte aligned: W even)
  const bool vec2 = ((uintptr_t)img & 7) == 0;
  for (int i = threadIdx.x; i < 42 * w2; i += 256) {
    const int row = i / w2, x = 2 * (i - row * w2);     // row = c * 14 + ky
    const int c = row / 14, ky = row - c * 14;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), istd = c == 0 ? is0 : (c == 1 ? is1 : is2);
    const float* src = img + (((long)f * 3 + c) * H + (py * 14 + ky)) * W + px0 * 14 + x;
    f32x2 v;
    if (vec2) v = *(const f32x2*)src;
    else v = (f32x2){src[0], src[1]};
    *(f32x2*)(pg_lds + row * (PG_SEG * 14) + x) = (f32x2){(v[0] - mean) * istd, (v[1] - mean) * istd};
  }
  __syncthreads();
  // output: one thread per (patch, c, ky) run of 14 elements = 7 packed words (28 bytes, 4-byte aligned), then the K padding
  for (int i = threadIdx.x; i < np * 42; i += 256) {
    const int p = i / 42, row = i - p * 42;
    const float* src = pg_lds + row * (PG_SEG * 14) + p * 14;
    bf16_t* dst = out + ((long)f * ph * pw + (long)py * pw + px0 + p) * KP + row * 14;
#pragma unroll
    for (int kx = 0; kx < 14; kx += 2) *(uint32_t*)(dst + kx) = pack16x2(src[kx], src[kx + 1], f16);
  }
  for (int i = threadIdx.x; i < np * ((KP - 588) >> 1); i += 256) {
    const int per = (KP - 588) >> 1, p = i / per, k = 588 + 2 * (i - p * per);
    *(uint32_t*)(out + ((long)f * ph * pw + (long)py * pw + px0 + p) * KP + k) = 0u;
  }
}

extern "C" int pi3_patch_gather(const float* img, int F, int H, int W, void* out, int KP, int out_dtype,
                                const float* mean3, const float* std3, void* stream) {
  if (!img || !out || F <= 0 || H <= 0 || W <= 0 || (H % 14) || (W % 14) || KP < 588 || (KP % 2) ||
      (out_dtype != 0 && out_dtype != 2)) {
    pi3_set_error("pi3_patch_gather: bad arguments F=%d H=%d W=%d KP=%d (H, W multiples of 14)", F, H, W, KP);
    return PI3_ERR_ARG;
  }
  const int pw = W / 14, ph = H / 14, nseg = (pw + PG_SEG - 1) / PG_SEG;
  const long nwg = (long)F * ph * nseg;
  if (nwg > 0x7fffffffL) {
    pi3_set_error("pi3_patch_gather: grid too large");
    return PI3_ERR_ARG;
  }
  static unsigned long long optin = 0;
  if (int rc = pi3_lds_optin((const void*)patch_gather_kernel, (int)sizeof(float) * 3 * 14 * PG_SEG * 14, &optin, "patch_gather")) return rc;
  hipLaunchKernelGGL(patch_gather_kernel, dim3((unsigned)nwg), dim3(256), sizeof(float) * 3 * 14 * PG_SEG * 14, (hipStream_t)stream, img, F, H, W,
                     (bf16_t*)out, KP, mean3[0], mean3[1], mean3[2], 1.0f / std3[0], 1.0f / std3[1], 1.0f / std3[2],
                     out_dtype == 2);
  return pi3_check_launch("patch_gather");
}

// ---------------------------------------------------------------------------------------------------------------
// Separable resample of the [Mi][Mj][D] positional-embedding grid to [oh][ow][D] with host-built tap matrices
// wy [oh][Mi], wx [ow][Mj] (bicubic, antialias, a = -0.5: the weights F.interpolate(mode="bicubic",
// antialias=True) uses at pi3/models/dinov2/models/vision_transformer.py:205-210).  Runs once per (H, W).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resample_grid_kernel(const float* __restrict__ src, int Mi, int Mj, int D,
                                                            const float* __restrict__ wy, const float* __restrict__ wx,
                                                            int oh, int ow, float* __restrict__ dst) {
  const int oy = blockIdx.x / ow, ox = blockIdx.x - oy * ow;
  for (int d = threadIdx.x; d < D; d += 256) {
    float acc = 0.f;
    for (int i = 0; i < Mi; ++i) {
      const float a = wy[oy * Mi + i];
      if (a == 0.f) continue;
      float rowacc = 0.f;
      for (int j = 0; j < Mj; ++j) {
        const float bwt = wx[ox * Mj + j];
        if (bwt != 0.f) rowacc += bwt * src[((long)i * Mj + j) * D + d];
      }
      acc += a * rowacc;
    }
    dst[((long)oy * ow + ox) * D + d] = acc;
  }
}

extern "C" int pi3_resample_grid(const float* src, int Mi, int Mj, int D, const float* wy, const float* wx, int oh,
                                 int ow, float* dst, void* stream) {
  if (!src || !wy || !wx || !dst || Mi <= 0 || Mj <= 0 || D <= 0 || oh <= 0 || ow <= 0) {
    pi3_set_error("pi3_resample_grid: bad arguments");
    return PI3_ERR_ARG;
  }
  hipLaunchKernelGGL(resample_grid_kernel, dim3(oh * ow), dim3(256), 0, (hipStream_t)stream, src, Mi, Mj, D, wy, wx,
                     oh, ow, dst);
  return pi3_check_launch("resample_grid");
}

// Fill token rows [f*T + t0, f*T + t0 + nt) of an fp32 [F*T][D] buffer with vals[nt][D] (+ optional add[D]):
// cls token + pos_embed[0] and the 4 encoder register tokens (vision_transformer.py:221-232).
__global__ __launch_bounds__(256) void fill_tokens_kernel(float* x, int F, int T, int D, int t0, int nt,
                                                          const float* __restrict__ vals) {
  const long total = (long)F * nt * D;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int d = (int)(i % D);
    const long ft = i / D;
    const int t = (int)(ft % nt);
    const long f = ft / nt;
    x[(f * T + t0 + t) * D + d] = vals[(long)t * D + d];
  }
}

extern "C" int pi3_fill_tokens(float* x, int F, int T, int D, int t0, int nt, const float* vals, void* stream) {
  if (!x || !vals || F <= 0 || T <= 0 || D <= 0 || t0 < 0 || nt <= 0 || t0 + nt > T) {
    pi3_set_error("pi3_fill_tokens: bad arguments");
    return PI3_ERR_ARG;
  }
  long blocks = ((long)F * nt * D + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_tokens_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, F, T, D, t0,
                     nt, vals);
  return pi3_check_launch("fill_tokens");
}
