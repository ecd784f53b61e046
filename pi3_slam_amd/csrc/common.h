// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the Pi3-SLAM hot path.
// Wave width is 64 everywhere; nothing in here is portable to 32-wide hardware and nothing tries to be.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// error codes returned through the C-ABI (include/pi3slam_hip.h)
enum {
  PI3_OK = 0,
  PI3_ERR_ARG = -1,      // bad shape / null pointer / unsupported size
  PI3_ERR_LAUNCH = -2,   // hipLaunch / runtime error (see pi3_last_error)
  PI3_ERR_WORKSPACE = -3 // caller workspace too small
};

void pi3_set_error(const char* fmt, ...);
int pi3_check_launch(const char* what);
// Opt `kern` into `bytes` of dynamic LDS (above the 64 KB default) on the CURRENT device, once per device: the attribute
// is per device, so `done_mask` (one function-local static per kernel instance) carries one bit per device ordinal.
// Returns PI3_OK or PI3_ERR_LAUNCH with the runtime's message in pi3_last_error().
int pi3_lds_optin(const void* kern, int bytes, unsigned long long* done_mask, const char* what);

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}

// fp32 -> bf16 round-to-nearest-even through the compiler's cast (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN).
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v;
  v[0] = (bf16_t)lo;
  v[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bijective XCD-aware remap (8 XCDs, blocks dealt round-robin): consecutive logical ids land on one XCD so
// neighbouring tiles share that XCD's L2. Speed only; correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// GELU(erf) = 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU default, pi3/models/dinov2/layers/mlp.py:36).  erf by Abramowitz-Stegun
// 7.1.26 (|abs error| <= 1.5e-7, far below the bf16 / fp32-accumulate noise of the GEMM feeding it): one v_rcp, one
// v_exp and five FMAs instead of the ~40-instruction branchy libm erff, which made the fc1 epilogue cost 70 % of its GEMM.
__device__ __forceinline__ float gelu_erf(float x) {
#ifdef PI3_GELU_OLD_FORM
  const float ax = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
  const float erf_abs = 1.0f - poly * e;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
#else
  // the same formula with the constants folded and the sign handled by x * erf(x / sqrt 2) = |x| erf(|x| / sqrt 2):
  // 0.5 x (1 + erf) = hx + |hx| erf_abs with hx = x / 2 - no copysign, 11 regular vector operations instead of 14
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.3275911f * 0.70710678118654752440f, 1.0f));
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float z = ax * 0.84932180028801904272f;          // sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-z^2)
  const float e = __builtin_amdgcn_exp2f(-z * z);
  const float hx = 0.5f * x;
  return fmaf(fabsf(hx), 1.0f - poly * e, hx);
#endif
}
