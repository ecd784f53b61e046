// Deterministic "recipe" weights: a counter-based integer generator that numpy reproduces bit for bit
// (pi3_slam_amd/weights.py: recipe_tensor).  There are no pretrained checkpoints offline, so parity
// fixtures, tests and the bench all use weights regenerated from (parameter name -> seed, offset, scale).
//   z = seed + (i + 1) * 0x9E3779B97F4A7C15;  splitmix64 finaliser;  u = z >> 40 (24 bits)
//   value = (float)((double)offset + (double)scale * (u * 2^-23 - 1))
// The product of two fp32 values is exact in fp64, so the result does not depend on whether the compiler contracts
// the multiply-add: one fp64 rounding of the sum, one fp64 -> fp32 rounding, identical in numpy and on the device.
#include "common.h"

__device__ __forceinline__ float recipe_value(uint64_t seed, uint64_t i, float offset, float scale) {
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float f = (float)(uint32_t)(z >> 40) * 1.1920928955078125e-07f - 1.0f;  // exact
  return (float)((double)offset + (double)scale * (double)f);
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256) void recipe_fill_kernel(void* out, long n, uint64_t seed, float offset,
                                                          float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = recipe_value(seed, (uint64_t)i, offset, scale);
    if constexpr (OUT_BF16)
      ((bf16_t*)out)[i] = (bf16_t)v;
    else
      ((float*)out)[i] = v;
  }
}

extern "C" int pi3_recipe_fill(void* out, long n, unsigned long long seed, float offset, float scale, int out_dtype,
                               void* stream) {
  if (!out || n <= 0) {
    pi3_set_error("pi3_recipe_fill: bad arguments");
    return PI3_ERR_ARG;
  }
  long blocks = (n + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (out_dtype == 0)
    hipLaunchKernelGGL(recipe_fill_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n,
                       (uint64_t)seed, offset, scale);
  else
    hipLaunchKernelGGL(recipe_fill_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, n,
                       (uint64_t)seed, offset, scale);
  return pi3_check_launch("recipe_fill");
}
