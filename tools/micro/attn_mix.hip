// Can two waves per SIMD overlap the attention tile's instruction mix (d = 64: 32 MFMA 32x32x16 + 16 MFMA 4x4x4 + 64 v_exp_f32
// + 32 v_cvt_pk_bf16_f32 per 64x64 tile and wave) when the operands are already in registers?  Streams are inline asm,
// one statement per instruction, so the order below is the order issued.
//   MODE 0  phased, as attn64.hip emits it: 16 MFMA | 16 x (4 exp, 2 cvt, 1 MFMA) | 16 row-sum MFMA
//   MODE 1  interleaved: 32 x (MFMA, exp, exp, cvt) with a row-sum MFMA in every other group
//   MODE 2  MFMAs only (32 + 16)        MODE 3  vector work only (64 exp + 32 cvt)
//   MODE 4  MODE 0 + the LDS fragment reads where hipcc puts them: 2 ds_read_b128 in front of every 4 QK^T MFMAs,
//           2 ds_read_b64_tr_b16 in front of every 2 P.V MFMAs, counted lgkmcnt waits
//   MODE 5  MODE 4 with the reads hoisted: the 8 K fragments of the tile before its first MFMA, the 16 V reads in front of
//           the exps (registers: +32 and +32)
//   MODE 6  MODE 5 + one barrier per tile
// hipcc --offload-arch=gfx950 -O3 -o attn_mix attn_mix.hip ; ./attn_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define BIG(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(fa), "v"(fb))
#define SMALL(ACC) asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(sa), "v"(sb))
#define EXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define CVT(D, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(Y))
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#define BIGF(ACC, FA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(FA), "v"(fb))
#define RD128(D, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(D) : "v"(la128))
#define RDTR(D, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=v"(D) : "v"(la64))
#define LGKM(N) asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory")
__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x16)(0.f);
  f32x4 l0 = (f32x4)(0.f), l1 = (f32x4)(0.f);
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (threadIdx.x & 7)); fb[i] = (__bf16)(0.02f * (threadIdx.x & 3)); }
  s16x4 sa = {0x3f80, 0x3f80, 0x3f80, 0x3f80}, sb = {0x3c00, 0x3c00, 0x3c00, 0x3c00};
  float e[16];
  for (int i = 0; i < 16; ++i) e[i] = 1e-3f * (threadIdx.x + i);
  unsigned c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  __shared__ __attribute__((aligned(16))) char tile[32768];
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((unsigned*)tile)[i] = 0x3c003c00u;
  __syncthreads();
  const unsigned la128 = (unsigned)(size_t)tile + (threadIdx.x & 63) * 16, la64 = (unsigned)(size_t)tile + 16384 + (threadIdx.x & 63) * 8;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) BIG(acc[i & 3]);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        EXP(e[(4 * g) & 15]); EXP(e[(4 * g + 1) & 15]); EXP(e[(4 * g + 2) & 15]); EXP(e[(4 * g + 3) & 15]);
        CVT(c[(2 * g) & 7], e[(4 * g) & 15], e[(4 * g + 1) & 15]);
        CVT(c[(2 * g + 1) & 7], e[(4 * g + 2) & 15], e[(4 * g + 3) & 15]);
        BIG(acc[4 + (g & 3)]);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { SMALL(l0); SMALL(l1); }
    } else if (MODE == 1) {
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        BIG(acc[g & 7]);
        EXP(e[(2 * g) & 15]); EXP(e[(2 * g + 1) & 15]);
        CVT(c[g & 7], e[(2 * g + 8) & 15], e[(2 * g + 9) & 15]);
        if (g & 1) { SMALL(l0); } else { SMALL(l1); }
      }
    } else if (MODE == 4) {
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        bf16x8 a0, a1;
        RD128(a0, 0); RD128(a1, 4096);
        LGKM(1); BIGF(acc[0], a0); BIGF(acc[1], a0);
        LGKM(0); BIGF(acc[2], a1); BIGF(acc[3], a1);
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        EXP(e[(4 * g) & 15]); EXP(e[(4 * g + 1) & 15]); EXP(e[(4 * g + 2) & 15]); EXP(e[(4 * g + 3) & 15]);
        CVT(c[(2 * g) & 7], e[(4 * g) & 15], e[(4 * g + 1) & 15]);
        CVT(c[(2 * g + 1) & 7], e[(4 * g + 2) & 15], e[(4 * g + 3) & 15]);
        if ((g & 1) == 0) {
          bf16x4 lo, hi;
          RDTR(lo, 0); RDTR(hi, 1024);
          LGKM(0);
          bf16x8 vf = cat4(lo, hi);
          BIGF(acc[4 + ((g >> 1) & 1) * 2], vf); BIGF(acc[5 + ((g >> 1) & 1) * 2], vf);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { SMALL(l0); SMALL(l1); }
    } else if (MODE == 5 || MODE == 6) {
      bf16x8 ka[8];
      bf16x4 vlo[8], vhi[8];
      RD128(ka[0], 0); RD128(ka[1], 4096); RD128(ka[2], 1024); RD128(ka[3], 5120);
      RD128(ka[4], 2048); RD128(ka[5], 6144); RD128(ka[6], 3072); RD128(ka[7], 7168);
      LGKM(6); BIGF(acc[0], ka[0]); BIGF(acc[1], ka[0]); BIGF(acc[2], ka[1]); BIGF(acc[3], ka[1]);
      LGKM(4); BIGF(acc[0], ka[2]); BIGF(acc[1], ka[2]); BIGF(acc[2], ka[3]); BIGF(acc[3], ka[3]);
      RDTR(vlo[0], 0); RDTR(vhi[0], 1024); RDTR(vlo[1], 512); RDTR(vhi[1], 1536);
      RDTR(vlo[2], 2048); RDTR(vhi[2], 3072); RDTR(vlo[3], 2560); RDTR(vhi[3], 3584);
      LGKM(10); BIGF(acc[0], ka[4]); BIGF(acc[1], ka[4]); BIGF(acc[2], ka[5]); BIGF(acc[3], ka[5]);
      LGKM(8); BIGF(acc[0], ka[6]); BIGF(acc[1], ka[6]); BIGF(acc[2], ka[7]); BIGF(acc[3], ka[7]);
      RDTR(vlo[4], 4096); RDTR(vhi[4], 5120); RDTR(vlo[5], 4608); RDTR(vhi[5], 5632);
      RDTR(vlo[6], 6144); RDTR(vhi[6], 7168); RDTR(vlo[7], 6656); RDTR(vhi[7], 7680);
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        EXP(e[(4 * g) & 15]); EXP(e[(4 * g + 1) & 15]); EXP(e[(4 * g + 2) & 15]); EXP(e[(4 * g + 3) & 15]);
        CVT(c[(2 * g) & 7], e[(4 * g) & 15], e[(4 * g + 1) & 15]);
        CVT(c[(2 * g + 1) & 7], e[(4 * g + 2) & 15], e[(4 * g + 3) & 15]);
        if (g == 0) LGKM(0);
        if ((g & 1) == 0) {
          bf16x8 vf = cat4(vlo[g >> 1], vhi[g >> 1]);
          BIGF(acc[4 + ((g >> 1) & 1) * 2], vf); BIGF(acc[5 + ((g >> 1) & 1) * 2], vf);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) { SMALL(l0); SMALL(l1); }
      if (MODE == 6) __builtin_amdgcn_s_barrier();
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 32; ++i) BIG(acc[i & 7]);
#pragma unroll
      for (int i = 0; i < 8; ++i) { SMALL(l0); SMALL(l1); }
    } else {
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        EXP(e[(2 * g) & 15]); EXP(e[(2 * g + 1) & 15]);
        CVT(c[g & 7], e[(2 * g + 8) & 15], e[(2 * g + 9) & 15]);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = l0[0] + l1[0];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + (float)c[i];
  for (int i = 0; i < 16; ++i) s += e[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 20; ++rep) {   // the first launches run while the clocks settle: keep the last one
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
  }
  hipDeviceSynchronize();
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  printf("%-44s waves/SIMD=%d  %7.0f counter ticks per tile and wave, %7.3f us per tile-round (%d waves), %.0f TF/s-equivalent\n", name,
         wps, (double)h / iters, ms * 1e3 / iters, wps,
         256.0 * 4 * wps * iters * (32.0 * 32 * 32 * 16 * 2) / (ms * 1e-3) / 1e12);
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int th : {256, 512}) {
    run<2>("MFMAs only (32 big + 16 row-sum)", th);
    run<3>("vector only (64 exp + 32 cvt)", th);
    run<0>("phased (as emitted today)", th);
    run<1>("interleaved (MFMA, exp, exp, cvt [, row-sum])", th);
    run<4>("phased + LDS reads in front of their MFMAs", th);
    run<5>("phased + LDS reads hoisted", th);
    run<6>("phased + LDS reads hoisted + barrier per tile", th);
  }
  return 0;
}
