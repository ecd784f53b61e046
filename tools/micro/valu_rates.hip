// Issue-rate microbenchmark for the attention inner loop (gfx950): cycles per instruction of v_exp_f32, v_cvt_pk_bf16_f32,
// v_add_f32 and their mixes, at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {   // 16 independent exps
      REP16(asm volatile("v_exp_f32 %0, %0\n" : "+v"(a0));) // dependent chain on a0 (latency)
    } else if (MODE == 1) {   // 16 exps over 8 registers (independent pairs)
      asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n"
                   "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 2) {   // 16 adds over 8 registers
      asm volatile("v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1\nv_add_f32 %2, %2, %2\nv_add_f32 %3, %3, %3\nv_add_f32 %4, %4, %4\nv_add_f32 %5, %5, %5\nv_add_f32 %6, %6, %6\nv_add_f32 %7, %7, %7\n"
                   "v_add_f32 %0, %0, %0\nv_add_f32 %1, %1, %1\nv_add_f32 %2, %2, %2\nv_add_f32 %3, %3, %3\nv_add_f32 %4, %4, %4\nv_add_f32 %5, %5, %5\nv_add_f32 %6, %6, %6\nv_add_f32 %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 3) {   // 8 exps interleaved with 8 adds
      asm volatile("v_exp_f32 %0, %0\nv_add_f32 %4, %4, %4\nv_exp_f32 %1, %1\nv_add_f32 %5, %5, %5\nv_exp_f32 %2, %2\nv_add_f32 %6, %6, %6\nv_exp_f32 %3, %3\nv_add_f32 %7, %7, %7\n"
                   "v_exp_f32 %0, %0\nv_add_f32 %4, %4, %4\nv_exp_f32 %1, %1\nv_add_f32 %5, %5, %5\nv_exp_f32 %2, %2\nv_add_f32 %6, %6, %6\nv_exp_f32 %3, %3\nv_add_f32 %7, %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 4) {   // 16 cvt_pk
      asm volatile("v_cvt_pk_bf16_f32 %8, %0, %1\nv_cvt_pk_bf16_f32 %9, %2, %3\nv_cvt_pk_bf16_f32 %10, %4, %5\nv_cvt_pk_bf16_f32 %11, %6, %7\n"
                   "v_cvt_pk_bf16_f32 %8, %1, %0\nv_cvt_pk_bf16_f32 %9, %3, %2\nv_cvt_pk_bf16_f32 %10, %5, %4\nv_cvt_pk_bf16_f32 %11, %7, %6\n"
                   "v_cvt_pk_bf16_f32 %8, %0, %2\nv_cvt_pk_bf16_f32 %9, %1, %3\nv_cvt_pk_bf16_f32 %10, %4, %6\nv_cvt_pk_bf16_f32 %11, %5, %7\n"
                   "v_cvt_pk_bf16_f32 %8, %2, %0\nv_cvt_pk_bf16_f32 %9, %3, %1\nv_cvt_pk_bf16_f32 %10, %6, %4\nv_cvt_pk_bf16_f32 %11, %7, %5\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    } else if (MODE == 5) {   // 8 x (exp, exp, cvt) = the softmax mix without MFMA
      asm volatile("v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_cvt_pk_bf16_f32 %8, %4, %5\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_cvt_pk_bf16_f32 %9, %6, %7\n"
                   "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_cvt_pk_bf16_f32 %10, %0, %1\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\nv_cvt_pk_bf16_f32 %11, %2, %3\n"
                   "v_exp_f32 %0, %0\nv_exp_f32 %1, %1\nv_cvt_pk_bf16_f32 %8, %4, %5\nv_exp_f32 %2, %2\nv_exp_f32 %3, %3\nv_cvt_pk_bf16_f32 %9, %6, %7\n"
                   "v_exp_f32 %4, %4\nv_exp_f32 %5, %5\nv_cvt_pk_bf16_f32 %10, %0, %1\nv_exp_f32 %6, %6\nv_exp_f32 %7, %7\nv_cvt_pk_bf16_f32 %11, %2, %3\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(b0 + b1 + b2 + b3);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int n_instr, int threads) {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int rep = 0; rep < 30; ++rep)   // the first launches run while the clocks ramp: keep the last one
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s waves/SIMD=%d  %.2f cycles per instruction per wave (%.2f per SIMD-instruction)\n", name, threads / 256,
         (double)h / iters / n_instr, (double)h / iters / n_instr / (threads / 256));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int th : {256, 512, 256, 512}) {
    run<0>("v_exp dependent chain", 16, th);
    run<1>("v_exp independent", 16, th);
    run<2>("v_add independent", 16, th);
    run<3>("v_exp + v_add alternating", 16, th);
    run<4>("v_cvt_pk_bf16_f32", 16, th);
    run<5>("exp exp cvt mix", 24, th);
  }
  return 0;
}
