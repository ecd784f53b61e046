// HBM write / read / copy bandwidth of the part as a function of the number of workgroups (CUs) taking part and of the
// bytes each lane keeps in flight: what a GEMM epilogue that only stores (bf16 output) or reads + stores (fp32
// residual) can reach, and whether fewer CUs storing at a time go faster per CU (they would if the chip-wide write rate
// were the limit and not a per-CU one).  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/wr_bw.hip -o tools/micro/wr_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: store only; 1: load only (sum kept); 2: copy; 3: read-modify-write in place (the residual epilogue)
template <int MODE, int UNROLL>
__global__ __launch_bounds__(512) void bw_kernel(f32x4* __restrict__ dst, const f32x4* __restrict__ src, long n, float* sink) {
  const long stride = (long)gridDim.x * 512;
  long i = (long)blockIdx.x * 512 + threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE == 0) v[u] = (f32x4){1.f, 2.f, 3.f, (float)u};
      else if (MODE == 3) v[u] = dst[i + u * stride];
      else v[u] = src[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (MODE == 1) acc += v[u];
      else if (MODE == 3) dst[i + u * stride] = v[u] + (f32x4){1.f, 1.f, 1.f, 1.f};
      else dst[i + u * stride] = v[u];
    }
  }
  if (MODE == 1 && acc[0] == 123.456f) *sink = acc[1];
}

template <int MODE, int UNROLL>
static void run(const char* name, f32x4* a, f32x4* b, long n, float* sink, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((bw_kernel<MODE, UNROLL>), dim3(wgs), dim3(512), 0, 0, a, b, n, sink);
  hipEventRecord(e0);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((bw_kernel<MODE, UNROLL>), dim3(wgs), dim3(512), 0, 0, a, b, n, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double bytes = (double)n * 16 * (MODE == 2 || MODE == 3 ? 2 : 1);
  printf("%-6s unroll %d wgs %5d: %8.3f ms  %6.2f TB/s  (%.1f GB/s per workgroup)\n", name, UNROLL, wgs, ms, bytes / ms / 1e9,
         bytes / ms / 1e6 / wgs);
}

int main() {
  const long n = (long)512 * 1024 * 1024 / 16;       // 512 MB per array (beyond the 256 MB infinity cache)
  f32x4 *a, *b; float* sink;
  hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&sink, 4);
  hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16);
  for (int wgs : {64, 128, 256, 512, 1024, 4096}) {
    run<0, 1>("store", a, b, n, sink, wgs);
    run<0, 4>("store", a, b, n, sink, wgs);
    run<0, 8>("store", a, b, n, sink, wgs);
    run<1, 4>("load", a, b, n, sink, wgs);
    run<2, 4>("copy", a, b, n, sink, wgs);
    run<3, 4>("rmw", a, b, n, sink, wgs);
  }
  return 0;
}
