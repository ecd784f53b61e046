// Which element of D = A * B each (lane, register) of v_mfma_f64_16x16x4_f64 holds, and which (row, k) / (k, col) the
// A / B operand lanes supply: A[i][k] = 1000 (i + 1) + k, B[k][j] = (j + 1) for k == kprobe else 0, so
// D[i][j] = (1000 (i + 1) + kprobe) (j + 1) identifies (i, j) and confirms the k of a lane group.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip && ./mfma_f64_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int amap, int kprobe) {
  const int l = threadIdx.x;
  // hypothesis: A lane l -> (i = l % 16, k = l / 16); B lane l -> (k = l / 16, j = l % 16)
  const int i = l % 16, k = l / 16;
  const double a = 1000.0 * (i + 1) + k;
  const double b = (k == kprobe) ? (double)(i + 1) : 0.0;
  v4 d = {0, 0, 0, 0};
  d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d, 0, 0, 0);
  for (int v = 0; v < 4; ++v) out[l * 4 + v] = d[v];
}
int main() {
  double* dev;
  hipMalloc(&dev, 256 * sizeof(double));
  double h[256];
  for (int kp = 0; kp < 4; ++kp) {
    probe<<<1, 64>>>(dev, 0, kp);
    hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    int bad_k = 0, hyp1 = 0, hyp2 = 0;
    for (int l = 0; l < 64; ++l)
      for (int v = 0; v < 4; ++v) {
        const double x = h[l * 4 + v];
        // decode: x = (1000 (i + 1) + kp) (j + 1)
        int fi = -1, fj = -1;
        for (int i = 0; i < 16; ++i)
          for (int j = 0; j < 16; ++j)
            if (x == (1000.0 * (i + 1) + kp) * (j + 1)) { fi = i; fj = j; }
        if (fi < 0) ++bad_k;
        if (fi == 4 * (l / 16) + v && fj == l % 16) ++hyp1;
        if (fi == (l / 16) + 4 * v && fj == l % 16) ++hyp2;
        if (kp == 0 && (l == 0 || l == 1 || l == 16 || l == 17)) printf("lane %2d reg %d -> D[%d][%d]\n", l, v, fi, fj);
      }
    printf("kprobe %d: undecodable %d, i = 4 (l / 16) + v: %d / 256, i = l / 16 + 4 v: %d / 256\n", kp, bad_k, hyp1, hyp2);
  }
  return 0;
}
