// debug: intermediates of the keypoint gather (post.hip: bil_setup / bil_sample) for one element
#include "../../pi3_slam_amd/csrc/post.hip"
#include <cstdio>
__global__ void dbg(const float* v4, float kx, float ky, int H, int W, float* out) {
  Bil b = bil_setup(kx, ky, H, W);
  out[0] = b.nw; out[1] = b.ne; out[2] = b.sw; out[3] = b.se; out[4] = (float)b.x0; out[5] = (float)b.y0;
  // a 2x2 image laid out so that base + y0*sy + x0*sx hits v4[0]
  const float r = bil_sample(v4 - (b.y0 * 2 + b.x0), 1, 2, b, H, W);
  out[6] = r;
  out[7] = __half2float(__float2half_rn(r));
  out[8] = __half2float(__float2half_rn(6.923828125f));
  const float gx = (kx / (float)(W - 1)) * 2.0f - 1.0f;
  out[9] = gx;
  out[10] = kx / (float)(W - 1);
}
int main() {
  float hv[4] = {6.9417434f, 6.922761f, 6.944261f, 6.925186f};
  float *dv, *dout;
  hipMalloc(&dv, 16 + 4096 * 4); hipMalloc(&dout, 64);
  // place the 2x2 block far enough into the buffer that the negative offset stays inside it
  hipMemcpy(dv + 1024, hv, 16, hipMemcpyHostToDevice);
  // x0 = 36, y0 = 147 -> offset 147*2+36 = 330 < 1024
  dbg<<<1, 1>>>(dv + 1024, 37.400001525878906f, 147.39999389648438f, 308, 406, dout);
  float o[16];
  hipMemcpy(o, dout, 44, hipMemcpyDeviceToHost);
  printf("nw %.9g ne %.9g sw %.9g se %.9g x0 %g y0 %g\nresult %.9g half %.9g tie-half %.9g gx %.9g div %.9g\n", o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7], o[8], o[9], o[10]);
  return 0;
}
