// Frame ingest on the device: the Resize + ToTensor of ChunkImageDataset._load_image_chunk
// (datasets/image_datasets.py:186-208).  torchvision's Resize on a PIL image is Pillow's ImagingResample with the
// bilinear (triangle) filter: antialiased taps, 22-bit fixed-point weights, a horizontal pass into a uint8 image and a
// vertical pass over it, each  clip8((2^21 + sum(pixel * k)) >> 22).  The weight tables come from the host
// (pi3_slam_amd/image_io.py: resample_coeffs, double arithmetic in Pillow's operation order); these kernels do the
// integer passes bit-exactly and fuse ToTensor (uint8 HWC -> float32 CHW / 255) into the second one.  Frames cross PCIe
// as uint8 (3 B/pixel at the source size) instead of float32 at the target size.
#include "common.h"

#define INGEST_PREC 22

__global__ __launch_bounds__(256) void ingest_horizontal_kernel(const uint8_t* __restrict__ src, int rows, int W0,
                                                                int W1, const int* __restrict__ bounds,
                                                                const int* __restrict__ coefs, int ksize,
                                                                uint8_t* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // (row, xo)
  if (i >= (long)rows * W1) return;
  const int xo = (int)(i % W1);
  const long row = i / W1;
  const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int* k = coefs + (long)xo * ksize;
  const uint8_t* s = src + (row * W0 + x0) * 3;
  int a0 = 1 << (INGEST_PREC - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    a0 += (int)s[3 * t] * w;
    a1 += (int)s[3 * t + 1] * w;
    a2 += (int)s[3 * t + 2] * w;
  }
  uint8_t* d = dst + i * 3;
  d[0] = (uint8_t)min(max(a0 >> INGEST_PREC, 0), 255);
  d[1] = (uint8_t)min(max(a1 >> INGEST_PREC, 0), 255);
  d[2] = (uint8_t)min(max(a2 >> INGEST_PREC, 0), 255);
}

__global__ __launch_bounds__(256) void ingest_vertical_kernel(const uint8_t* __restrict__ src, int N, int H0, int W1,
                                                              int H1, const int* __restrict__ bounds,
                                                              const int* __restrict__ coefs, int ksize,
                                                              float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // (frame, yo, x)
  if (i >= (long)N * H1 * W1) return;
  const int x = (int)(i % W1);
  const int yo = (int)((i / W1) % H1);
  const long f = i / ((long)W1 * H1);
  const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int* k = coefs + (long)yo * ksize;
  const uint8_t* s = src + ((f * H0 + y0) * W1 + x) * 3;
  int a0 = 1 << (INGEST_PREC - 1), a1 = a0, a2 = a0;
  for (int t = 0; t < n; ++t) {
    const int w = k[t];
    const uint8_t* q = s + (long)t * W1 * 3;
    a0 += (int)q[0] * w;
    a1 += (int)q[1] * w;
    a2 += (int)q[2] * w;
  }
  const long plane = (long)H1 * W1;
  float* d = dst + f * 3 * plane + (long)yo * W1 + x;
  d[0] = (float)min(max(a0 >> INGEST_PREC, 0), 255) / 255.0f;           // ToTensor: correctly rounded fp32 division
  d[plane] = (float)min(max(a1 >> INGEST_PREC, 0), 255) / 255.0f;
  d[2 * plane] = (float)min(max(a2 >> INGEST_PREC, 0), 255) / 255.0f;
}

extern "C" int pi3_ingest_frames(const unsigned char* src, int N, int H0, int W0, int H1, int W1, const int* xbounds,
                                 const int* xcoefs, int xksize, const int* ybounds, const int* ycoefs, int yksize,
                                 unsigned char* tmp, float* dst, void* stream) {
  if (!src || !xbounds || !xcoefs || !ybounds || !ycoefs || !tmp || !dst || N <= 0 || H0 <= 0 || W0 <= 0 || H1 <= 0 ||
      W1 <= 0 || xksize <= 0 || yksize <= 0) {
    pi3_set_error("pi3_ingest_frames: bad arguments");
    return PI3_ERR_ARG;
  }
  const long n1 = (long)N * H0 * W1, n2 = (long)N * H1 * W1;
  hipLaunchKernelGGL(ingest_horizontal_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, N * H0, W0, W1, xbounds, xcoefs, xksize, tmp);
  hipLaunchKernelGGL(ingest_vertical_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     tmp, N, H0, W1, H1, ybounds, ycoefs, yksize, dst);
  return pi3_check_launch("ingest_frames");
}
