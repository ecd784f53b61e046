// Undistortion of the input frames (pi3/utils/undistortion.py:95-138 map builder, :157-177 cv2.remap), the ingest of
// BASELINE config 4 (EuRoC + euroc_cam0_calib.json).  SURVEY.md §8f rank 4.
//
// Map builder: the reference walks every target pixel (c, r) in python, lifts it through the UNDISTORTED camera
// (ImageToCameraCoordinates, distortion zeroed, aspect ratio 1) and projects the ray through the DISTORTED camera
// (CameraToImageCoordinates) with pytheia's camera models.  pytheia (TheiaSfM camera models) is not vendored: the four
// models are restated from TheiaSfM's published headers (pinhole_camera_model.h, pinhole_radial_tangential_camera_model.h,
// fisheye_camera_model.h, division_undistortion_camera_model.h) - parity unpinned.  One thread per pixel, fp64 like the
// library, maps stored as float32 like the reference (np.float32 arrays).
// Remap: cv2.remap(img, map_x, map_y, INTER_LINEAR) on 8-bit images, BORDER_CONSTANT 0, restated from OpenCV's published
// imgwarp.cpp (cv2 is not installed here: parity unpinned): coordinates rounded to 1/32 pixel (cvRound = round half to
// even), bilinear weights (32-fx)(32-fy)*32 ... in 15-bit fixed point (exact, they sum to 32768), result
// (sum + 2^14) >> 15, taps outside the image read 0.  Fused with ToTensor (uint8 HWC -> float32 CHW / 255).
#include "common.h"

enum { UD_PINHOLE = 0, UD_PINHOLE_RADIAL_TANGENTIAL = 1, UD_FISHEYE = 2, UD_DIVISION_UNDISTORTION = 3 };

struct UndistortParams {
  // undistorted camera (rays): focal, aspect ratio, principal point, skew; distorted camera: the same + distortion
  double fu, aru, cxu, cyu, sku;
  double fd, ard, cxd, cyd, skd;
  double k[4];
  double t[2];
  int model, H, W;
};

__global__ __launch_bounds__(256) void undistort_map_kernel(UndistortParams p, float* __restrict__ map_x,
                                                            float* __restrict__ map_y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.H * p.W) return;
  const double c = (double)(i % p.W), r = (double)(i / p.W);
  // ImageToCameraCoordinates of the undistorted camera (zero distortion: the undistortion step is the identity)
  double y = (r - p.cyu) / (p.fu * p.aru);
  double x = (p.model == UD_DIVISION_UNDISTORTION) ? (c - p.cxu) / p.fu : (c - p.cxu - p.sku * y) / p.fu;
  // CameraToImageCoordinates of the distorted camera for the ray (x, y, 1)
  double u, v;
  if (p.model == UD_DIVISION_UNDISTORTION) {
    double ux = p.fd * x, uy = p.fd * p.ard * y;            // undistorted pixel relative to the principal point
    const double rr = ux * ux + uy * uy, kk = p.k[0];
    if (fabs(kk) >= 1e-15 && rr >= 1e-15) {
      const double inner = 1.0 - 4.0 * kk * rr;
      if (inner >= 0.0) {
        const double sc = (1.0 - sqrt(inner)) / (2.0 * kk * rr);
        ux *= sc; uy *= sc;
      }
    }
    u = ux + p.cxd; v = uy + p.cyd;
  } else {
    double dx, dy;
    if (p.model == UD_FISHEYE) {
      const double r_sq = x * x + y * y;
      if (r_sq < 1e-8) {
        dx = x; dy = y;
      } else {
        const double rad = sqrt(r_sq);
        const double th = atan2(rad, 1.0), th2 = th * th;
        const double thd = th * (1.0 + th2 * (p.k[0] + th2 * (p.k[1] + th2 * (p.k[2] + th2 * p.k[3]))));
        dx = thd * x / rad; dy = thd * y / rad;
      }
    } else {
      const double r_sq = x * x + y * y;
      const double k3 = (p.model == UD_PINHOLE_RADIAL_TANGENTIAL) ? p.k[2] : 0.0;
      const double radial = 1.0 + r_sq * (p.k[0] + r_sq * (p.k[1] + r_sq * k3));
      dx = x * radial; dy = y * radial;
      if (p.model == UD_PINHOLE_RADIAL_TANGENTIAL) {
        const double xy = x * y;
        dx += 2.0 * p.t[0] * xy + p.t[1] * (r_sq + 2.0 * x * x);
        dy += p.t[0] * (r_sq + 2.0 * y * y) + 2.0 * p.t[1] * xy;
      }
    }
    u = p.fd * dx + p.skd * dy + p.cxd;
    v = p.fd * p.ard * dy + p.cyd;
  }
  map_x[i] = (float)u;
  map_y[i] = (float)v;
}

// params (HOST pointer, 16 doubles): [0..4] undistorted camera f, aspect, cx, cy, skew; [5..9] distorted camera f, aspect,
// cx, cy, skew; [10..13] radial k1..k4 (division model: k in [10]); [14..15] tangential t1, t2
extern "C" int pi3_undistort_maps(const double* params, int model, int H, int W, float* map_x, float* map_y,
                                  void* stream) {
  if (!params || !map_x || !map_y || H <= 0 || W <= 0 || model < 0 || model > 3) {
    pi3_set_error("pi3_undistort_maps: bad arguments model=%d H=%d W=%d", model, H, W);
    return PI3_ERR_ARG;
  }
  UndistortParams p;
  p.fu = params[0]; p.aru = params[1]; p.cxu = params[2]; p.cyu = params[3]; p.sku = params[4];
  p.fd = params[5]; p.ard = params[6]; p.cxd = params[7]; p.cyd = params[8]; p.skd = params[9];
  for (int i = 0; i < 4; ++i) p.k[i] = params[10 + i];
  p.t[0] = params[14]; p.t[1] = params[15];
  p.model = model; p.H = H; p.W = W;
  hipLaunchKernelGGL(undistort_map_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, map_x, map_y);
  return pi3_check_launch("undistort_maps");
}

__global__ __launch_bounds__(256) void remap_bilinear_kernel(const uint8_t* __restrict__ src, int N, int H0, int W0,
                                                             const float* __restrict__ map_x,
                                                             const float* __restrict__ map_y, int H, int W,
                                                             float* __restrict__ dst) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // (frame, y, x)
  const long plane = (long)H * W;
  if (i >= (long)N * plane) return;
  const long f = i / plane;
  const int pix = (int)(i - f * plane);
  // cvRound(map * 32): round half to even, as rint in the default rounding mode; saturate like cv::saturate_cast<int>
  const float mx = map_x[pix] * 32.0f, my = map_y[pix] * 32.0f;
  const int sxq = (mx != mx) ? 0 : (int)fminf(fmaxf(rintf(mx), -2147483648.0f), 2147483520.0f);
  const int syq = (my != my) ? 0 : (int)fminf(fmaxf(rintf(my), -2147483648.0f), 2147483520.0f);
  // integer part goes through saturate_cast<short>, the fraction stays 5 bits
  const int sx = min(max(sxq >> 5, -32768), 32767), sy = min(max(syq >> 5, -32768), 32767);
  const int fx = sxq & 31, fy = syq & 31;
  const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
  const uint8_t* img = src + f * (long)H0 * W0 * 3;
  const bool x0 = (unsigned)sx < (unsigned)W0, x1 = (unsigned)(sx + 1) < (unsigned)W0;
  const bool y0 = (unsigned)sy < (unsigned)H0, y1 = (unsigned)(sy + 1) < (unsigned)H0;
  int acc[3] = {1 << 14, 1 << 14, 1 << 14};
  if (y0 && x0) { const uint8_t* q = img + ((long)sy * W0 + sx) * 3; acc[0] += q[0] * w00; acc[1] += q[1] * w00; acc[2] += q[2] * w00; }
  if (y0 && x1) { const uint8_t* q = img + ((long)sy * W0 + sx + 1) * 3; acc[0] += q[0] * w01; acc[1] += q[1] * w01; acc[2] += q[2] * w01; }
  if (y1 && x0) { const uint8_t* q = img + ((long)(sy + 1) * W0 + sx) * 3; acc[0] += q[0] * w10; acc[1] += q[1] * w10; acc[2] += q[2] * w10; }
  if (y1 && x1) { const uint8_t* q = img + ((long)(sy + 1) * W0 + sx + 1) * 3; acc[0] += q[0] * w11; acc[1] += q[1] * w11; acc[2] += q[2] * w11; }
  float* d = dst + f * 3 * plane + pix;
  d[0] = (float)min(acc[0] >> 15, 255) / 255.0f;
  d[plane] = (float)min(acc[1] >> 15, 255) / 255.0f;
  d[2 * plane] = (float)min(acc[2] >> 15, 255) / 255.0f;
}

extern "C" int pi3_remap_bilinear_u8(const unsigned char* src, int N, int H0, int W0, const float* map_x,
                                     const float* map_y, int H, int W, float* dst, void* stream) {
  if (!src || !map_x || !map_y || !dst || N <= 0 || H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) {
    pi3_set_error("pi3_remap_bilinear_u8: bad arguments");
    return PI3_ERR_ARG;
  }
  const long n = (long)N * H * W;
  hipLaunchKernelGGL(remap_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     N, H0, W0, map_x, map_y, H, W, dst);
  return pi3_check_launch("remap_bilinear_u8");
}
