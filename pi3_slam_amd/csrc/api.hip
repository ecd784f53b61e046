// C-ABI plumbing shared by every entry point: error string, launch checking, version.
// The boundary is declared in include/pi3slam_hip.h; nothing here depends on torch.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

void pi3_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pi3_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pi3_set_error("%s: %s", what, hipGetErrorString(e));
    return PI3_ERR_LAUNCH;
  }
  return PI3_OK;
}

int pi3_lds_optin(const void* kern, int bytes, unsigned long long* done_mask, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done_mask, __ATOMIC_RELAXED) & bit) return PI3_OK;
  hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    pi3_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d) on device %d: %s", what, bytes, dev,
                  hipGetErrorString(e));
    return PI3_ERR_LAUNCH;
  }
  __atomic_fetch_or(done_mask, bit, __ATOMIC_RELAXED);
  return PI3_OK;
}

extern "C" const char* pi3_last_error(void) { return g_err; }

extern "C" int pi3_abi_version(void) { return 4; }   // 4: pi3_sim3_umeyama_weighted (real-valued pair weights); 2: caller-provided workspaces (attention, group-norm statistics); 3: narrow-N GEMM / conv forms, pi3_cast_rows_pad, 4-channel granularity of the MoGe staging kernels

// Number of visible devices (does not create a context); used by the loader to fail loudly on a box without a GPU.
extern "C" int pi3_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
