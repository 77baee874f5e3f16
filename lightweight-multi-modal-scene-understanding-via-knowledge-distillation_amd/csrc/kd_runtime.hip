// kd_runtime.hip -- error reporting, version, and small shared launchers for the C-ABI library.
#include "kd_common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void kd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int kd_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    kd_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return (int)e;
  }
  return KD_OK;
}

namespace {
__global__ void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += slab[(int64_t)k * n + i];
  out[i] = s;
}
}  // namespace

int kd_slab_reduce_launch(const float* slab, int nsplit, int64_t n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, slab, nsplit, n, out);
  return kd_check_launch("kd_slab_reduce");
}

extern "C" {
int kd_version(void) { return 100; }
const char* kd_last_error_string(void) { return g_err; }
const char* kd_arch(void) { return "gfx950"; }
}
