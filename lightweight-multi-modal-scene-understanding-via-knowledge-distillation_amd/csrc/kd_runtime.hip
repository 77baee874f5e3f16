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
// out[i] = sum_k slab[k][i].  Block (64 elements, 16 split lanes): the split dimension is walked by
// 16 lanes in parallel and combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, int64_t n,
                                                           float* __restrict__ out) {
  __shared__ float sm[16][64];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  float s = 0.f;
  if (i < n)
    for (int k = threadIdx.y; k < nsplit; k += 16) s += slab[(int64_t)k * n + i];
  sm[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sm[k][threadIdx.x];
    out[i] = t;
  }
}
}  // namespace

int kd_slab_reduce_launch(const float* slab, int nsplit, int64_t n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64, 16), 0, st, slab, nsplit, n, out);
  return kd_check_launch("kd_slab_reduce");
}

extern "C" {
int kd_version(void) { return 100; }
const char* kd_last_error_string(void) { return g_err; }
const char* kd_arch(void) { return "gfx950"; }
}
