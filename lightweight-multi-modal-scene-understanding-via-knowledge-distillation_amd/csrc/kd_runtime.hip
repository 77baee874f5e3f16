// kd_runtime.hip -- error reporting, version, and small shared launchers for the C-ABI library.
#include "kd_common.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

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
// out[i] = sum_k slab[k][i].  The split dimension is walked by 16 lanes in parallel (lane y adds rows y, y + 16, ... in order)
// and the 16 partial sums are combined through LDS in lane order: deterministic, and the same order for both forms below.
// Round 4: where n is a multiple of 4 (every weight matrix) a thread owns FOUR consecutive elements (16-byte loads, four rows in
// flight): the dword form ran the 33 reductions of a KD step at ~1.9 TB/s (13-18 us each; 11.6 % of the kernel time of a
// B = 4 step).  Same sums, bit for bit.
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
// block (16 element quads, LY split lanes), 64 consecutive elements.  LY = 16 everywhere except tall, narrow slabs (the depthwise /
// layer-0 / fusion weight gradients: 2048 rows of a few hundred to a few thousand elements, i.e. a handful of workgroups each walking
// 128 rows per lane: 15-19 us per call in the r04 trace), which take 64 lanes.  The summation order depends on LY, LY only on (nsplit, n).
template <int LY>
__global__ __launch_bounds__(16 * LY) void slab_reduce4_kernel(const float* __restrict__ slab, int nsplit, int64_t n,
                                                               float* __restrict__ out) {
  __shared__ float4 sm[LY][16];
  const int64_t i = ((int64_t)blockIdx.x * 16 + threadIdx.x) * 4;
  float4 s = kd_zero4();
  if (i < n) {
    int k = threadIdx.y;
    for (; k + 3 * LY < nsplit; k += 4 * LY) {               // four rows of this lane in flight, added in row order
      const float4 a = kd_ld4(slab + (int64_t)k * n + i), b = kd_ld4(slab + (int64_t)(k + LY) * n + i);
      const float4 c = kd_ld4(slab + (int64_t)(k + 2 * LY) * n + i), d = kd_ld4(slab + (int64_t)(k + 3 * LY) * n + i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
      s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < nsplit; k += LY) {
      const float4 a = kd_ld4(slab + (int64_t)k * n + i);
      s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
  }
  sm[threadIdx.y][threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.y == 0 && i < n) {
    float4 t = kd_zero4();
#pragma unroll 16
    for (int k = 0; k < LY; ++k) { const float4 v = sm[k][threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    kd_st4(out + i, t);
  }
}
// Tall slabs (one row per GEMM tile: tens of thousands of rows): groups of ~sqrt(rows) rows are first summed, in
// double, into the group's first row -- grid (ceil(n/64), groups), block (64, 4) -- so the final pass walks
// rows/group rows instead of serialising the whole column in a handful of workgroups.  The slab is scratch.
__global__ void slab_group_sum_kernel(float* slab, int rows, int64_t n, int group) {
  __shared__ double sm[4][64];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int r0 = blockIdx.y * group;
  const int r1 = r0 + group < rows ? r0 + group : rows;
  double a = 0.0;
  if (i < n)
    for (int r = r0 + threadIdx.y; r < r1; r += 4) a += (double)slab[(int64_t)r * n + i];
  sm[threadIdx.y][threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.y == 0 && i < n)
    slab[(int64_t)r0 * n + i] = (float)(sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
__global__ void slab_strided_sum_kernel(const float* __restrict__ slab, int ngroups, int group, int64_t n, float* __restrict__ out) {
  __shared__ double sm[4][64];
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  double a = 0.0;
  if (i < n)
    for (int k = threadIdx.y; k < ngroups; k += 4) a += (double)slab[(int64_t)k * group * n + i];
  sm[threadIdx.y][threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.y == 0 && i < n) out[i] = (float)(sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}
}  // namespace

// Outputs of >= 64 MB are consumed by a later kernel long after they have left L2 / MALL: storing them non-temporally
// keeps the caches for what IS re-read (weights, per-cell tables, the other operand's rows).  Measured +1.1 % on the
// KD step from the GEMM outputs alone.
int kd_nt_store(size_t bytes) {
  static const int mode = [] { const char* e = getenv("KD_NT_STORE"); return e ? atoi(e) : 1; }();
  return mode && bytes >= ((size_t)64 << 20);
}

int kd_slab_reduce_tall_launch(float* slab, int rows, int64_t n, float* out, hipStream_t st) {
  int group = 1;
  while ((int64_t)group * group < rows) ++group;
  const int ngroups = (rows + group - 1) / group;
  const unsigned nx = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(slab_group_sum_kernel, dim3(nx, ngroups), dim3(64, 4), 0, st, slab, rows, n, group);
  hipLaunchKernelGGL(slab_strided_sum_kernel, dim3(nx), dim3(64, 4), 0, st, slab, ngroups, group, n, out);
  return kd_check_launch("kd_slab_reduce_tall");
}

int kd_slab_reduce_launch(const float* slab, int nsplit, int64_t n, float* out, hipStream_t st) {
  if (n % 4 == 0 && kd_aligned16(slab) && kd_aligned16(out)) {
    if (nsplit >= 1024 && n <= 16384)
      hipLaunchKernelGGL(slab_reduce4_kernel<64>, dim3((unsigned)((n + 63) / 64)), dim3(16, 64), 0, st, slab, nsplit, n, out);
    else
      hipLaunchKernelGGL(slab_reduce4_kernel<16>, dim3((unsigned)((n + 63) / 64)), dim3(16, 16), 0, st, slab, nsplit, n, out);
  }
  else
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64, 16), 0, st, slab, nsplit, n, out);
  return kd_check_launch("kd_slab_reduce");
}

namespace {
struct CopySegs { const float* s[4]; float* d[4]; long long n[4]; long long first[5]; };   // first[i]: first 256-element block of segment i
__global__ __launch_bounds__(256) void copy_segments_kernel(CopySegs c) {
  int k = 0;
  while (k < 3 && (long long)blockIdx.x >= c.first[k + 1]) ++k;
  const long long i = ((long long)blockIdx.x - c.first[k]) * 256 + threadIdx.x;
  if (i < c.n[k]) c.d[k][i] = c.s[k][i];
}
}  // namespace

extern "C" {
// Up to four small device-to-device copies in ONE launch (packed parameter gradients -> their slots in the flat gradient buffer:
// a kernel that produces several parameters' gradients in one workspace hands them over without a copy launch per parameter).
// Unused segments: n = 0.
int kd_copy_segments(const float* s0, float* d0, int64_t n0, const float* s1, float* d1, int64_t n1, const float* s2, float* d2,
                     int64_t n2, const float* s3, float* d3, int64_t n3, void* stream) {
  CopySegs c{{s0, s1, s2, s3}, {d0, d1, d2, d3}, {n0, n1, n2, n3}, {0, 0, 0, 0, 0}};
  for (int k = 0; k < 4; ++k) {
    KD_REQUIRE(c.n[k] >= 0 && (c.n[k] == 0 || (c.s[k] && c.d[k])), KD_ERR_ARG, "kd_copy_segments: segment %d", k);
    c.first[k + 1] = c.first[k] + (c.n[k] + 255) / 256;
  }
  KD_REQUIRE(c.first[4] > 0, KD_ERR_ARG, "kd_copy_segments: nothing to copy");
  hipLaunchKernelGGL(copy_segments_kernel, dim3((unsigned)c.first[4]), dim3(256), 0, (hipStream_t)stream, c);
  return kd_check_launch("kd_copy_segments");
}

int kd_version(void) { return 100; }
const char* kd_last_error_string(void) { return g_err; }
const char* kd_arch(void) { return "gfx950"; }
}
