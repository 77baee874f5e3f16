// kd_input.hip -- per-frame input preparation on the device (reference: src/data_loading/pandaset_dataset.py).
//   * remap_semantic (:13-20) + rasterize_bev (:23-45): raw PandaSet class ids -> {0,1}, then a BEV mask where
//     each cell takes the label of the FIRST point (input order) whose label is non-zero.  The reference walks
//     the points in a Python loop; here every labelled point does an atomicMin of its in-frame index on its
//     cell and a second kernel reads the winner's label: order-independent, bit-exact, one pass over the points.
//   * point stacking + zero padding / subsample gather (:113-127)
//   * uint8 HWC image -> float32 CHW / 255 (:108-111)
// All HBM-bound byte/integer work: coalesced streams, no LDS needed.
#include "kd_common.h"

namespace {

constexpr int32_t kEmpty = 0x7f7f7f7f;          // hipMemsetAsync(0x7f) sentinel: "no labelled point yet"

__device__ __forceinline__ int64_t label_of(int64_t id, int remap, uint64_t bits) {
  if (!remap) return id;
  return (id >= 0 && id < 64) ? (int64_t)((bits >> id) & 1ull) : 0;
}

// numpy float32 order of operations, no contraction, IEEE division: ((v - lo) / span * (n - 1)) -> trunc -> clip
__device__ __forceinline__ int axis_cell(float v, float lo, float span, int n) {
  const float t = __fmul_rn(__fdiv_rn(__fsub_rn(v, lo), span), (float)(n - 1));
  int c = (int)t;
  c = c < 0 ? 0 : c;
  return c > n - 1 ? n - 1 : c;
}

__device__ __forceinline__ int frame_of(const int64_t* __restrict__ off, int B, int64_t i) {
  int lo = 0, hi = B;                             // largest b with off[b] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

struct RastArgs {
  const float* x; const float* y; const int64_t* cls; const int64_t* off; int B; int64_t n;
  uint64_t bits; int remap; int H, W; float x0, xs, x1, y0, ys, y1; int32_t* first; int64_t* mask;
};

__global__ __launch_bounds__(256) void raster_claim_kernel(RastArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * 256) {
    const float x = a.x[i], y = a.y[i];
    const bool keep = x >= a.x0 && x <= a.x1 && y >= a.y0 && y <= a.y1;      // NaN fails every comparison
    if (!keep || label_of(a.cls[i], a.remap, a.bits) == 0) continue;
    const int b = frame_of(a.off, a.B, i);
    const int cell = axis_cell(y, a.y0, a.ys, a.H) * a.W + axis_cell(x, a.x0, a.xs, a.W);
    atomicMin(a.first + (int64_t)b * a.H * a.W + cell, (int32_t)(i - a.off[b]));
  }
}

__global__ __launch_bounds__(256) void raster_resolve_kernel(RastArgs a) {
  const int64_t ncell = (int64_t)a.B * a.H * a.W;
  for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < ncell; k += (int64_t)gridDim.x * 256) {
    const int32_t f = a.first[k];
    const int b = (int)(k / ((int64_t)a.H * a.W));
    a.mask[k] = f == kEmpty ? 0 : label_of(a.cls[a.off[b] + f], a.remap, a.bits);
  }
}

__global__ __launch_bounds__(256) void remap_kernel(const int64_t* __restrict__ raw, int64_t n, uint64_t bits,
                                                    int64_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = label_of(raw[i], 1, bits);
}

__global__ __launch_bounds__(256) void points_prepare_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ z, const float* __restrict__ w,
                                                             const int64_t* __restrict__ choice, int64_t n_take,
                                                             int64_t max_points, float* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < max_points; j += (int64_t)gridDim.x * 256) {
    float4 v = kd_zero4();
    if (j < n_take) {
      const int64_t s = choice ? choice[j] : j;
      v = make_float4(x[s], y[s], z[s], w[s]);
    }
    kd_st4(out + j * 4, v);
  }
}

__global__ __launch_bounds__(256) void image_chw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int HW) {
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const uint8_t* s = in + (int64_t)p * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(int64_t)c * HW + p] = __fdiv_rn((float)s[c], 255.f);
  }
}

int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

extern "C" {

size_t kd_bev_rasterize_ws_bytes(int B, int H, int W) { return (size_t)B * H * W * sizeof(int32_t); }

int kd_bev_rasterize(const float* x, const float* y, const int64_t* cls, const int64_t* offsets, int B, int64_t n_total,
                     int remap, uint64_t remap_bits, int H, int W, float x_min, float x_span, float x_max, float y_min,
                     float y_span, float y_max, void* ws, size_t ws_bytes, int64_t* mask, void* stream) {
  KD_REQUIRE(offsets && mask && ws && B > 0 && H > 0 && W > 0 && n_total >= 0, KD_ERR_ARG, "kd_bev_rasterize: bad args");
  KD_REQUIRE(n_total == 0 || (x && y && cls), KD_ERR_ARG, "kd_bev_rasterize: null point arrays with n_total=%lld", (long long)n_total);
  KD_REQUIRE(n_total < (int64_t)kEmpty, KD_ERR_SHAPE, "kd_bev_rasterize: too many points");
  KD_REQUIRE(x_span > 0.f && y_span > 0.f, KD_ERR_ARG, "kd_bev_rasterize: empty range");
  KD_REQUIRE(ws_bytes >= kd_bev_rasterize_ws_bytes(B, H, W), KD_ERR_WORKSPACE, "kd_bev_rasterize: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(ws, 0x7f, kd_bev_rasterize_ws_bytes(B, H, W), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_bev_rasterize: memset failed");
  RastArgs a{x, y, cls, offsets, B, n_total, remap_bits, remap, H, W, x_min, x_span, x_max, y_min, y_span, y_max,
             (int32_t*)ws, mask};
  if (n_total > 0) hipLaunchKernelGGL(raster_claim_kernel, dim3(grid_for(n_total)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(raster_resolve_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, st, a);
  return kd_check_launch("kd_bev_rasterize");
}

int kd_semantic_remap(const int64_t* raw, int64_t n, uint64_t remap_bits, int64_t* out, void* stream) {
  KD_REQUIRE(n >= 0 && (n == 0 || (raw && out)), KD_ERR_ARG, "kd_semantic_remap: bad args");
  if (n == 0) return KD_OK;
  hipLaunchKernelGGL(remap_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, raw, n, remap_bits, out);
  return kd_check_launch("kd_semantic_remap");
}

int kd_points_prepare(const float* x, const float* y, const float* z, const float* intensity, const int64_t* choice,
                      int64_t n, int64_t max_points, float* out, void* stream) {
  KD_REQUIRE(out && max_points > 0 && n >= 0, KD_ERR_ARG, "kd_points_prepare: bad args");
  KD_REQUIRE(n == 0 || (x && y && z && intensity), KD_ERR_ARG, "kd_points_prepare: null coordinate arrays");
  KD_REQUIRE(kd_aligned16(out), KD_ERR_ALIGN, "kd_points_prepare: out must be 16-byte aligned");
  KD_REQUIRE(!choice || n >= max_points, KD_ERR_ARG, "kd_points_prepare: a subsample needs n >= max_points");
  const int64_t take = choice ? max_points : (n < max_points ? n : max_points);
  hipLaunchKernelGGL(points_prepare_kernel, dim3(grid_for(max_points)), dim3(256), 0, (hipStream_t)stream, x, y, z,
                     intensity, choice, take, max_points, out);
  return kd_check_launch("kd_points_prepare");
}

int kd_image_u8hwc_to_f32chw(const uint8_t* in, float* out, int H, int W, void* stream) {
  KD_REQUIRE(in && out && H > 0 && W > 0, KD_ERR_ARG, "kd_image_u8hwc_to_f32chw: bad args");
  hipLaunchKernelGGL(image_chw_kernel, dim3(grid_for((int64_t)H * W)), dim3(256), 0, (hipStream_t)stream, in, out, H * W);
  return kd_check_launch("kd_image_u8hwc_to_f32chw");
}

}  // extern "C"
