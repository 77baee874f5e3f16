// kd_common.h -- shared device helpers for the gfx950 (MI355X / CDNA4) KD kernels.
//
// Conventions used by every kernel in this directory
//   * activations are fp32 NHWC, addressed as a row-major matrix [M = B*H*W, C] with a row
//     stride `ld` (floats) so that a tensor may live inside a wider concat buffer;
//   * a "deferred" operand is a RAW conv output plus per-channel (scale, shift) and an
//     activation id: value = act(raw * scale + shift).  BatchNorm + activation are never
//     materialised between two convolutions -- the consumer applies them on load;
//   * BatchNorm batch statistics are produced by the conv kernel's epilogue as per-block
//     partial sums in a slab [rows][2][C] and reduced (in fp64, fixed order => deterministic)
//     by kd_bn_finalize_train.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#define KD_ACT_NONE 0
#define KD_ACT_RELU 1
#define KD_ACT_RELU6 2

#define KD_OK 0
#define KD_ERR_ARG (-1)
#define KD_ERR_ALIGN (-2)
#define KD_ERR_WORKSPACE (-3)
#define KD_ERR_SHAPE (-4)

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void kd_set_error(const char* fmt, ...);
int kd_check_launch(const char* what);

#define KD_REQUIRE(cond, code, ...)        \
  do {                                     \
    if (!(cond)) {                         \
      kd_set_error(__VA_ARGS__);           \
      return (code);                       \
    }                                      \
  } while (0)

// Kernels that need more than 64 KB of dynamic LDS opt in with hipFuncSetAttribute.  The attribute belongs to the
// (function, device) pair: `raised` keeps one bit per device, so a process that drives several devices raises the limit
// on each, and a failed call is reported instead of being remembered as done.
static inline hipError_t kd_raise_dynamic_lds(const void* fn, size_t bytes, std::atomic<uint64_t>& raised) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (raised.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) raised.fetch_or(bit, std::memory_order_release);
  return e;
}

static inline bool kd_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// z = raw*scale + shift, as ONE fused multiply-add: forward, backward masks and the scatter-max
// tie test all call this, so a recomputed value is bit-identical to the one used in the forward.
__device__ __forceinline__ float kd_affine(float raw, float sc, float sh) { return fmaf(raw, sc, sh); }

// Activations are a clamp to [lo, hi] with bounds SELECTED from the activation id (none: -inf..inf,
// ReLU: 0..inf, ReLU6: 0..6).  No branches: a runtime `if (act == ...)` around every element makes
// hipcc split the surrounding loads into one basic block each and drain the memory queue per load.
__device__ __forceinline__ float kd_act_lo(int act) { return act == KD_ACT_NONE ? -INFINITY : 0.f; }
__device__ __forceinline__ float kd_act_hi(int act) { return act == KD_ACT_RELU6 ? 6.f : INFINITY; }
__device__ __forceinline__ float kd_act(float z, int act) {
  // v_max_f32 / v_min_f32 (one instruction per bound; the compare + select form costs two).  Same values as
  // `z > lo ? z : lo` then `z < hi ? z : hi`, including NaN -> lo.
  return __builtin_fminf(__builtin_fmaxf(z, kd_act_lo(act)), kd_act_hi(act));
}
// derivative mask of kd_act at pre-activation z (ATen threshold_backward / hardtanh_backward:
// strict inequalities on both sides)
__device__ __forceinline__ float kd_act_mask(float z, int act) {
  return (z > kd_act_lo(act) && z < kd_act_hi(act)) ? 1.f : 0.f;
}

__device__ __forceinline__ float4 kd_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void kd_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 kd_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// streaming variants: tensors written once and read again only by a later kernel (gigabytes later) bypass the
// cache hierarchy's retention, leaving L2 / MALL to the operands that ARE re-read (weights, per-cell tables)
__device__ __forceinline__ void kd_st4_nt(float* p, float4 v) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p));
}

__device__ __forceinline__ float4 kd_affine_act4(float4 x, float4 sc, float4 sh, int act) {
  float4 r;
  r.x = kd_act(kd_affine(x.x, sc.x, sh.x), act);
  r.y = kd_act(kd_affine(x.y, sc.y, sh.y), act);
  r.z = kd_act(kd_affine(x.z, sc.z, sh.z), act);
  r.w = kd_act(kd_affine(x.w, sc.w, sh.w), act);
  return r;
}

// backward operand: dy_raw = al*(d*mask(z)) + be*x + ga with z = x*sc+sh  (al/be/ga from
// kd_bn_bwd_finalize).  With act == NONE the mask is 1 and sc/sh are not read.
__device__ __forceinline__ float kd_bwd_operand(float d, float x, float al, float be, float ga,
                                                float sc, float sh, int act) {
  float g = d * kd_act_mask(kd_affine(x, sc, sh), act);
  return fmaf(al, g, fmaf(be, x, ga));
}

// Workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is fence + s_barrier, and the fence makes hipcc
// drain vmcnt to 0 first: every global load still in flight (register prefetch) and, worse, every global STORE
// already issued is waited for at each barrier (~2-3 us per drained store burst in the GEMM epilogue).  Use this one
// wherever the barrier only protects an LDS image; threads here never communicate through global memory.
__device__ __forceinline__ void kd_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LiDAR point-MLP layer 0 (Conv1d 4 -> C, lidar_encoder.py:26): ONE evaluation order, used by the statistics pass and
// by every kernel that recomputes the layer from the 16-byte point instead of reading its [P, C] output from HBM, so a
// recomputed value is bit-identical to the one the BatchNorm statistics were taken over.
__device__ __forceinline__ float kd_l0_raw(float4 pt, float4 w, float b) {
  return fmaf(w.w, pt.w, fmaf(w.z, pt.z, fmaf(w.y, pt.y, fmaf(w.x, pt.x, b))));
}
__device__ __forceinline__ float4 kd_l0_raw4(float4 pt, const float4 (&w)[4], float4 b) {
  return make_float4(kd_l0_raw(pt, w[0], b.x), kd_l0_raw(pt, w[1], b.y), kd_l0_raw(pt, w[2], b.z), kd_l0_raw(pt, w[3], b.w));
}

__device__ __forceinline__ float kd_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// out[i] = sum_s slab[s][i], s in fixed order (deterministic); defined in kd_runtime.hip
int kd_nt_store(size_t bytes);       // 1: a tensor of this size should be stored with the non-temporal hint (KD_NT_STORE=0 disables)
int kd_slab_reduce_launch(const float* slab, int nsplit, int64_t n, float* out, hipStream_t st);
int kd_slab_reduce_tall_launch(float* slab, int rows, int64_t n, float* out, hipStream_t st);   // clobbers the slab

// Generic "channel-group x row-slot" thread layout used by the HBM-bound NHWC kernels:
// a thread owns 4 consecutive channels (one float4) and walks rows with a grid stride.
struct KdCgLayout {
  int groups;      // C / 4
  int slots;       // row slots per block = 256 / groups (>= 1)
  int grid;        // blocks
};
static inline KdCgLayout kd_cg_layout(int64_t rows, int C, int max_blocks = 2048) {
  KdCgLayout l;
  l.groups = C / 4;
  l.slots = 256 / l.groups;
  if (l.slots < 1) l.slots = 1;
  int64_t need = (rows + l.slots - 1) / l.slots;
  l.grid = (int)(need < max_blocks ? need : max_blocks);
  if (l.grid < 1) l.grid = 1;
  return l;
}
