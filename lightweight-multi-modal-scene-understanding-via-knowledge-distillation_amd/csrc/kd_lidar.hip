// kd_lidar.hip -- LiDAR branch kernels that are not plain GEMMs (lidar_encoder.py:42-99):
//   * point_mlp layer 0 (Conv1d 4->64, k=1, bias): K=4 is VALU work, not MFMA work;
//   * BEV binning + scatter-max of the activated 128-ch point features into the [B,H,W,C] grid
//     (== zeros.scatter_reduce_(amax, include_self=False) because post-ReLU values are >= 0, so an
//     unsigned-integer atomicMax on the fp32 bit pattern into a zero grid is exact and
//     order-independent => bitwise deterministic);
//   * its backward with ATen's even tie-split rule.
// Layers 1 and 2 of the point MLP (64->128->128) go through kd_pwconv_gemm with M = B*N points.
#include "kd_common.h"

namespace {

struct BevGeom { float x0, xd, y0, yd; int H, W; };

// lidar_encoder.py:46-55,69-71 -- arithmetic kept in the reference's order, no contraction
__device__ __forceinline__ bool bev_cell(const float4 pt, const BevGeom g, int& cell) {
  const float xn = __fdiv_rn(__fsub_rn(pt.x, g.x0), g.xd);
  const float yn = __fdiv_rn(__fsub_rn(pt.y, g.y0), g.yd);
  const bool valid = (xn >= 0.f) && (xn <= 1.f) && (yn >= 0.f) && (yn <= 1.f);   // NaN => false
  if (!valid) { cell = -1; return false; }
  long long ix = (long long)__fmul_rn(xn, (float)(g.W - 1));
  long long iy = (long long)__fmul_rn(yn, (float)(g.H - 1));
  ix = ix < 0 ? 0 : (ix > g.W - 1 ? g.W - 1 : ix);
  iy = iy < 0 ? 0 : (iy > g.H - 1 ? g.H - 1 : iy);
  cell = (int)(iy * g.W + ix);
  return true;
}

// ---- layer 0: y[p, co] = b[co] + sum_i w[co, i] * pt[p, i] -------------------------------------
struct L0Args {
  const float* pts; const float* w; const float* b; float* y; float* partial;
  int64_t P; int C; int groups, slots; const int* p_dev;
};
__global__ __launch_bounds__(256) void lidar_l0_fwd_kernel(L0Args a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 wr[4], bias = kd_zero4();
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = kd_ld4(a.w + (c0 + j) * 4);
    if (a.b) bias = kd_ld4(a.b + c0);
  }
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (a.p_dev) { const int64_t pv = *a.p_dev; a.P = pv < a.P ? pv : a.P; }
  if (active) {
    // Four points of this thread in flight (round 4): one dependent 16-byte load per ~30 VALU operations left the kernel 82 % of its
    // wave cycles waiting (profiles/r03_gemm_pmc_utilisation.txt).  The sums still receive the points in the same order.
    const int64_t stride = (int64_t)gridDim.x * a.slots;
    auto one = [&](int64_t p, float4 pt) __attribute__((always_inline)) {
      const float4 v = kd_l0_raw4(pt, wr, bias);
      if (a.y) kd_st4(a.y + p * a.C + c0, v);          // y == NULL: statistics only, consumers recompute the layer
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    };
    int64_t p = (int64_t)blockIdx.x * a.slots + slot;
    for (; p + 3 * stride < a.P; p += 4 * stride) {
      const float4 q0 = kd_ld4(a.pts + p * 4), q1 = kd_ld4(a.pts + (p + stride) * 4), q2 = kd_ld4(a.pts + (p + 2 * stride) * 4),
                   q3 = kd_ld4(a.pts + (p + 3 * stride) * 4);
      one(p, q0); one(p + stride, q1); one(p + 2 * stride, q2); one(p + 3 * stride, q3);
    }
    for (; p < a.P; p += stride) one(p, kd_ld4(a.pts + p * 4));
  }
  if (a.partial) {
    kd_st4(red + tid * 4, s1);
    kd_st4(red + 1024 + tid * 4, s2);
    __syncthreads();
    for (int i = tid; i < 2 * a.C; i += 256) {
      const int st = i / a.C, c = i % a.C;
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[st * 1024 + (sl * a.groups + c / 4) * 4 + (c & 3)];
      a.partial[((int64_t)blockIdx.x * 2 + st) * a.C + c] = s;
    }
  }
}

// layer-0 backward: dw[co,i] = sum_p dyeff[p,co]*pt[p,i]; db[co] = sum_p dyeff[p,co]
struct L0BwdArgs {
  const float* D; const float* Y; const float* al; const float* be; const float* ga;
  const float* pts; float* slab;     // slab [grid][C*5]: dw (C*4) then db (C)
  int64_t P; int C; int groups, slots;
  const float* w; const float* b;    // Y == NULL: the layer output is recomputed from the point
};
__global__ __launch_bounds__(256) void lidar_l0_bwd_kernel(L0BwdArgs a) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 acc[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) acc[t] = kd_zero4();
  if (active) {
    const float4 al = kd_ld4(a.al + c0), be = kd_ld4(a.be + c0), ga = kd_ld4(a.ga + c0);
    float4 wr[4], bias = kd_zero4();
    if (!a.Y) {
#pragma unroll
      for (int j = 0; j < 4; ++j) wr[j] = kd_ld4(a.w + (c0 + j) * 4);
      if (a.b) bias = kd_ld4(a.b + c0);
    }
    const int64_t stride = (int64_t)gridDim.x * a.slots;
    // (four points in flight, accumulated in the same order: see lidar_l0_fwd_kernel)
    auto one = [&](int64_t p, float4 pt) __attribute__((always_inline)) {
      // D == NULL: the gradient-dependent part comes from kd_lidar_l1_dgrad's moments; this pass adds sum (be*y + ga) * pt
      const float4 d = a.D ? kd_ld4(a.D + p * a.C + c0) : kd_zero4();
      const float4 y = a.Y ? kd_ld4(a.Y + p * a.C + c0) : kd_l0_raw4(pt, wr, bias);
      float4 g;
      g.x = kd_bwd_operand(d.x, y.x, al.x, be.x, ga.x, 0.f, 0.f, KD_ACT_NONE);
      g.y = kd_bwd_operand(d.y, y.y, al.y, be.y, ga.y, 0.f, 0.f, KD_ACT_NONE);
      g.z = kd_bwd_operand(d.z, y.z, al.z, be.z, ga.z, 0.f, 0.f, KD_ACT_NONE);
      g.w = kd_bwd_operand(d.w, y.w, al.w, be.w, ga.w, 0.f, 0.f, KD_ACT_NONE);
      const float pv[4] = {pt.x, pt.y, pt.z, pt.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i].x = fmaf(g.x, pv[i], acc[i].x); acc[i].y = fmaf(g.y, pv[i], acc[i].y);
        acc[i].z = fmaf(g.z, pv[i], acc[i].z); acc[i].w = fmaf(g.w, pv[i], acc[i].w);
      }
      acc[4].x += g.x; acc[4].y += g.y; acc[4].z += g.z; acc[4].w += g.w;
    };
    int64_t p = (int64_t)blockIdx.x * a.slots + slot;
    for (; p + 3 * stride < a.P; p += 4 * stride) {
      const float4 q0 = kd_ld4(a.pts + p * 4), q1 = kd_ld4(a.pts + (p + stride) * 4), q2 = kd_ld4(a.pts + (p + 2 * stride) * 4),
                   q3 = kd_ld4(a.pts + (p + 3 * stride) * 4);
      one(p, q0); one(p + stride, q1); one(p + 2 * stride, q2); one(p + 3 * stride, q3);
    }
    for (; p < a.P; p += stride) one(p, kd_ld4(a.pts + p * 4));
  }
  for (int t = 0; t < 5; ++t) {
    __syncthreads();
    kd_st4(red + tid * 4, acc[t]);
    __syncthreads();
    for (int c = tid; c < a.C; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      const int64_t o = (int64_t)blockIdx.x * a.C * 5;
      if (t < 4) a.slab[o + c * 4 + t] = s; else a.slab[o + a.C * 4 + c] = s;
    }
  }
}

// ---- scatter-max ---------------------------------------------------------------------------------
struct ScatArgs {
  const float* pts; const float* y; const float* sc; const float* sh; int act;   // deferred [P, C]
  float* grid;                      // [B, H*W, C], zero-initialised by the caller entry point
  unsigned* cnt;                    // bwd: tie counts [B*H*W*C]
  const float* dout;                // bwd: dL/dgrid [B, H*W, C]
  const float* mean; const float* invstd;
  float* G; float* partial;
  int B; int64_t N; int C; BevGeom geo; int groups, slots; int pass;
};

__global__ __launch_bounds__(256) void scatter_max_fwd_kernel(ScatArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  // Lane -> channel map for atomics: thread gidx owns channels gidx, gidx+G, gidx+2G, gidx+3G (G = C/4), so
  // one atomic wave-instruction covers CONSECUTIVE words (whole 128-B segments) of a cell row instead of
  // every 4th word of all its lines -- the shape the memory-side atomic units run at full rate.
  const int G = a.groups;
  float sc[4], sh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sc[j] = a.sc[gidx + G * j]; sh[j] = a.sh[gidx + G * j]; }
  const int64_t P = (int64_t)a.B * a.N;
  const int HW = a.geo.H * a.geo.W;
  for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < P; p += (int64_t)gridDim.x * a.slots) {
    int cell;
    if (!bev_cell(kd_ld4(a.pts + p * 4), a.geo, cell)) continue;
    const float* yp = a.y + p * a.C + gidx;
    unsigned* dst = reinterpret_cast<unsigned*>(a.grid + ((p / a.N) * HW + cell) * a.C + gidx);
    float v[4];
    unsigned cur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = kd_act(kd_affine(yp[G * j], sc[j], sh[j]), a.act); cur[j] = dst[G * j]; }
    // The cell value only ever grows, so a (possibly stale) plain read that is already >= v proves
    // the atomic would be a no-op: ~12 points share a cell, most of them lose and skip the atomic.
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned u = __float_as_uint(v[j]);
      if (v[j] > 0.f && u > cur[j]) atomicMax(dst + G * j, u);
    }
  }
}

// pass 0: count ties per (cell, channel); pass 1: G = tie ? dout/cnt : 0, plus BN-backward sums.
__global__ __launch_bounds__(256) void scatter_max_bwd_kernel(ScatArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    const float4 sc = kd_ld4(a.sc + c0), sh = kd_ld4(a.sh + c0);
    float4 mu = kd_zero4(), inv = kd_zero4();
    if (a.pass == 1) { mu = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const int64_t P = (int64_t)a.B * a.N;
    const int HW = a.geo.H * a.geo.W;
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < P; p += (int64_t)gridDim.x * a.slots) {
      int cell;
      const bool valid = bev_cell(kd_ld4(a.pts + p * 4), a.geo, cell);
      float4 g = kd_zero4();
      if (valid) {
        const float4 yr = kd_ld4(a.y + p * a.C + c0);
        const float4 v = kd_affine_act4(yr, sc, sh, a.act);
        const int64_t o = ((p / a.N) * HW + cell) * a.C + c0;
        const float4 mx = kd_ld4(a.grid + o);
        const bool hx = v.x > 0.f && v.x == mx.x, hy = v.y > 0.f && v.y == mx.y;
        const bool hz = v.z > 0.f && v.z == mx.z, hw = v.w > 0.f && v.w == mx.w;
        if (a.pass == 0) {
          if (hx) atomicAdd(a.cnt + o + 0, 1u);
          if (hy) atomicAdd(a.cnt + o + 1, 1u);
          if (hz) atomicAdd(a.cnt + o + 2, 1u);
          if (hw) atomicAdd(a.cnt + o + 3, 1u);
        } else if (hx || hy || hz || hw) {
          const float4 d = kd_ld4(a.dout + o);
          const uint4 n = *reinterpret_cast<const uint4*>(a.cnt + o);
          if (hx) g.x = d.x / (float)n.x;
          if (hy) g.y = d.y / (float)n.y;
          if (hz) g.z = d.z / (float)n.z;
          if (hw) g.w = d.w / (float)n.w;
          s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
          s2.x = fmaf(g.x, (yr.x - mu.x) * inv.x, s2.x);
          s2.y = fmaf(g.y, (yr.y - mu.y) * inv.y, s2.y);
          s2.z = fmaf(g.z, (yr.z - mu.z) * inv.z, s2.z);
          s2.w = fmaf(g.w, (yr.w - mu.w) * inv.w, s2.w);
        }
      }
      if (a.pass == 1) kd_st4(a.G + p * a.C + c0, g);
    }
  }
  if (a.pass == 1) {
    kd_st4(red + tid * 4, s1);
    kd_st4(red + 1024 + tid * 4, s2);
    __syncthreads();
    for (int i = tid; i < 2 * a.C; i += 256) {
      const int st = i / a.C, c = i % a.C;
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[st * 1024 + (sl * a.groups + c / 4) * 4 + (c & 3)];
      a.partial[((int64_t)blockIdx.x * 2 + st) * a.C + c] = s;
    }
  }
}

}  // namespace

extern "C" {

int kd_lidar_l0_fwd(const float* pts, const float* w, const float* b, float* y, float* partial, int64_t P, int C,
                    const int* p_dev, void* stream) {
  KD_REQUIRE(pts && w && (y || partial) && P > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_lidar_l0_fwd: bad args");
  const KdCgLayout l = kd_cg_layout(P, C);
  L0Args a{pts, w, b, y, partial, P, C, l.groups, l.slots, p_dev};
  hipLaunchKernelGGL(lidar_l0_fwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_lidar_l0_fwd");
}

size_t kd_lidar_l0_bwd_ws_bytes(int64_t P, int C) { return (size_t)kd_cg_layout(P, C).grid * C * 5 * sizeof(float); }

// dwb: [C*4] weight gradient followed by [C] bias gradient.  Y == NULL: the layer output is recomputed from (w, b).
int kd_lidar_l0_bwd(const float* D, const float* Y, const float* w, const float* b, const float* al, const float* be,
                    const float* ga, const float* pts, float* dwb, int64_t P, int C, void* ws, size_t ws_bytes,
                    void* stream) {
  KD_REQUIRE((D || !Y) && (Y || w) && al && be && ga && pts && dwb && ws && P > 0 && C % 4 == 0, KD_ERR_ARG, "kd_lidar_l0_bwd: bad args");
  const KdCgLayout l = kd_cg_layout(P, C);
  KD_REQUIRE(ws_bytes >= (size_t)l.grid * C * 5 * sizeof(float), KD_ERR_WORKSPACE, "kd_lidar_l0_bwd: workspace too small");
  L0BwdArgs a{D, Y, al, be, ga, pts, (float*)ws, P, C, l.groups, l.slots, w, b};
  hipLaunchKernelGGL(lidar_l0_bwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  int rc = kd_check_launch("kd_lidar_l0_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, l.grid, (int64_t)C * 5, dwb, (hipStream_t)stream);
}

// grid[B,H,W,C] = scatter-max of act(y*sc+sh) over the valid points; zeroes `grid` first.
int kd_lidar_scatter_max_fwd(const float* pts, const float* y, const float* sc, const float* sh, int act,
                             float* grid, int B, int64_t N, int C, int H, int W, float x0, float x1, float y0,
                             float y1, void* stream) {
  KD_REQUIRE(pts && y && sc && sh && grid && B > 0 && N > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_lidar_scatter_max_fwd: bad args");
  KD_REQUIRE(act == KD_ACT_RELU || act == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_scatter_max_fwd: needs a non-negative activation");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grid, 0, (size_t)B * H * W * C * sizeof(float), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_scatter_max_fwd: memset failed: %s", hipGetErrorString(e));
  const KdCgLayout l = kd_cg_layout((int64_t)B * N, C, 4096);
  ScatArgs a{pts, y, sc, sh, act, grid, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, N, C,
             BevGeom{x0, x1 - x0, y0, y1 - y0, H, W}, l.groups, l.slots, 0};
  hipLaunchKernelGGL(scatter_max_fwd_kernel, dim3(l.grid), dim3(256), 0, st, a);
  return kd_check_launch("kd_lidar_scatter_max_fwd");
}

int64_t kd_lidar_scatter_stat_rows(int64_t P, int C) { return kd_cg_layout(P, C, 4096).grid; }
size_t kd_lidar_scatter_bwd_ws_bytes(int B, int H, int W, int C) { return (size_t)B * H * W * C * sizeof(unsigned); }

// G[P,C] = dL/d(act(bn(y))) * relu'(.)  under the even tie-split rule; partial = (sum G, sum G*xhat).
int kd_lidar_scatter_max_bwd(const float* pts, const float* y, const float* sc, const float* sh, int act,
                             const float* grid, const float* dout, const float* mean, const float* invstd, float* G,
                             float* partial, int B, int64_t N, int C, int H, int W, float x0, float x1, float y0,
                             float y1, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(pts && y && sc && sh && grid && dout && mean && invstd && G && partial && ws, KD_ERR_ARG, "kd_lidar_scatter_max_bwd: bad args");
  const size_t need = (size_t)B * H * W * C * sizeof(unsigned);
  KD_REQUIRE(ws_bytes >= need, KD_ERR_WORKSPACE, "kd_lidar_scatter_max_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(ws, 0, need, st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_scatter_max_bwd: memset failed: %s", hipGetErrorString(e));
  const KdCgLayout l = kd_cg_layout((int64_t)B * N, C, 4096);
  ScatArgs a{pts, y, sc, sh, act, const_cast<float*>(grid), (unsigned*)ws, dout, mean, invstd, G, partial, B, N, C,
             BevGeom{x0, x1 - x0, y0, y1 - y0, H, W}, l.groups, l.slots, 0};
  hipLaunchKernelGGL(scatter_max_bwd_kernel, dim3(l.grid), dim3(256), 0, st, a);
  a.pass = 1;
  hipLaunchKernelGGL(scatter_max_bwd_kernel, dim3(l.grid), dim3(256), 0, st, a);
  return kd_check_launch("kd_lidar_scatter_max_bwd");
}

// cell[p] = flat BEV cell index (iy*W+ix) or -1 for invalid points: exposed for the integer parity tests.
__global__ void bev_index_kernel(const float* pts, int* cell, int64_t P, BevGeom g) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int c;
  bev_cell(kd_ld4(pts + p * 4), g, c);
  cell[p] = c;
}
int kd_lidar_bev_index(const float* pts, int* cell, int64_t P, int H, int W, float x0, float x1, float y0, float y1,
                       void* stream) {
  KD_REQUIRE(pts && cell && P > 0, KD_ERR_ARG, "kd_lidar_bev_index: bad args");
  hipLaunchKernelGGL(bev_index_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, cell,
                     P, BevGeom{x0, x1 - x0, y0, y1 - y0, H, W});
  return kd_check_launch("kd_lidar_bev_index");
}

// ---- inference-only fast path -------------------------------------------------------------------
// In eval mode (running BatchNorm statistics) a point outside the BEV range influences nothing:
// it is never scattered and there are no batch statistics for it to perturb.  So the frozen
// teacher compacts the in-range points (~62 % on the synthetic / ~the same on PandaSet) once and
// runs its point MLP only on those.  Compaction order is arbitrary (atomic append) -- scatter-max
// is order-independent, so the result is still bitwise deterministic.
__global__ void lidar_compact_kernel(const float* __restrict__ pts, float* __restrict__ out_pts, int* __restrict__ out_cell,
                                     int* __restrict__ counter, int64_t P, int64_t N, BevGeom g) {
  // ONE global atomic per 1024-thread workgroup (a single counter word sustains only ~88 atomics/us):
  // ballot + popcount inside each wave, wave totals through LDS, lane 0 of the block reserves the range.
  __shared__ int wave_cnt[16], wave_off[16], block_base;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float4 pt = kd_zero4();
  int c = -1;
  bool valid = false;
  if (p < P) { pt = kd_ld4(pts + p * 4); valid = bev_cell(pt, g, c); }
  const unsigned long long m = __ballot(valid);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_cnt[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { wave_off[w] = tot; tot += wave_cnt[w]; }
    block_base = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  if (valid) {
    const int idx = block_base + wave_off[wave] + before;
    kd_st4(out_pts + (int64_t)idx * 4, pt);
    out_cell[idx] = (int)(p / N) * (g.H * g.W) + c;
  }
}
int kd_lidar_compact(const float* pts, float* out_pts, int* out_cell, int* counter, int B, int64_t N, int H, int W,
                     float x0, float x1, float y0, float y1, void* stream) {
  KD_REQUIRE(pts && out_pts && out_cell && counter && B > 0 && N > 0, KD_ERR_ARG, "kd_lidar_compact: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(counter, 0, sizeof(int), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_compact: memset failed: %s", hipGetErrorString(e));
  const int64_t P = (int64_t)B * N;
  hipLaunchKernelGGL(lidar_compact_kernel, dim3((unsigned)((P + 1023) / 1024)), dim3(1024), 0, st, pts, out_pts, out_cell,
                     counter, P, N, BevGeom{x0, x1 - x0, y0, y1 - y0, H, W});
  return kd_check_launch("kd_lidar_compact");
}

// scatter-max over pre-binned points: cell_idx[p] is the flat (batch, cell) row of `grid`
__global__ __launch_bounds__(256) void scatter_max_idx_kernel(const float* __restrict__ y, const float* __restrict__ sc,
                                                              const float* __restrict__ sh, int act,
                                                              const int* __restrict__ cell_idx, float* grid, int64_t P, int C,
                                                              int groups, int slots, const int* p_dev) {
  const int tid = threadIdx.x;
  const int gidx = tid % groups, slot = tid / groups;
  if (slot >= slots) return;
  const int G = groups;                 // same consecutive-word lane map as scatter_max_fwd_kernel
  float s[4], h[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s[j] = sc[gidx + G * j]; h[j] = sh[gidx + G * j]; }
  if (p_dev) { const int64_t pv = *p_dev; P = pv < P ? pv : P; }
  for (int64_t p = (int64_t)blockIdx.x * slots + slot; p < P; p += (int64_t)gridDim.x * slots) {
    const float* yp = y + p * C + gidx;
    unsigned* dst = reinterpret_cast<unsigned*>(grid + (int64_t)cell_idx[p] * C + gidx);
    float v[4];
    unsigned cur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = kd_act(kd_affine(yp[G * j], s[j], h[j]), act); cur[j] = dst[G * j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned u = __float_as_uint(v[j]);
      if (v[j] > 0.f && u > cur[j]) atomicMax(dst + G * j, u);
    }
  }
}
int kd_lidar_scatter_max_idx_fwd(const float* y, const float* sc, const float* sh, int act, const int* cell_idx,
                                 float* grid, int64_t P, int C, int64_t ncells, const int* p_dev, void* stream) {
  KD_REQUIRE(y && sc && sh && cell_idx && grid && P >= 0 && C % 4 == 0 && C <= 1024 && ncells > 0, KD_ERR_ARG, "kd_lidar_scatter_max_idx_fwd: bad args");
  KD_REQUIRE(act == KD_ACT_RELU || act == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_scatter_max_idx_fwd: needs a non-negative activation");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grid, 0, (size_t)ncells * C * sizeof(float), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_scatter_max_idx_fwd: memset failed: %s", hipGetErrorString(e));
  if (P == 0) return KD_OK;
  const KdCgLayout l = kd_cg_layout(P, C, 4096);
  hipLaunchKernelGGL(scatter_max_idx_kernel, dim3(l.grid), dim3(256), 0, st, y, sc, sh, act, cell_idx, grid, P, C, l.groups,
                     l.slots, p_dev);
  return kd_check_launch("kd_lidar_scatter_max_idx_fwd");
}

}  // extern "C"

// ---- cell-sorted (segmented) scatter-max: the training path --------------------------------------
// The atomic scatter above costs an atomic RMW stream forward and two more sweeps (tie count, then
// gradient) backward.  Training instead bins the point ids by grid row once per step (counting sort:
// histogram with the atomic's return value as the in-cell rank, exclusive scan, fill) and then walks
// each cell's points with ONE wave: the max, the tie count and the gradient split are plain register
// work, every grid row is written exactly once (no memset, no atomics) and the tie counts never touch
// memory.  The in-cell order is arbitrary (atomic rank) but max, tie count and the per-point gradient do
// not depend on it, and the BN-backward sums see at most one distinct term per (cell, channel) (tied
// holders contribute identical terms), so results stay bitwise run-to-run deterministic.
namespace {

constexpr int SCAN_CHUNK = 2048;   // elements per 256-thread block (8 per thread)

__global__ void seg_count_kernel(const float* __restrict__ pts, int* __restrict__ row_of_point, int* __restrict__ rank,
                                 int* counts, int64_t P, int64_t N, BevGeom g) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  int c;
  if (bev_cell(kd_ld4(pts + p * 4), g, c)) {
    const int row = (int)(p / N) * (g.H * g.W) + c;
    row_of_point[p] = row;
    rank[p] = atomicAdd(counts + row, 1);
  } else {
    row_of_point[p] = -1;
  }
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const int* __restrict__ v, int* __restrict__ bsum, int64_t n) {
  __shared__ int red[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  int s = 0;
  for (int i = threadIdx.x; i < SCAN_CHUNK; i += 256)
    if (base + i < n) s += v[base + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}

// exclusive scan of the block sums, in place; one 1024-thread block walks them in chunks with a carry
__global__ __launch_bounds__(1024) void scan_bsums_kernel(int* bsum, int nb) {
  __shared__ int sh[1024];
  int carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const int v = i < nb ? bsum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const int t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) bsum[i] = carry + sh[threadIdx.x] - v;
    carry += sh[1023];
    __syncthreads();
  }
}

// v[i] <- exclusive prefix of v (in place); thread t of block b owns the 8 consecutive elements at b*2048 + t*8
__global__ __launch_bounds__(256) void scan_apply_kernel(int* v, const int* __restrict__ bsum, int64_t n) {
  __shared__ int sh[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK + (int64_t)threadIdx.x * 8;
  int x[8];
  int tot = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { x[j] = base + j < n ? v[base + j] : 0; tot += x[j]; }
  sh[threadIdx.x] = tot;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int t = (int)threadIdx.x >= off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int run = bsum[blockIdx.x] + sh[threadIdx.x] - tot;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (base + j < n) v[base + j] = run;
    run += x[j];
  }
}

__global__ void seg_fill_kernel(const int* __restrict__ row_of_point, const int* __restrict__ rank,
                                const int* __restrict__ start, int* __restrict__ perm, int64_t P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int row = row_of_point[p];
  if (row >= 0) perm[start[row] + rank[p]] = (int)p;
}

// ---- stable sort of the POINTS by (frame, cell) ---------------------------------------------------
// Same bins as above but with a DETERMINISTIC in-cell order (ascending point id), so the point MLP itself can run on
// the sorted points: every cell's rows are then one contiguous range (the segmented kernels stream instead of
// gathering 512-byte rows), the out-of-range points are one contiguous tail, and the eval path's compacted point
// list is simply the head of the same array.  Order-dependent sums (BatchNorm statistics, weight gradients) see a
// fixed order, so training stays bitwise run-to-run deterministic -- the atomic rank above cannot offer that.
// Counting sort, three passes: (A) 1024-point blocks: in-wave rank by key comparison across lanes, the 16 waves
// take turns on an LDS histogram (block-local stable rank), the histogram goes to T[frame][block][key];
// (B) per (frame, key) exclusive prefix over the blocks + totals, global scan of the totals (-> seg_start);
// (C) scatter: position = seg_start[key] + T[frame][block][key] + block-local rank.
constexpr int SORT_BLK = 1024;
constexpr int SORT_MAX_BINS = 36865;          // H*W + 1 ints of LDS (144 KB of the CU's 160 KB): grids up to 192 x 192

__global__ __launch_bounds__(SORT_BLK) void stable_count_kernel(const float* __restrict__ pts, int* __restrict__ key_out,
                                                                int* __restrict__ lrank_out, int* __restrict__ T, int64_t N,
                                                                int nblk, BevGeom g) {
  extern __shared__ int hist[];
  const int HW = g.H * g.W, nb = HW + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  for (int i = tid; i < nb; i += SORT_BLK) hist[i] = 0;
  const int64_t n = (int64_t)blk * SORT_BLK + tid;
  const bool act = n < N;
  const int64_t p = (int64_t)b * N + (act ? n : 0);
  int key = -1;
  if (act) {
    int c;
    key = bev_cell(kd_ld4(pts + p * 4), g, c) ? c : HW;
  }
  int r = 0, cnt = 0;                            // lower lanes / all lanes of this wave with the same key
#pragma unroll 16
  for (int j = 0; j < 64; ++j) {
    const int kj = __shfl(key, j);
    cnt += kj == key ? 1 : 0;
    r += (kj == key && j < lane) ? 1 : 0;
  }
  __syncthreads();
  int lr = 0;
  for (int w = 0; w < SORT_BLK / 64; ++w) {
    if (wave == w && act) lr = hist[key] + r;
    __syncthreads();
    if (wave == w && act && r == cnt - 1) hist[key] += cnt;      // the last lane of each key group
    __syncthreads();
  }
  if (act) { key_out[p] = key; lrank_out[p] = lr; }
  int* Tb = T + ((int64_t)b * nblk + blk) * nb;
  for (int i = tid; i < nb; i += SORT_BLK) Tb[i] = hist[i];
}

// T[b][blk][key] <- exclusive prefix over blk; totals to counts[b*HW + key] (in range) / inv[b] (key == HW)
__global__ void stable_prefix_kernel(int* T, int* __restrict__ counts, int* __restrict__ inv, int B, int nblk, int HW) {
  const int nb = HW + 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx == 0) counts[(int64_t)B * HW] = 0;
  if (idx >= (int64_t)B * nb) return;
  const int b = (int)(idx / nb), key = (int)(idx % nb);
  int run = 0;
  for (int blk = 0; blk < nblk; ++blk) {
    int* t = T + ((int64_t)b * nblk + blk) * nb + key;
    const int v = *t;
    *t = run;
    run += v;
  }
  if (key < HW) counts[(int64_t)b * HW + key] = run; else inv[b] = run;
}

// inv[b] <- first position of frame b's out-of-range points (they follow all in-range points, frame by frame)
__global__ void stable_inv_base_kernel(const int* __restrict__ seg_start, int* inv, int B, int64_t ncells) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int run = seg_start[ncells];
  for (int b = 0; b < B; ++b) { const int v = inv[b]; inv[b] = run; run += v; }
}

__global__ void stable_fill_kernel(const float* __restrict__ pts, const int* __restrict__ key, const int* __restrict__ lrank,
                                   const int* __restrict__ T, const int* __restrict__ seg_start, const int* __restrict__ inv_base,
                                   float* __restrict__ pts_sorted, int* __restrict__ row_sorted, int* __restrict__ perm,
                                   int64_t P, int64_t N, int nblk, int HW) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int b = (int)(p / N), blk = (int)((p % N) / SORT_BLK), k = key[p];
  const int base = k < HW ? seg_start[(int64_t)b * HW + k] : inv_base[b];
  const int64_t pos = (int64_t)base + T[((int64_t)b * nblk + blk) * (HW + 1) + k] + lrank[p];
  kd_st4(pts_sorted + pos * 4, kd_ld4(pts + p * 4));
  row_sorted[pos] = k < HW ? b * HW + k : -1;
  if (perm) perm[pos] = (int)p;
}

struct SegArgs {
  const float* y; const float* sc; const float* sh; int act;          // deferred [P, C]
  const int* start; const int* perm;                                  // [ncells + 1], [P]
  float* grid; const float* dout; const float* mean; const float* invstd;
  float* G; float* partial; int64_t ncells; int C;
  // Load balance: a grid row with more than `long_len` points (0 = no limit) is left to the chunked kernels below -- one
  // wave per 64 consecutive sorted rows -- so a scene that piles thousands of points into a few cells (a real LiDAR
  // sweep does, around the sensor) costs what a uniform one costs.  rowid [P]: grid row of each sorted point;
  // cnt [ncells, C]: holder counts of the long rows (float, exact); prow0: first slab row of the chunked kernel.
  int long_len; const int* rowid; float* cnt; int prow0;
};
constexpr int SEG_LONG = 256;

template <int VEC> struct SegVec;
template <> struct SegVec<1> { typedef float type; };
template <> struct SegVec<2> { typedef float2 type; };
template <> struct SegVec<4> { typedef float4 type; };

template <int VEC> __device__ __forceinline__ void seg_ld(float (&r)[VEC], const float* p) {
  const typename SegVec<VEC>::type v = *reinterpret_cast<const typename SegVec<VEC>::type*>(p);
  __builtin_memcpy(r, &v, sizeof(v));
}
template <int VEC> __device__ __forceinline__ void seg_st(float* p, const float (&r)[VEC]) {
  typename SegVec<VEC>::type v;
  __builtin_memcpy(&v, r, sizeof(v));
  *reinterpret_cast<typename SegVec<VEC>::type*>(p) = v;
}

// one wave per grid row; lane l owns channels [l*VEC, l*VEC + VEC)  (C == 64*VEC)
template <int VEC, bool PERM>
__global__ __launch_bounds__(256) void seg_max_fwd_kernel(SegArgs a) {
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VEC;
  float sc[VEC], sh[VEC];
  seg_ld<VEC>(sc, a.sc + c0);
  seg_ld<VEC>(sh, a.sh + c0);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < a.ncells; row += nw) {
    const int s = __builtin_amdgcn_readfirstlane(a.start[row]);
    int e = __builtin_amdgcn_readfirstlane(a.start[row + 1]);
    if (a.long_len > 0 && e - s > a.long_len) e = s;     // long row: zero here, maxima from seg_long_fwd_kernel's atomics
    float m[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) m[j] = 0.f;
    int i = s;
    for (; i + 4 <= e; i += 4) {
      float r[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i + u]) : i + u);
        seg_ld<VEC>(r[u], a.y + p * a.C + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = kd_act(kd_affine(r[u][j], sc[j], sh[j]), a.act);
          m[j] = v > m[j] ? v : m[j];      // +0 start, NaN and -0 never win: same bits as the atomic form
        }
    }
    for (; i < e; ++i) {
      float r[VEC];
      const int64_t p = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i]) : i);
      seg_ld<VEC>(r, a.y + p * a.C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = kd_act(kd_affine(r[j], sc[j], sh[j]), a.act);
        m[j] = v > m[j] ? v : m[j];
      }
    }
    seg_st<VEC>(a.grid + row * a.C + c0, m);
  }
}

// TABLE: instead of G[P, C] the kernel leaves share[ncells, C] = dout / (number of holders) (0 where nobody holds) in
// a.G; kd_lidar_l2_dgrad / _wgrad rebuild G on the fly from (y, grid, share), so the [P, C] gradient never exists.
template <int VEC, bool PERM, bool TABLE = false>
__global__ __launch_bounds__(256) void seg_max_bwd_kernel(SegArgs a) {
  __shared__ float red[4][2][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * VEC;
  float sc[VEC], sh[VEC], mu[VEC], inv[VEC], s1[VEC], s2[VEC];
  seg_ld<VEC>(sc, a.sc + c0);
  seg_ld<VEC>(sh, a.sh + c0);
  seg_ld<VEC>(mu, a.mean + c0);
  seg_ld<VEC>(inv, a.invstd + c0);
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.ncells; row += nw) {
    const int s = __builtin_amdgcn_readfirstlane(a.start[row]);
    const int e = __builtin_amdgcn_readfirstlane(a.start[row + 1]);
    if (s == e) continue;
    if (TABLE && a.long_len > 0 && e - s > a.long_len) {   // long row: counted and shared out by the chunked kernels
      float z[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) z[j] = 0.f;
      seg_st<VEC>(a.cnt + row * a.C + c0, z);
      continue;
    }
    float mx[VEC], d[VEC];
    seg_ld<VEC>(mx, a.grid + row * a.C + c0);
    seg_ld<VEC>(d, a.dout + row * a.C + c0);
    int cnt[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) cnt[j] = 0;
    // (round 4: a single-load fast path for rows of at most 16 points -- all rows in registers, both passes from them -- measured
    // 0.4 ms per step SLOWER than the two sweeps below (75.75 vs 75.3 ms, same box): 131 registers cut the occupancy from 8 to 3
    // waves per SIMD, and the second sweep's L2 hits were never the cost.  Not kept.)
    // sweep 1: holders per channel
    int i = s;
    for (; i + 4 <= e; i += 4) {
      float r[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t p = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i + u]) : i + u);
        seg_ld<VEC>(r[u], a.y + p * a.C + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = kd_act(kd_affine(r[u][j], sc[j], sh[j]), a.act);
          cnt[j] += (v > 0.f && v == mx[j]) ? 1 : 0;
        }
    }
    for (; i < e; ++i) {
      float r[VEC];
      const int64_t p = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i]) : i);
      seg_ld<VEC>(r, a.y + p * a.C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = kd_act(kd_affine(r[j], sc[j], sh[j]), a.act);
        cnt[j] += (v > 0.f && v == mx[j]) ? 1 : 0;
      }
    }
    float share[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) share[j] = d[j] / (float)cnt[j];   // only read where cnt >= 1
    if (TABLE) {
      float t[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) t[j] = cnt[j] > 0 ? share[j] : 0.f;
      seg_st<VEC>(a.G + row * a.C + c0, t);
    }
    // sweep 2: the rows come back from L2 (a cell's points were read a few hundred cycles ago)
    i = s;
    for (; i + 4 <= e; i += 4) {
      float r[4][VEC];
      int64_t p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p[u] = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i + u]) : i + u);
        seg_ld<VEC>(r[u], a.y + p[u] * a.C + c0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = kd_act(kd_affine(r[u][j], sc[j], sh[j]), a.act);
          g[j] = (v > 0.f && v == mx[j]) ? share[j] : 0.f;
          s1[j] += g[j];
          s2[j] = fmaf(g[j], (r[u][j] - mu[j]) * inv[j], s2[j]);
        }
        if (!TABLE) seg_st<VEC>(a.G + p[u] * a.C + c0, g);
      }
    }
    for (; i < e; ++i) {
      float r[VEC], g[VEC];
      const int64_t p = (PERM ? __builtin_amdgcn_readfirstlane(a.perm[i]) : i);
      seg_ld<VEC>(r, a.y + p * a.C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = kd_act(kd_affine(r[j], sc[j], sh[j]), a.act);
        g[j] = (v > 0.f && v == mx[j]) ? share[j] : 0.f;
        s1[j] += g[j];
        s2[j] = fmaf(g[j], (r[j] - mu[j]) * inv[j], s2[j]);
      }
      if (!TABLE) seg_st<VEC>(a.G + p * a.C + c0, g);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { red[wave][0][c0 + j] = s1[j]; red[wave][1][c0 + j] = s2[j]; }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
    const int st = i / a.C, c = i % a.C;
    a.partial[((int64_t)blockIdx.x * 2 + st) * a.C + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
  }
}

// ---- chunked kernels for the long rows: wave w owns sorted rows [64w, 64w + 64) -----------------------------------
// A long row (> SEG_LONG >= 64 points) that meets a chunk contains the chunk's first or its last point, so the two
// lookups below reject every chunk of an ordinary scene.  MODE 0: forward maxima (atomicMax on the zeroed grid row);
// MODE 1: holder counts (float atomicAdd of small integers: exact, order-independent); MODE 2: share = dout / count
// (stored once, by the chunk that holds the row's first point) and the BatchNorm-backward sums of the chunk's points.
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void seg_long_kernel(SegArgs a) {
  __shared__ float red[4][2][64 * VEC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = lane * VEC;
  float sc[VEC], sh[VEC], mu[VEC], inv[VEC], s1[VEC], s2[VEC];
  seg_ld<VEC>(sc, a.sc + c0);
  seg_ld<VEC>(sh, a.sh + c0);
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; inv[j] = 0.f; }
  if (MODE == 2) { seg_ld<VEC>(mu, a.mean + c0); seg_ld<VEC>(inv, a.invstd + c0); }
  const int nvalid = __builtin_amdgcn_readfirstlane(a.start[a.ncells]);
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w * 64 < nvalid; w += nw) {
    const int i0 = (int)(w * 64), i1 = i0 + 64 < nvalid ? i0 + 64 : nvalid;
    const int cf = __builtin_amdgcn_readfirstlane(a.rowid[i0]), cl = __builtin_amdgcn_readfirstlane(a.rowid[i1 - 1]);
    const bool lf = a.start[cf + 1] - a.start[cf] > a.long_len, ll = a.start[cl + 1] - a.start[cl] > a.long_len;
    if (!lf && !ll) continue;
    int i = i0;
    while (i < i1) {
      const int cell = __builtin_amdgcn_readfirstlane(a.rowid[i]);
      const int cs = __builtin_amdgcn_readfirstlane(a.start[cell]), ce = __builtin_amdgcn_readfirstlane(a.start[cell + 1]);
      const int rend = ce < i1 ? ce : i1;                 // this chunk's run of `cell`: [i, rend)
      if (ce - cs <= a.long_len) { i = rend; continue; }
      float acc[VEC], mx[VEC], share[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { acc[j] = 0.f; mx[j] = 0.f; share[j] = 0.f; }
      if (MODE >= 1) seg_ld<VEC>(mx, a.grid + (int64_t)cell * a.C + c0);
      if (MODE == 2) {
        float d[VEC], n[VEC];
        seg_ld<VEC>(d, a.dout + (int64_t)cell * a.C + c0);
        seg_ld<VEC>(n, a.cnt + (int64_t)cell * a.C + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) share[j] = n[j] > 0.f ? d[j] / n[j] : 0.f;
        if (i == cs) seg_st<VEC>(a.G + (int64_t)cell * a.C + c0, share);
      }
      int k = i;
      for (; k + 4 <= rend; k += 4) {
        float r[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u) seg_ld<VEC>(r[u], a.y + (int64_t)(k + u) * a.C + c0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float v = kd_act(kd_affine(r[u][j], sc[j], sh[j]), a.act);
            if (MODE == 0) acc[j] = v > acc[j] ? v : acc[j];
            if (MODE == 1) acc[j] += (v > 0.f && v == mx[j]) ? 1.f : 0.f;
            if (MODE == 2) {
              const float g = (v > 0.f && v == mx[j]) ? share[j] : 0.f;
              s1[j] += g;
              s2[j] = fmaf(g, (r[u][j] - mu[j]) * inv[j], s2[j]);
            }
          }
      }
      for (; k < rend; ++k) {
        float r[VEC];
        seg_ld<VEC>(r, a.y + (int64_t)k * a.C + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float v = kd_act(kd_affine(r[j], sc[j], sh[j]), a.act);
          if (MODE == 0) acc[j] = v > acc[j] ? v : acc[j];
          if (MODE == 1) acc[j] += (v > 0.f && v == mx[j]) ? 1.f : 0.f;
          if (MODE == 2) {
            const float g = (v > 0.f && v == mx[j]) ? share[j] : 0.f;
            s1[j] += g;
            s2[j] = fmaf(g, (r[j] - mu[j]) * inv[j], s2[j]);
          }
        }
      }
      if (MODE == 0) {
        unsigned* dst = reinterpret_cast<unsigned*>(a.grid + (int64_t)cell * a.C + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (acc[j] > 0.f) atomicMax(dst + j, __float_as_uint(acc[j]));
      }
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (acc[j] > 0.f) atomicAdd(a.cnt + (int64_t)cell * a.C + c0 + j, acc[j]);
      }
      i = rend;
    }
  }
  if (MODE == 2) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[wave][0][c0 + j] = s1[j]; red[wave][1][c0 + j] = s2[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * a.C; i += 256) {
      const int st = i / a.C, c = i % a.C;
      a.partial[((int64_t)(a.prow0 + blockIdx.x) * 2 + st) * a.C + c] = red[0][st][c] + red[1][st][c] + red[2][st][c] + red[3][st][c];
    }
  }
}

inline int seg_long_grid(int64_t P) { const int64_t b = (P + 255) / 256; return (int)(b < 1024 ? b : 1024); }

// rows of G that belong to out-of-range points get no gradient: they are not in any segment
// (a 256-thread block sweeps 16 * (1024 / C) consecutive points: one wave per handful of stores would be launch-bound)
__global__ __launch_bounds__(256) void seg_zero_invalid_kernel(const int* __restrict__ row_of_point, float* __restrict__ G,
                                                               int64_t P, int c4) {
  const int per = 256 / c4;                      // points per sweep of the block
  const int q = threadIdx.x % c4, sub = threadIdx.x / c4;
  const int64_t base = (int64_t)blockIdx.x * per * 16;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int64_t p = base + (int64_t)k * per + sub;
    if (p < P && row_of_point[p] < 0) kd_st4(G + (p * c4 + q) * 4, kd_zero4());
  }
}

// the in-range points in cell order, with their grid rows: the eval path's point list (see kd_lidar_compact)
__global__ void seg_gather_kernel(const float* __restrict__ pts, const int* __restrict__ perm, const int* __restrict__ row_of_point,
                                  const int* __restrict__ nvalid, float* __restrict__ out_pts, int* __restrict__ out_row, int64_t P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P || i >= *nvalid) return;
  const int p = perm[i];
  kd_st4(out_pts + i * 4, kd_ld4(pts + (int64_t)p * 4));
  out_row[i] = row_of_point[p];
}

inline int seg_grid(int64_t ncells) { const int64_t b = (ncells + 3) / 4; return (int)(b < 4096 ? b : 4096); }

}  // namespace

extern "C" {

size_t kd_lidar_cell_sort_ws_bytes(int B, int64_t N, int H, int W) {
  const int64_t n = (int64_t)B * H * W + 1;
  return ((size_t)B * N + (size_t)((n + SCAN_CHUNK - 1) / SCAN_CHUNK)) * sizeof(int);
}

// Bins the B*N points by BEV grid row.  row_of_point[p] = b*H*W + cell, or -1 for out-of-range points;
// seg_start[r] .. seg_start[r+1] delimit row r's point ids inside perm (seg_start[B*H*W] = number of in-range points).
int kd_lidar_cell_sort(const float* pts, int B, int64_t N, int H, int W, float x0, float x1, float y0, float y1,
                       int* row_of_point, int* seg_start, int* perm, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(pts && row_of_point && seg_start && perm && ws && B > 0 && N > 0 && H > 0 && W > 0, KD_ERR_ARG, "kd_lidar_cell_sort: bad args");
  const int64_t P = (int64_t)B * N, n = (int64_t)B * H * W + 1;
  KD_REQUIRE(P < (int64_t)INT32_MAX && n < (int64_t)INT32_MAX, KD_ERR_SHAPE, "kd_lidar_cell_sort: more than 2^31 points or cells");
  KD_REQUIRE(ws_bytes >= kd_lidar_cell_sort_ws_bytes(B, N, H, W), KD_ERR_WORKSPACE, "kd_lidar_cell_sort: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  int* rank = (int*)ws;
  int* bsum = rank + P;
  const int nb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  hipError_t e = hipMemsetAsync(seg_start, 0, (size_t)n * sizeof(int), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_cell_sort: memset failed: %s", hipGetErrorString(e));
  const BevGeom g{x0, x1 - x0, y0, y1 - y0, H, W};
  hipLaunchKernelGGL(seg_count_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, pts, row_of_point, rank, seg_start, P, N, g);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nb), dim3(256), 0, st, seg_start, bsum, n);
  hipLaunchKernelGGL(scan_bsums_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(256), 0, st, seg_start, bsum, n);
  hipLaunchKernelGGL(seg_fill_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, row_of_point, rank, seg_start, perm, P);
  return kd_check_launch("kd_lidar_cell_sort");
}

size_t kd_lidar_sort_points_ws_bytes(int B, int64_t N, int H, int W) {
  const int64_t P = (int64_t)B * N, nblk = (N + SORT_BLK - 1) / SORT_BLK, nb = (int64_t)H * W + 1, n = (int64_t)B * H * W + 1;
  return (size_t)(2 * P + (int64_t)B * nblk * nb + B + (n + SCAN_CHUNK - 1) / SCAN_CHUNK) * sizeof(int);
}

// Stable sort of the points by (frame, cell); out-of-range points go to the end (frame by frame, original order).
//   pts_sorted[B*N, 4]; row_sorted[B*N] = grid row of each sorted point or -1; seg_start[B*H*W + 1] as in
//   kd_lidar_cell_sort (seg_start[B*H*W] = number of in-range points); perm (optional, may be NULL): sorted -> original id.
// Returns KD_ERR_SHAPE when H*W + 1 exceeds the LDS histogram (callers fall back to kd_lidar_cell_sort).
int kd_lidar_sort_points(const float* pts, int B, int64_t N, int H, int W, float x0, float x1, float y0, float y1,
                         float* pts_sorted, int* row_sorted, int* seg_start, int* perm, void* ws, size_t ws_bytes,
                         void* stream) {
  KD_REQUIRE(pts && pts_sorted && row_sorted && seg_start && ws && B > 0 && N > 0 && H > 0 && W > 0, KD_ERR_ARG, "kd_lidar_sort_points: bad args");
  const int64_t P = (int64_t)B * N, ncells = (int64_t)B * H * W, n = ncells + 1;
  const int HW = H * W, nb = HW + 1;
  KD_REQUIRE(nb <= SORT_MAX_BINS, KD_ERR_SHAPE, "kd_lidar_sort_points: H*W + 1 = %d bins exceed %d", nb, SORT_MAX_BINS);
  KD_REQUIRE(P < (int64_t)INT32_MAX && n < (int64_t)INT32_MAX, KD_ERR_SHAPE, "kd_lidar_sort_points: more than 2^31 points or cells");
  KD_REQUIRE(ws_bytes >= kd_lidar_sort_points_ws_bytes(B, N, H, W), KD_ERR_WORKSPACE, "kd_lidar_sort_points: workspace too small");
  const int nblk = (int)((N + SORT_BLK - 1) / SORT_BLK);
  KD_REQUIRE((int64_t)B * nblk < (int64_t)INT32_MAX, KD_ERR_SHAPE, "kd_lidar_sort_points: too many blocks");
  hipStream_t st = (hipStream_t)stream;
  int* key = (int*)ws;
  int* lrank = key + P;
  int* T = lrank + P;
  int* inv = T + (int64_t)B * nblk * nb;
  int* bsum = inv + B;
  const int nsb = (int)((n + SCAN_CHUNK - 1) / SCAN_CHUNK);
  const BevGeom g{x0, x1 - x0, y0, y1 - y0, H, W};
  if ((size_t)nb * sizeof(int) > 48 * 1024) {      // above the default dynamic-LDS limit: opt in (idempotent, host-side only)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(stable_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       SORT_MAX_BINS * (int)sizeof(int));
    KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_sort_points: cannot raise the LDS limit: %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(stable_count_kernel, dim3((unsigned)(B * nblk)), dim3(SORT_BLK), (size_t)nb * sizeof(int), st, pts, key, lrank, T,
                     N, nblk, g);
  const int64_t nbk = (int64_t)B * nb;
  hipLaunchKernelGGL(stable_prefix_kernel, dim3((unsigned)((nbk + 255) / 256)), dim3(256), 0, st, T, seg_start, inv, B, nblk, HW);
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nsb), dim3(256), 0, st, seg_start, bsum, n);
  hipLaunchKernelGGL(scan_bsums_kernel, dim3(1), dim3(1024), 0, st, bsum, nsb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nsb), dim3(256), 0, st, seg_start, bsum, n);
  hipLaunchKernelGGL(stable_inv_base_kernel, dim3(1), dim3(64), 0, st, seg_start, inv, B, ncells);
  hipLaunchKernelGGL(stable_fill_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, pts, key, lrank, T, seg_start, inv,
                     pts_sorted, row_sorted, perm, P, N, nblk, HW);
  return kd_check_launch("kd_lidar_sort_points");
}

// out_pts[i] = pts[perm[i]], out_row[i] = row_of_point[perm[i]] for i < *nvalid_dev (= seg_start[B*H*W]): the same
// contract as kd_lidar_compact's outputs, but ordered by grid row, which lets kd_lidar_l2_fwd_scatter merge
// neighbouring rows before it touches the grid.
int kd_lidar_gather_sorted(const float* pts, const int* perm, const int* row_of_point, const int* nvalid_dev, float* out_pts,
                           int* out_row, int64_t P, void* stream) {
  KD_REQUIRE(pts && perm && row_of_point && nvalid_dev && out_pts && out_row && P > 0, KD_ERR_ARG, "kd_lidar_gather_sorted: bad args");
  hipLaunchKernelGGL(seg_gather_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, perm, row_of_point,
                     nvalid_dev, out_pts, out_row, P);
  return kd_check_launch("kd_lidar_gather_sorted");
}

// grid[ncells, C] = per-row max of act(y*sc+sh) over the row's points (0 for empty rows); every row is written.
// row_sorted (optional, only without perm): grid row of every sorted point; with it, rows holding more than 256 points
// are processed 64 points per wave (load balance on concentrated scenes); P = number of points (bounds the launch).
int kd_lidar_seg_max_fwd(const float* y, const float* sc, const float* sh, int act, const int* seg_start, const int* perm,
                         const int* row_sorted, float* grid, int64_t P, int64_t ncells, int C, void* stream) {
  KD_REQUIRE(y && sc && sh && seg_start && grid && ncells > 0, KD_ERR_ARG, "kd_lidar_seg_max_fwd: bad args");
  KD_REQUIRE(C == 64 || C == 128 || C == 256, KD_ERR_SHAPE, "kd_lidar_seg_max_fwd: C must be 64, 128 or 256 (got %d)", C);
  KD_REQUIRE(act == KD_ACT_RELU || act == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_seg_max_fwd: needs a non-negative activation");
  const bool chunked = row_sorted != nullptr && perm == nullptr && P > 0;
  SegArgs a{y, sc, sh, act, seg_start, perm, grid, nullptr, nullptr, nullptr, nullptr, nullptr, ncells, C,
            chunked ? SEG_LONG : 0, row_sorted, nullptr, 0};
  const dim3 gr(seg_grid(ncells)), bl(256);
  hipStream_t st = (hipStream_t)stream;
#define KD_SEG_LAUNCH(kern)                                                                          \
  do {                                                                                              \
    if (C == 64) { if (perm) hipLaunchKernelGGL((kern<1, true>), gr, bl, 0, st, a); else hipLaunchKernelGGL((kern<1, false>), gr, bl, 0, st, a); } \
    else if (C == 128) { if (perm) hipLaunchKernelGGL((kern<2, true>), gr, bl, 0, st, a); else hipLaunchKernelGGL((kern<2, false>), gr, bl, 0, st, a); } \
    else { if (perm) hipLaunchKernelGGL((kern<4, true>), gr, bl, 0, st, a); else hipLaunchKernelGGL((kern<4, false>), gr, bl, 0, st, a); } \
  } while (0)
  KD_SEG_LAUNCH(seg_max_fwd_kernel);
  if (chunked) {
    const dim3 gl(seg_long_grid(P));
    if (C == 64) hipLaunchKernelGGL((seg_long_kernel<1, 0>), gl, bl, 0, st, a);
    else if (C == 128) hipLaunchKernelGGL((seg_long_kernel<2, 0>), gl, bl, 0, st, a);
    else hipLaunchKernelGGL((seg_long_kernel<4, 0>), gl, bl, 0, st, a);
  }
  return kd_check_launch("kd_lidar_seg_max_fwd");
}

int64_t kd_lidar_seg_stat_rows(int64_t ncells) { return seg_grid(ncells); }

// G[P, C] and partial[kd_lidar_seg_stat_rows][2][C]: same contract as kd_lidar_scatter_max_bwd.
int kd_lidar_seg_max_bwd(const float* y, const float* sc, const float* sh, int act, const float* grid, const float* dout,
                         const float* mean, const float* invstd, const int* seg_start, const int* perm,
                         const int* row_of_point, float* G, float* partial, int64_t P, int64_t ncells, int C, void* stream) {
  KD_REQUIRE(y && sc && sh && grid && dout && mean && invstd && seg_start && row_of_point && G && partial && P > 0 && ncells > 0,
             KD_ERR_ARG, "kd_lidar_seg_max_bwd: bad args");
  KD_REQUIRE(C == 64 || C == 128 || C == 256, KD_ERR_SHAPE, "kd_lidar_seg_max_bwd: C must be 64, 128 or 256 (got %d)", C);
  SegArgs a{y, sc, sh, act, seg_start, perm, const_cast<float*>(grid), dout, mean, invstd, G, partial, ncells, C, 0, nullptr, nullptr, 0};
  const dim3 gr(seg_grid(ncells)), bl(256);
  hipStream_t st = (hipStream_t)stream;
  const int c4 = C / 4;
  const int64_t per_block = (int64_t)(256 / c4) * 16;
  hipLaunchKernelGGL(seg_zero_invalid_kernel, dim3((unsigned)((P + per_block - 1) / per_block)), dim3(256), 0, st, row_of_point, G, P, c4);
  KD_SEG_LAUNCH(seg_max_bwd_kernel);
#undef KD_SEG_LAUNCH
  return kd_check_launch("kd_lidar_seg_max_bwd");
}

// Table form of kd_lidar_seg_max_bwd for rows sorted by cell (kd_lidar_sort_points; row r of `y` belongs to grid row
// g iff seg_start[g] <= r < seg_start[g+1]): share[ncells, C] = dout / holders (rows of empty cells are not written and
// never read), partial as in kd_lidar_seg_max_bwd.  G itself is rebuilt by kd_lidar_l2_dgrad / kd_lidar_l2_wgrad.
// partial has kd_lidar_seg_share_stat_rows(ncells, P) rows; cnt_ws [ncells, C] floats is scratch (holder counts of the
// rows with more than 256 points, which the chunked kernels process 64 points per wave).
int64_t kd_lidar_seg_share_stat_rows(int64_t ncells, int64_t P) { return seg_grid(ncells) + seg_long_grid(P); }

int kd_lidar_seg_share_bwd(const float* y, const float* sc, const float* sh, int act, const float* grid, const float* dout,
                           const float* mean, const float* invstd, const int* seg_start, const int* row_sorted, float* share,
                           float* cnt_ws, float* partial, int64_t P, int64_t ncells, int C, void* stream) {
  KD_REQUIRE(y && sc && sh && grid && dout && mean && invstd && seg_start && row_sorted && share && cnt_ws && partial && ncells > 0 && P > 0,
             KD_ERR_ARG, "kd_lidar_seg_share_bwd: bad args");
  KD_REQUIRE(C == 64 || C == 128 || C == 256, KD_ERR_SHAPE, "kd_lidar_seg_share_bwd: C must be 64, 128 or 256 (got %d)", C);
  SegArgs a{y, sc, sh, act, seg_start, nullptr, const_cast<float*>(grid), dout, mean, invstd, share, partial, ncells, C,
            SEG_LONG, row_sorted, cnt_ws, seg_grid(ncells)};
  const dim3 gr(seg_grid(ncells)), gl(seg_long_grid(P)), bl(256);
  hipStream_t st = (hipStream_t)stream;
  if (C == 64) {
    hipLaunchKernelGGL((seg_max_bwd_kernel<1, false, true>), gr, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<1, 1>), gl, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<1, 2>), gl, bl, 0, st, a);
  } else if (C == 128) {
    hipLaunchKernelGGL((seg_max_bwd_kernel<2, false, true>), gr, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<2, 1>), gl, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<2, 2>), gl, bl, 0, st, a);
  } else {
    hipLaunchKernelGGL((seg_max_bwd_kernel<4, false, true>), gr, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<4, 1>), gl, bl, 0, st, a);
    hipLaunchKernelGGL((seg_long_kernel<4, 2>), gl, bl, 0, st, a);
  }
  return kd_check_launch("kd_lidar_seg_share_bwd");
}

}  // extern "C"
