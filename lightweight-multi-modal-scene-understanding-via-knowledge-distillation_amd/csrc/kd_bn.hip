// kd_bn.hip -- BatchNorm as per-channel coefficient kernels (nn.BatchNorm2d / BatchNorm1d semantics,
// eps 1e-5, momentum 0.1: camera_encoder.py:25,33,40,65  lidar_encoder.py:27,30,33  fusion_module.py:13,27,30).
//
// The conv kernels leave per-block partial sums in a slab [rows][2][C]; the finalize kernels below
// reduce them in fp64 in a fixed order (deterministic) and emit the coefficient vectors that the
// NEXT kernel folds into its operand loads:
//   forward : scale = gamma*invstd, shift = beta - mean*scale         (value = act(raw*scale+shift))
//   backward: dy_raw = al*G + be*X + ga                                (G = dL/dact * act'(z))
#include "kd_common.h"

namespace {

// grid.x = ceil(C/64); block (64, 4): 4 row lanes per channel
// Stage 1 of the slab reduction (only for tall slabs): every group of `group` consecutive rows is
// summed IN PLACE into the group's first row, so the finalize kernels below walk rows/group rows
// instead of serialising thousands of dependent loads in a handful of workgroups.
// grid (ceil(C/64), ceil(rows/group)), block (64, 4).  The slab is scratch: clobbering it is fine.
__global__ void slab_prereduce_kernel(float* partial, int rows, int C, int pstride, int group) {
  __shared__ double sm[4][2][64];
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int r0 = blockIdx.y * group;
  const int r1 = r0 + group < rows ? r0 + group : rows;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (int r = r0 + threadIdx.y; r < r1; r += 4) {
      a += (double)partial[((int64_t)r * 2 + 0) * pstride + c];
      b += (double)partial[((int64_t)r * 2 + 1) * pstride + c];
    }
  sm[threadIdx.y][0][threadIdx.x] = a;
  sm[threadIdx.y][1][threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.y == 0 && c < C) {
    partial[((int64_t)r0 * 2 + 0) * pstride + c] = (float)(sm[0][0][threadIdx.x] + sm[1][0][threadIdx.x] + sm[2][0][threadIdx.x] + sm[3][0][threadIdx.x]);
    partial[((int64_t)r0 * 2 + 1) * pstride + c] = (float)(sm[0][1][threadIdx.x] + sm[1][1][threadIdx.x] + sm[2][1][threadIdx.x] + sm[3][1][threadIdx.x]);
  }
}

// returns the row step the finalize kernel must use (1 = slab untouched).  The finalize kernels walk the slab with FRL = 64 row
// lanes per channel and FCW = 16 channels per workgroup (a 2048-row slab -- what the streaming GEMMs and the depthwise kernels
// leave -- is 32 rows per lane, 16 in flight), so only slabs above 8192 rows need the first stage: one launch per BatchNorm
// instead of two, 58 launches fewer per KD step.  Measured per BatchNorm (round 3, rocprofv3 averages over the step's 58
// finalizes): the two launches of rounds 1-2: 6.4 + 6.5 us; one launch, 64 channels x 16 row lanes: 16.4 us; 16 x 64 (this):
// 14.7 us; 8 x 128 with all 16 rows of a lane in flight: 17.6-21.2 us; a single launch whose last workgroup (ticket counter)
// finalizes: 20 us (its device-scope release / acquire fences write back and invalidate L2).  I.e. the single launch costs
// ~0.1 ms per step more GPU time than the pair and saves 58 launches of host work; nothing here is bandwidth.
constexpr int FCW = 16, FRL = 64, FUN = 8;
static int slab_prereduce(float* partial, int& rows, int C, int pstride, hipStream_t st) {
  if (rows <= 8192) return 1;
  int group = 1;
  while (group * group < rows) ++group;                      // ~sqrt(rows)
  const int ngroups = (rows + group - 1) / group;
  hipLaunchKernelGGL(slab_prereduce_kernel, dim3((C + 63) / 64, ngroups), dim3(64, 4), 0, st, partial, rows, C, pstride, group);
  rows = ngroups;
  return group;
}

// block (FCW, FRL): lane y of channel c sums rows y, y+FRL, ... (FUN rows in flight); lane 0 then adds the FRL partial sums in order
__device__ __forceinline__ void reduce_slab2(const float* partial, int rows, int rstep, int C, int pstride, int c,
                                             double& s1, double& s2, double (*sm)[2][FCW]) {
  double a = 0.0, b = 0.0;
  if (c < C) {
    int r = threadIdx.y;
    for (; r + (FUN - 1) * FRL < rows; r += FUN * FRL) {
      float v[FUN][2];
#pragma unroll
      for (int u = 0; u < FUN; ++u) {
        v[u][0] = partial[((int64_t)(r + u * FRL) * rstep * 2 + 0) * pstride + c];
        v[u][1] = partial[((int64_t)(r + u * FRL) * rstep * 2 + 1) * pstride + c];
      }
#pragma unroll
      for (int u = 0; u < FUN; ++u) { a += (double)v[u][0]; b += (double)v[u][1]; }
    }
    float t[FUN][2];                                  // the remainder, also fetched together
#pragma unroll
    for (int u = 0; u < FUN; ++u) {
      const int rr = r + u * FRL;
      const bool ok = rr < rows;
      const int64_t q = (int64_t)(ok ? rr : 0) * rstep * 2 * pstride + c;
      t[u][0] = ok ? partial[q] : 0.f;
      t[u][1] = ok ? partial[q + pstride] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < FUN; ++u) { a += (double)t[u][0]; b += (double)t[u][1]; }
  }
  sm[threadIdx.y][0][threadIdx.x] = a;
  sm[threadIdx.y][1][threadIdx.x] = b;
  __syncthreads();
  s1 = s2 = 0.0;
  if (threadIdx.y == 0)
#pragma unroll 8
    for (int y = 0; y < FRL; ++y) { s1 += sm[y][0][threadIdx.x]; s2 += sm[y][1][threadIdx.x]; }
}

// Tall slabs (round 4).  The tiled GEMM leaves one statistics row per 128 matrix rows (8192 rows for a 1 M-row layer), the depthwise /
// resize / reduce kernels one per workgroup (2048): with 16 channels per workgroup the form above puts a 64-channel layer's 4 MB slab
// on FOUR compute units, each lane walking 128 rows eight at a time -- 33-58 us per BatchNorm where the short slabs of the streaming
// GEMMs take 5-7 (rocprofv3 trace of the step, r04).  Here a workgroup owns FOUR channels (one 16-byte load per row and statistic) and
// 256 row lanes: a 64-channel layer runs on 16 compute units and a lane walks 32 rows.  Lane y sums rows y, y + 256, ... in order
// (eight rows in flight), in double; 64 threads then add 32 lanes each, in lane order, and 8 threads the eight group sums: fixed order.
// Needs C, pstride multiples of 4 and a 16-byte-aligned slab; returns true in the four threads that hold a channel's sums.
constexpr int TL = 256, TUN = 8;
__device__ __forceinline__ bool reduce_slab2_tall(const float* partial, int rows, int rstep, int pstride, int c0, double& s1, double& s2,
                                                  double (*sm)[8], double (*sg)[8]) {
  const int y = threadIdx.x;
  const int64_t rs = (int64_t)rstep * 2 * pstride;
  const float* base = partial + c0;
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  int r = y;
  for (; r + (TUN - 1) * TL < rows; r += TUN * TL) {
    float4 v[TUN], w[TUN];
#pragma unroll
    for (int u = 0; u < TUN; ++u) {
      const float* q = base + (int64_t)(r + u * TL) * rs;
      v[u] = kd_ld4(q);
      w[u] = kd_ld4(q + pstride);
    }
#pragma unroll
    for (int u = 0; u < TUN; ++u) {
      a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w;
      b[0] += (double)w[u].x; b[1] += (double)w[u].y; b[2] += (double)w[u].z; b[3] += (double)w[u].w;
    }
  }
  {
    float4 v[TUN], w[TUN];                            // the remainder, also fetched together
#pragma unroll
    for (int u = 0; u < TUN; ++u) {
      const int rr = r + u * TL;
      const bool ok = rr < rows;
      const float* q = base + (int64_t)(ok ? rr : 0) * rs;
      v[u] = kd_ld4(q);
      w[u] = kd_ld4(q + pstride);
      if (!ok) { v[u] = kd_zero4(); w[u] = kd_zero4(); }
    }
#pragma unroll
    for (int u = 0; u < TUN; ++u) {
      a[0] += (double)v[u].x; a[1] += (double)v[u].y; a[2] += (double)v[u].z; a[3] += (double)v[u].w;
      b[0] += (double)w[u].x; b[1] += (double)w[u].y; b[2] += (double)w[u].z; b[3] += (double)w[u].w;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { sm[y][i] = a[i]; sm[y][4 + i] = b[i]; }
  __syncthreads();
  if (y < 64) {                                       // thread (group = y >> 3, slot = y & 7): lanes 32 group .. 32 group + 31 of one slot
    const int gq = y >> 3, slot = y & 7;
    double t = 0.0;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) t += sm[32 * gq + k][slot];
    sg[gq][slot] = t;
  }
  __syncthreads();
  s1 = s2 = 0.0;
  if (y >= 4) return false;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s1 += sg[k][y]; s2 += sg[k][4 + y]; }
  return true;
}

struct TrainTail {
  double count; const float* gamma; const float* beta; float eps, momentum; float* running_mean; float* running_var; int64_t* nbt;
  float* mean; float* invstd; float* scale; float* shift;
};
__device__ __forceinline__ void finalize_train_tail(const TrainTail& t, int c, double s1, double s2) {
  const double count = t.count;
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float inv = (float)(1.0 / sqrt(var + (double)t.eps));
  const float g = t.gamma ? t.gamma[c] : 1.f, b = t.beta ? t.beta[c] : 0.f;
  const float muf = (float)mu;
  t.mean[c] = muf;
  t.invstd[c] = inv;
  const float sc = g * inv;
  t.scale[c] = sc;
  t.shift[c] = fmaf(-muf, sc, b);
  if (t.running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    t.running_mean[c] = (float)((1.0 - t.momentum) * (double)t.running_mean[c] + t.momentum * mu);
    t.running_var[c] = (float)((1.0 - t.momentum) * (double)t.running_var[c] + t.momentum * unbiased);
  }
  if (t.nbt && c == 0) *t.nbt += 1;
}
__global__ __launch_bounds__(FCW * FRL) void bn_finalize_train_kernel(const float* partial, int rows, int rstep, int C, int pstride, TrainTail t) {
  __shared__ double sm[FRL][2][FCW];
  const int c = blockIdx.x * FCW + threadIdx.x;
  double s1, s2;
  reduce_slab2(partial, rows, rstep, C, pstride, c, s1, s2, sm);
  if (threadIdx.y != 0 || c >= C) return;
  finalize_train_tail(t, c, s1, s2);
}
__global__ __launch_bounds__(TL) void bn_finalize_train_tall_kernel(const float* partial, int rows, int rstep, int pstride, TrainTail t) {
  __shared__ double sm[TL][8];
  __shared__ double sg[8][8];
  double s1, s2;
  if (reduce_slab2_tall(partial, rows, rstep, pstride, blockIdx.x * 4, s1, s2, sm, sg)) finalize_train_tail(t, blockIdx.x * 4 + threadIdx.x, s1, s2);
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, int C, float* mean, float* invstd, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv = 1.f / sqrtf(rv[c] + eps);
  const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
  const float sc = g * inv;
  if (mean) mean[c] = rm[c];
  if (invstd) invstd[c] = inv;
  scale[c] = sc;
  shift[c] = fmaf(-rm[c], sc, b);
}

struct BwdTail {
  double count; const float* gamma; const float* mean; const float* invstd; int training; float* dgamma; float* dbeta; float* al; float* be;
  float* ga; float* dbias;
};
__device__ __forceinline__ void bwd_finalize_tail(const BwdTail& t, int c, double s1, double s2) {
  if (t.dbeta) t.dbeta[c] = (float)s1;
  if (t.dgamma) t.dgamma[c] = (float)s2;
  const double g = t.gamma ? (double)t.gamma[c] : 1.0;
  const double inv = (double)t.invstd[c], mu = (double)t.mean[c];
  const double a = g * inv;
  if (t.training) {
    const double c1 = s1 / t.count, c2 = s2 / t.count;
    t.al[c] = (float)a;
    t.be[c] = (float)(-a * c2 * inv);
    t.ga[c] = (float)(a * (c2 * inv * mu - c1));
    if (t.dbias) t.dbias[c] = 0.f;      // sum_m dy_raw == 0 identically under batch statistics
  } else {
    t.al[c] = (float)a;
    t.be[c] = 0.f;
    t.ga[c] = 0.f;
    if (t.dbias) t.dbias[c] = (float)(a * s1);
  }
}
__global__ __launch_bounds__(FCW * FRL) void bn_bwd_finalize_kernel(const float* partial, int rows, int rstep, int C, int pstride, BwdTail t) {
  __shared__ double sm[FRL][2][FCW];
  const int c = blockIdx.x * FCW + threadIdx.x;
  double s1, s2;
  reduce_slab2(partial, rows, rstep, C, pstride, c, s1, s2, sm);
  if (threadIdx.y != 0 || c >= C) return;
  bwd_finalize_tail(t, c, s1, s2);
}
__global__ __launch_bounds__(TL) void bn_bwd_finalize_tall_kernel(const float* partial, int rows, int rstep, int pstride, BwdTail t) {
  __shared__ double sm[TL][8];
  __shared__ double sg[8][8];
  double s1, s2;
  if (reduce_slab2_tall(partial, rows, rstep, pstride, blockIdx.x * 4, s1, s2, sm, sg)) bwd_finalize_tail(t, blockIdx.x * 4 + threadIdx.x, s1, s2);
}

struct ApplyArgs {
  const float* x; int64_t ldx; const float* sc; const float* sh; int act;
  const float* res; int64_t ldr; float* out; int64_t ldo;
  int64_t M; int C; int groups, slots; int nt;
  const float* rsc; const float* rsh; int ract;       // the residual is itself a deferred tensor: res = ract(res_raw * rsc + rsh)
};
__global__ __launch_bounds__(256) void bn_apply_kernel(ApplyArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), rsc = sc, rsh = sh;
  if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
  if (a.rsc) { rsc = kd_ld4(a.rsc + c0); rsh = kd_ld4(a.rsh + c0); }
  for (int64_t m = (int64_t)blockIdx.x * a.slots + slot; m < a.M; m += (int64_t)gridDim.x * a.slots) {
    float4 v = kd_ld4(a.x + m * a.ldx + c0);
    v = kd_affine_act4(v, sc, sh, a.act);
    if (a.res) {
      float4 r = kd_ld4(a.res + m * a.ldr + c0);
      if (a.rsc) r = kd_affine_act4(r, rsc, rsh, a.ract);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (a.nt) kd_st4_nt(a.out + m * a.ldo + c0, v); else kd_st4(a.out + m * a.ldo + c0, v);
  }
}

struct BwdReduceArgs {
  const float* D; int64_t ldd; const float* X; int64_t ldx; const float* sc; const float* sh; int act;
  const float* mean; const float* invstd; float* partial; int64_t M; int C; int groups, slots;
};
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BwdReduceArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    const float4 mu = kd_ld4(a.mean + c0), inv = kd_ld4(a.invstd + c0);
    for (int64_t m = (int64_t)blockIdx.x * a.slots + slot; m < a.M; m += (int64_t)gridDim.x * a.slots) {
      const float4 d = kd_ld4(a.D + m * a.ldd + c0);
      const float4 x = kd_ld4(a.X + m * a.ldx + c0);
      const float gx = d.x * kd_act_mask(kd_affine(x.x, sc.x, sh.x), a.act);
      const float gy = d.y * kd_act_mask(kd_affine(x.y, sc.y, sh.y), a.act);
      const float gz = d.z * kd_act_mask(kd_affine(x.z, sc.z, sh.z), a.act);
      const float gw = d.w * kd_act_mask(kd_affine(x.w, sc.w, sh.w), a.act);
      s1.x += gx; s1.y += gy; s1.z += gz; s1.w += gw;
      s2.x = fmaf(gx, (x.x - mu.x) * inv.x, s2.x);
      s2.y = fmaf(gy, (x.y - mu.y) * inv.y, s2.y);
      s2.z = fmaf(gz, (x.z - mu.z) * inv.z, s2.z);
      s2.w = fmaf(gw, (x.w - mu.w) * inv.w, s2.w);
    }
  }
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

// the four-channel form where the slab is tall enough for it to pay (KD_BN_TALL=0: never -- the A/B switch of tools/)
bool slab_is_tall(const float* partial, int rows, int C, int pstride) {
  static const int min_rows = [] { const char* e = getenv("KD_BN_TALL"); return e ? atoi(e) : 1024; }();
  return min_rows > 0 && rows >= min_rows && C % 4 == 0 && pstride % 4 == 0 && kd_aligned16(partial);
}

}  // namespace

extern "C" {

// Batch statistics -> forward coefficients; updates running stats / num_batches_tracked in place
// (pass null to skip).  partial: [rows][2][C] holding (sum, sum of squares) over `count` samples.
int kd_bn_finalize_train(float* partial, int rows, int C, int pstride, int64_t count, const float* gamma, const float* beta,
                         float eps, float momentum, float* running_mean, float* running_var, int64_t* nbt,
                         float* mean, float* invstd, float* scale, float* shift, void* stream) {
  KD_REQUIRE(partial && rows > 0 && C > 0 && pstride >= C && count > 0 && mean && invstd && scale && shift, KD_ERR_ARG, "kd_bn_finalize_train: bad args");
  const int rstep = slab_prereduce(partial, rows, C, pstride, (hipStream_t)stream);
  const TrainTail t{(double)count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, invstd, scale, shift};
  if (slab_is_tall(partial, rows, C, pstride))
    hipLaunchKernelGGL(bn_finalize_train_tall_kernel, dim3(C / 4), dim3(TL), 0, (hipStream_t)stream, partial, rows, rstep, pstride, t);
  else
    hipLaunchKernelGGL(bn_finalize_train_kernel, dim3((C + FCW - 1) / FCW), dim3(FCW, FRL), 0, (hipStream_t)stream, partial, rows, rstep, C,
                       pstride, t);
  return kd_check_launch("kd_bn_finalize_train");
}

int kd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* mean, float* invstd, float* scale, float* shift, void* stream) {
  KD_REQUIRE(running_mean && running_var && scale && shift && C > 0, KD_ERR_ARG, "kd_bn_eval_coeffs: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, C, mean, invstd, scale, shift);
  return kd_check_launch("kd_bn_eval_coeffs");
}

int64_t kd_rowwise_stat_rows(int64_t M, int C) { return kd_cg_layout(M, C).grid; }

// out = act(x*sc+sh) (+ res).  sc == null: identity affine.
int kd_bn_act_apply(const float* x, int64_t ldx, const float* sc, const float* sh, int act, const float* res,
                    int64_t ldr, float* out, int64_t ldo, int64_t M, int C, void* stream) {
  KD_REQUIRE(x && out && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bn_act_apply: bad args (C=%d)", C);
  KD_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && (!res || ldr % 4 == 0), KD_ERR_SHAPE, "kd_bn_act_apply: ld must be a multiple of 4");
  const KdCgLayout l = kd_cg_layout(M, C);
  ApplyArgs a{x, ldx, sc, sh, act, res, ldr, out, ldo, M, C, l.groups, l.slots, kd_nt_store((size_t)M * C * sizeof(float)), nullptr, nullptr, 0};
  hipLaunchKernelGGL(bn_apply_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bn_act_apply");
}

// out = act(x*sc+sh) + ract(res*rsc+rsh): the residual is a deferred tensor too (raw conv output + its BatchNorm coefficients),
// e.g. the stem's output under stage 1's residual connection -- it never has to exist in HBM in activated form.
int kd_bn_act_apply_res(const float* x, int64_t ldx, const float* sc, const float* sh, int act, const float* res, int64_t ldr,
                        const float* rsc, const float* rsh, int ract, float* out, int64_t ldo, int64_t M, int C, void* stream) {
  KD_REQUIRE(x && out && res && rsc && rsh && M > 0 && C > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bn_act_apply_res: bad args (C=%d)", C);
  KD_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0, KD_ERR_SHAPE, "kd_bn_act_apply_res: ld must be a multiple of 4");
  const KdCgLayout l = kd_cg_layout(M, C);
  ApplyArgs a{x, ldx, sc, sh, act, res, ldr, out, ldo, M, C, l.groups, l.slots, kd_nt_store((size_t)M * C * sizeof(float)), rsc, rsh, ract};
  hipLaunchKernelGGL(bn_apply_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bn_act_apply_res");
}

// partial[grid][2][C] = (sum G, sum G*xhat) with G = D * act'(X*sc+sh), xhat = (X-mean)*invstd.
int kd_bn_bwd_reduce(const float* D, int64_t ldd, const float* X, int64_t ldx, const float* sc, const float* sh,
                     int act, const float* mean, const float* invstd, float* partial, int64_t M, int C, void* stream) {
  KD_REQUIRE(D && X && mean && invstd && partial && M > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bn_bwd_reduce: bad args");
  KD_REQUIRE(act == KD_ACT_NONE || (sc && sh), KD_ERR_ARG, "kd_bn_bwd_reduce: mask needs sc/sh");
  const KdCgLayout l = kd_cg_layout(M, C);
  BwdReduceArgs a{D, ldd, X, ldx, sc, sh, act, mean, invstd, partial, M, C, l.groups, l.slots};
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bn_bwd_reduce");
}

int kd_bn_bwd_finalize(float* partial, int rows, int C, int pstride, int64_t count, const float* gamma, const float* mean,
                       const float* invstd, int training, float* dgamma, float* dbeta, float* al, float* be,
                       float* ga, float* dbias, void* stream) {
  KD_REQUIRE(partial && rows > 0 && C > 0 && pstride >= C && count > 0 && mean && invstd && al && be && ga, KD_ERR_ARG, "kd_bn_bwd_finalize: bad args");
  const int rstep = slab_prereduce(partial, rows, C, pstride, (hipStream_t)stream);
  const BwdTail t{(double)count, gamma, mean, invstd, training, dgamma, dbeta, al, be, ga, dbias};
  if (slab_is_tall(partial, rows, C, pstride))
    hipLaunchKernelGGL(bn_bwd_finalize_tall_kernel, dim3(C / 4), dim3(TL), 0, (hipStream_t)stream, partial, rows, rstep, pstride, t);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FCW - 1) / FCW), dim3(FCW, FRL), 0, (hipStream_t)stream, partial, rows, rstep, C,
                       pstride, t);
  return kd_check_launch("kd_bn_bwd_finalize");
}

}  // extern "C"
