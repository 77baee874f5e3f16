// kd_loss.hip -- loss, metric and optimiser kernels of the KD step.
//   * weighted cross-entropy with ignore_index (trainer.py:55,88) fused with the temperature-softmax
//     KL term of the KD objective (build-defined) -- forward value and
//     dL/dlogits in one call;
//   * feature MSE forward + gradient;
//   * argmax + confusion matrix (SegmentationMetrics.update, trainer.py:18-26) -- integer exact;
//   * AdamW over flat parameter / gradient / moment buffers (trainer.py:56, torch.optim.AdamW math).
#include "kd_common.h"

namespace {

constexpr int MAXC = 4;

struct SegArgs {
  const float* zs; const float* zt; const int64_t* target; const float* cw;
  int ignore_index; float T; float alpha; float gscale; const float* gdev;
  float* slab; float* losses; float* dzs;
  int64_t npix; int HW; int NC;
};

__device__ __forceinline__ void softmax_c(const float* z, int NC, float invT, float* p, float* logp) {
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) if (j < NC) mx = fmaxf(mx, z[j] * invT);
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) if (j < NC) { p[j] = expf(z[j] * invT - mx); s += p[j]; }
  const float ls = logf(s);
#pragma unroll
  for (int j = 0; j < MAXC; ++j) if (j < NC) { logp[j] = z[j] * invT - mx - ls; p[j] = p[j] / s; }
}

__device__ __forceinline__ void block_sum3(float a, float b, float c, float* out3) {
  __shared__ float red[3][4];
  a = kd_wave_sum(a); b = kd_wave_sum(b); c = kd_wave_sum(c);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
  __syncthreads();
  if (threadIdx.x < 3) out3[threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void seg_loss_partial_kernel(SegArgs a) {
  float swn = 0.f, sw = 0.f, skl = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.npix; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / a.HW, hw = i % a.HW;
    float z[MAXC], p[MAXC], lp[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) if (j < a.NC) z[j] = a.zs[(b * a.NC + j) * a.HW + hw];
    const int64_t y = a.target[i];
    if (y != a.ignore_index && y >= 0 && y < a.NC) {
      softmax_c(z, a.NC, 1.f, p, lp);
      float nll = 0.f;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j == (int)y) nll = -lp[j];
      const float w = a.cw ? a.cw[y] : 1.f;
      swn = fmaf(w, nll, swn);
      sw += w;
    }
    if (a.zt) {
      float zt[MAXC], pt[MAXC], lpt[MAXC];
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j < a.NC) zt[j] = a.zt[(b * a.NC + j) * a.HW + hw];
      softmax_c(z, a.NC, 1.f / a.T, p, lp);
      softmax_c(zt, a.NC, 1.f / a.T, pt, lpt);
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j < a.NC) skl += pt[j] > 0.f ? pt[j] * (lpt[j] - lp[j]) : 0.f;
    }
  }
  block_sum3(swn, sw, skl, a.slab + (int64_t)blockIdx.x * 4);
}

__global__ void seg_loss_final_kernel(const float* slab, int nblk, double npix, float* losses) {
  // losses: [0] CE (weighted mean), [1] KL (per-pixel mean), [2] sum of weights
  __shared__ double red[3][256];
  double s[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblk; i += 256)
    for (int k = 0; k < 3; ++k) s[k] += (double)slab[(int64_t)i * 4 + k];
  for (int k = 0; k < 3; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < 256; ++i) for (int k = 0; k < 3; ++k) t[k] += red[k][i];
    losses[0] = (float)(t[0] / t[1]);
    losses[1] = (float)(t[2] / npix);
    losses[2] = (float)t[1];
  }
}

__global__ __launch_bounds__(256) void seg_loss_grad_kernel(SegArgs a) {
  const float sumw = a.losses[2];
  const float gs = a.gscale * (a.gdev ? a.gdev[0] : 1.f);      // upstream gradient as a device scalar
  const float klc = a.zt ? gs * a.alpha * a.T / (float)a.npix : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.npix; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / a.HW, hw = i % a.HW;
    float z[MAXC], p[MAXC], lp[MAXC], g[MAXC];
#pragma unroll
    for (int j = 0; j < MAXC; ++j) { g[j] = 0.f; if (j < a.NC) z[j] = a.zs[(b * a.NC + j) * a.HW + hw]; }
    const int64_t y = a.target[i];
    if (y != a.ignore_index && y >= 0 && y < a.NC) {
      softmax_c(z, a.NC, 1.f, p, lp);
      const float w = (a.cw ? a.cw[y] : 1.f) * gs / sumw;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j < a.NC) g[j] = w * (p[j] - (j == (int)y ? 1.f : 0.f));
    }
    if (a.zt) {
      float zt[MAXC], pt[MAXC], lpt[MAXC];
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j < a.NC) zt[j] = a.zt[(b * a.NC + j) * a.HW + hw];
      softmax_c(z, a.NC, 1.f / a.T, p, lp);
      softmax_c(zt, a.NC, 1.f / a.T, pt, lpt);
#pragma unroll
      for (int j = 0; j < MAXC; ++j) if (j < a.NC) g[j] = fmaf(klc, p[j] - pt[j], g[j]);
    }
#pragma unroll
    for (int j = 0; j < MAXC; ++j) if (j < a.NC) a.dzs[(b * a.NC + j) * a.HW + hw] = g[j];
  }
}

__global__ __launch_bounds__(256) void mse_kernel(const float* a, const float* b, int64_t n4, float gcoef,
                                                  const float* gdev, float* da, float* slab) {
  float s = 0.f;
  if (gdev) gcoef *= gdev[0];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 x = kd_ld4(a + i * 4), y = kd_ld4(b + i * 4);
    const float4 d = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
    s = fmaf(d.x, d.x, fmaf(d.y, d.y, fmaf(d.z, d.z, fmaf(d.w, d.w, s))));
    if (da) kd_st4(da + i * 4, make_float4(gcoef * d.x, gcoef * d.y, gcoef * d.z, gcoef * d.w));
  }
  __shared__ float sh3[3];
  block_sum3(s, 0.f, 0.f, sh3);
  __syncthreads();
  if (threadIdx.x == 0 && slab) slab[blockIdx.x] = sh3[0];
}

// total = ce + ckl * kl + beta * (mse_c + mse_l), rounded after every operation like the fp32 tensor expression it replaces
__global__ void kd_total_kernel(const float* ce_kl, const float* mse_c, const float* mse_l, float ckl, float beta, float* total) {
  if (threadIdx.x == 0) {
    const float t = __fadd_rn(ce_kl[0], __fmul_rn(ckl, ce_kl[1]));
    const float m = __fadd_rn(mse_c ? mse_c[0] : 0.f, mse_l ? mse_l[0] : 0.f);
    total[0] = __fadd_rn(t, __fmul_rn(beta, m));
  }
}

// the two feature-MSE values from their per-block partial sums (the reduction of mse_final_kernel, same order) and the KD objective's
// total in one launch: out[0] = mse_c, out[1] = mse_l, out[2] = ce_kl[0] + ckl * ce_kl[1] + beta * (mse_c + mse_l)
__global__ void kd_objective_final_kernel(const float* ce_kl, const float* slab_c, int nblk_c, double n_c, const float* slab_l, int nblk_l,
                                          double n_l, float ckl, float beta, float* out) {
  __shared__ double red[2][256];
  double sc = 0.0, sl = 0.0;
  for (int i = threadIdx.x; i < nblk_c; i += 256) sc += (double)slab_c[i];
  for (int i = threadIdx.x; i < nblk_l; i += 256) sl += (double)slab_l[i];
  red[0][threadIdx.x] = sc;
  red[1][threadIdx.x] = sl;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tc = 0.0, tl = 0.0;
    for (int i = 0; i < 256; ++i) { tc += red[0][i]; tl += red[1][i]; }
    const float mc = (float)(tc / n_c), ml = (float)(tl / n_l);
    out[0] = mc;
    out[1] = ml;
    out[2] = __fadd_rn(__fadd_rn(ce_kl[0], __fmul_rn(ckl, ce_kl[1])), __fmul_rn(beta, __fadd_rn(mc, ml)));
  }
}

__global__ void mse_final_kernel(const float* slab, int nblk, double n, float* loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)slab[i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 256; ++i) t += red[i];
    *loss = (float)(t / n);
  }
}

__global__ __launch_bounds__(256) void confusion_kernel(const float* z, const int64_t* target, int ignore_index,
                                                        int64_t npix, int HW, int NC, unsigned long long* conf,
                                                        int64_t* pred) {
  __shared__ unsigned int loc[MAXC * MAXC];
  if (threadIdx.x < MAXC * MAXC) loc[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / HW, hw = i % HW;
    int best = 0;
    float bv = z[(b * NC) * HW + hw];
    for (int j = 1; j < NC; ++j) {
      const float v = z[(b * NC + j) * HW + hw];
      if (v > bv) { bv = v; best = j; }           // first maximum wins, like torch.argmax
    }
    if (pred) pred[i] = best;
    const int64_t t = target ? target[i] : ignore_index;
    if (target && t != ignore_index && t >= 0 && t < NC) atomicAdd(&loc[t * NC + best], 1u);
  }
  __syncthreads();
  if (conf && threadIdx.x < NC * NC && loc[threadIdx.x]) atomicAdd(&conf[threadIdx.x], (unsigned long long)loc[threadIdx.x]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2sqrt,
                                                    float ginv) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * ginv;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

}  // namespace

extern "C" {

size_t kd_seg_loss_ws_bytes(int64_t npix) {
  int64_t g = (npix + 255) / 256;
  if (g > 1024) g = 1024;
  return (size_t)g * 4 * sizeof(float);
}

// losses[0] = CE_w(zs, target), losses[1] = mean-per-pixel KL(softmax(zt/T) || softmax(zs/T)) (0 if zt null),
// losses[2] = sum of class weights over kept pixels.
// dzs = gscale * d/dzs ( CE + alpha*T^2*KL ).   zt == null: CE only.   dzs == null: forward only.
int kd_seg_loss_fwd_bwd(const float* zs, const float* zt, const int64_t* target, const float* class_w,
                        int ignore_index, float T, float alpha, float gscale, const float* gscale_dev, float* losses,
                        float* dzs, int B, int NC, int HW, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(zs && target && losses && ws && B > 0 && HW > 0, KD_ERR_ARG, "kd_seg_loss_fwd_bwd: bad args");
  KD_REQUIRE(NC >= 2 && NC <= MAXC, KD_ERR_SHAPE, "kd_seg_loss_fwd_bwd: num_classes=%d unsupported (2..4)", NC);
  const int64_t npix = (int64_t)B * HW;
  int64_t grid = (npix + 255) / 256;
  if (grid > 1024) grid = 1024;
  KD_REQUIRE(ws_bytes >= (size_t)grid * 4 * sizeof(float), KD_ERR_WORKSPACE, "kd_seg_loss_fwd_bwd: workspace too small");
  SegArgs a{zs, zt, target, class_w, ignore_index, T, alpha, gscale, gscale_dev, (float*)ws, losses, dzs, npix, HW, NC};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_loss_partial_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)grid, (double)npix, losses);
  if (dzs) hipLaunchKernelGGL(seg_loss_grad_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
  return kd_check_launch("kd_seg_loss_fwd_bwd");
}

size_t kd_mse_ws_bytes(int64_t n) {
  int64_t g = (n / 4 + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (size_t)g * sizeof(float);
}

// loss = mean((a-b)^2) (skipped when loss == null); da = gcoef * gscale_dev[0] * (a-b) (skipped when
// da == null; gscale_dev may be null).  Pass gcoef = 2*beta/n.
int kd_mse_fwd_bwd(const float* a, const float* b, int64_t n, float gcoef, const float* gscale_dev, float* loss,
                   float* da, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(a && b && (loss || da) && ws && n > 0 && n % 4 == 0, KD_ERR_ARG, "kd_mse_fwd_bwd: bad args (n must be a multiple of 4)");
  int64_t grid = (n / 4 + 255) / 256;
  if (grid > 2048) grid = 2048;
  KD_REQUIRE(ws_bytes >= (size_t)grid * sizeof(float), KD_ERR_WORKSPACE, "kd_mse_fwd_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)grid), dim3(256), 0, st, a, b, n / 4, gcoef, gscale_dev, da,
                     loss ? (float*)ws : nullptr);
  if (loss) hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, (int)grid, (double)n, loss);
  return kd_check_launch("kd_mse_fwd_bwd");
}

// The KD objective's value from its parts: total = ce_kl[0] + ckl * ce_kl[1] + beta * (mse_c + mse_l),
// each operation rounded to fp32 in this order (what `ce + (alpha*T*T) * kl + beta * (mse_c + mse_l)` on fp32 scalars gives).
int kd_kd_total(const float* ce_kl, const float* mse_c, const float* mse_l, float ckl, float beta, float* total, void* stream) {
  KD_REQUIRE(ce_kl && total, KD_ERR_ARG, "kd_kd_total: bad args");
  hipLaunchKernelGGL(kd_total_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ce_kl, mse_c, mse_l, ckl, beta, total);
  return kd_check_launch("kd_kd_total");
}

// kd_mse_fwd_bwd without its one-block final reduction: the per-block sums of (a-b)^2 go to `slab` (kd_mse_slab_blocks(n) floats) for
// kd_kd_objective_final, da = gcoef * (a-b) as in kd_mse_fwd_bwd (skipped when NULL).
int64_t kd_mse_slab_blocks(int64_t n) { int64_t g = (n / 4 + 255) / 256; return g > 2048 ? 2048 : (g < 1 ? 1 : g); }
int kd_mse_partial(const float* a, const float* b, int64_t n, float gcoef, float* da, float* slab, void* stream) {
  KD_REQUIRE(a && b && slab && n > 0 && n % 4 == 0, KD_ERR_ARG, "kd_mse_partial: bad args (n must be a multiple of 4)");
  hipLaunchKernelGGL(mse_kernel, dim3((unsigned)kd_mse_slab_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n / 4, gcoef, (const float*)nullptr, da, slab);
  return kd_check_launch("kd_mse_partial");
}

// out[0] = mean((a_c-b_c)^2), out[1] = mean((a_l-b_l)^2) from the slabs of two kd_mse_partial calls over n_c / n_l elements, and
// out[2] = ce_kl[0] + ckl * ce_kl[1] + beta * (out[0] + out[1]) (the rounding order of kd_kd_total): three tiny launches in one.
int kd_kd_objective_final(const float* ce_kl, const float* slab_c, int64_t n_c, const float* slab_l, int64_t n_l, float ckl, float beta,
                          float* out, void* stream) {
  KD_REQUIRE(ce_kl && slab_c && slab_l && out && n_c > 0 && n_l > 0, KD_ERR_ARG, "kd_kd_objective_final: bad args");
  hipLaunchKernelGGL(kd_objective_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ce_kl, slab_c, (int)kd_mse_slab_blocks(n_c), (double)n_c,
                     slab_l, (int)kd_mse_slab_blocks(n_l), (double)n_l, ckl, beta, out);
  return kd_check_launch("kd_kd_objective_final");
}

// conf[NC*NC] (uint64, ACCUMULATED into) and/or pred[B*HW] (int64 argmax over the class dim).
int kd_argmax_confusion(const float* logits, const int64_t* target, int ignore_index, uint64_t* conf, int64_t* pred,
                        int B, int NC, int HW, void* stream) {
  KD_REQUIRE(logits && (conf || pred) && B > 0 && NC >= 1 && NC <= MAXC, KD_ERR_ARG, "kd_argmax_confusion: bad args");
  const int64_t npix = (int64_t)B * HW;
  int64_t grid = (npix + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, logits, target,
                     ignore_index, npix, HW, NC, (unsigned long long*)conf, pred);
  return kd_check_launch("kd_argmax_confusion");
}

// One AdamW step over flat buffers; `step` is the 1-based step count; grads are scaled by ginv first
// (1/world_size after a sum all-reduce).
int kd_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float ginv, void* stream) {
  KD_REQUIRE(p && g && m && v && n > 0 && step >= 1, KD_ERR_ARG, "kd_adamw_step: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                     beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2), ginv);
  return kd_check_launch("kd_adamw_step");
}

// Graph-replay-safe AdamW: everything that changes from step to step lives in a device array
//   state[0] = lr   state[1] = step count (as float)   state[2] = 1 - beta1^step   state[3] = sqrt(1 - beta2^step)
// The tick kernel advances the step and refreshes the bias corrections; the update kernel reads them.
// A captured hipGraph of the training step therefore stays correct on every replay (a host-side
// `step` argument would be frozen at capture time).  lr is changed by writing state[0] between replays.
__global__ void adamw_tick_kernel(float* state, float b1, float b2) {
  const float t = state[1] + 1.f;
  state[1] = t;
  state[2] = (float)(1.0 - pow((double)b1, (double)t));
  state[3] = (float)sqrt(1.0 - pow((double)b2, (double)t));
}
__global__ __launch_bounds__(256) void adamw_dev_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                        const float* state, float b1, float b2, float eps, float wd,
                                                        float ginv) {
  const float lr = state[0], bc1 = state[2], bc2sqrt = state[3];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gi = g[i] * ginv;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}
int kd_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, float beta1, float beta2,
                      float eps, float weight_decay, float ginv, void* stream) {
  KD_REQUIRE(p && g && m && v && state && n > 0, KD_ERR_ARG, "kd_adamw_step_dev: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(adamw_tick_kernel, dim3(1), dim3(1), 0, st, state, beta1, beta2);
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)grid), dim3(256), 0, st, p, g, m, v, n, (const float*)state, beta1,
                     beta2, eps, weight_decay, ginv);
  return kd_check_launch("kd_adamw_step_dev");
}

}  // extern "C"
