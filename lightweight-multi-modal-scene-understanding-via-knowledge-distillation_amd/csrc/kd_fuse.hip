// kd_fuse.hip -- the small HBM-bound pieces around the GEMMs (fusion_module.py):
//   * bilinear resize (align_corners=False) of a deferred lateral, accumulated into the FPN sum
//     (CameraFPNLite.forward :58-63) and its exact adjoint;
//   * WeightedFusion's attention tail: ReLU -> 1x1 conv to 2 logits -> softmax -> weighted sum
//     (:115-120, inlined at :251-253) and its backward;
//   * the classifier 1x1 conv (Cin -> num_classes, bias) writing NCHW logits (:170,173) + backward.
#include "kd_common.h"

namespace {

// aten area_pixel_compute_source_index (align_corners=False, linear) + the index/lambda pair
__device__ __forceinline__ void bl_src(int o, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

struct BlArgs {
  const float* in; const float* sc; const float* sh; int act;     // deferred [B,Hi,Wi,C]
  float* out; int accumulate;                                      // [B,Ho,Wo,C]
  const float* dout; const float* mean; const float* invstd; float* gin; float* partial;   // bwd
  int B, Hi, Wi, Ho, Wo, C; float sh_, sw_; int groups, slots;
};

__global__ __launch_bounds__(256) void bilinear_fwd_kernel(BlArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
  if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
    const int wo = (int)(p % a.Wo), ho = (int)((p / a.Wo) % a.Ho), b = (int)(p / ((int64_t)a.Wo * a.Ho));
    int h0, h1, w0, w1;
    float lh0, lh1, lw0, lw1;
    bl_src(ho, a.Hi, a.sh_, h0, h1, lh0, lh1);
    bl_src(wo, a.Wi, a.sw_, w0, w1, lw0, lw1);
    const float* base = a.in + (int64_t)b * a.Hi * a.Wi * a.C + c0;
    const float4 v00 = kd_affine_act4(kd_ld4(base + ((int64_t)h0 * a.Wi + w0) * a.C), sc, sh, a.act);
    const float4 v01 = kd_affine_act4(kd_ld4(base + ((int64_t)h0 * a.Wi + w1) * a.C), sc, sh, a.act);
    const float4 v10 = kd_affine_act4(kd_ld4(base + ((int64_t)h1 * a.Wi + w0) * a.C), sc, sh, a.act);
    const float4 v11 = kd_affine_act4(kd_ld4(base + ((int64_t)h1 * a.Wi + w1) * a.C), sc, sh, a.act);
    float4 r;
    r.x = lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
    r.y = lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
    r.z = lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
    r.w = lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
    float* op = a.out + p * a.C + c0;
    if (a.accumulate) {
      const float4 o = kd_ld4(op);
      r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
    }
    kd_st4(op, r);
  }
}

// FPN sum in ONE pass (CameraFPNLite.forward, fusion_module.py:58-63): out = sum_i resize(act_i(in_i * sc_i + sh_i)) over up to
// three deferred laterals -- the accumulate form above reads and rewrites the [B, Ho, Wo, C] sum once per lateral.  The terms
// are added in lateral order with the same per-term expression, so the result equals the accumulate form bit for bit.
struct Bl3Args {
  const float* in[3]; const float* sc[3]; const float* sh[3]; int act[3]; int Hi[3], Wi[3]; float shh[3], sww[3]; int nin;
  float* out; int B, Ho, Wo, C, groups, slots;
};
__global__ __launch_bounds__(256) void bilinear_sum_fwd_kernel(Bl3Args a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 4;
  float4 sc[3], sh[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    sc[t] = make_float4(1.f, 1.f, 1.f, 1.f); sh[t] = kd_zero4();
    if (t < a.nin && a.sc[t]) { sc[t] = kd_ld4(a.sc[t] + c0); sh[t] = kd_ld4(a.sh[t] + c0); }
  }
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
    const int wo = (int)(p % a.Wo), ho = (int)((p / a.Wo) % a.Ho), b = (int)(p / ((int64_t)a.Wo * a.Ho));
    float4 r = kd_zero4();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t < a.nin) {
        int h0, h1, w0, w1;
        float lh0, lh1, lw0, lw1;
        bl_src(ho, a.Hi[t], a.shh[t], h0, h1, lh0, lh1);
        bl_src(wo, a.Wi[t], a.sww[t], w0, w1, lw0, lw1);
        const float* base = a.in[t] + (int64_t)b * a.Hi[t] * a.Wi[t] * a.C + c0;
        const float4 v00 = kd_affine_act4(kd_ld4(base + ((int64_t)h0 * a.Wi[t] + w0) * a.C), sc[t], sh[t], a.act[t]);
        float4 v;
        if (a.Hi[t] == a.Ho && a.Wi[t] == a.Wo && a.act[t] != KD_ACT_NONE) {
          // a lateral already at the output size (weights 1, 0, 0, 0): one load instead of four.  The general expression below gives
          // 1 * (1 * v00 + 0 * v01) + 0 * (...) = v00 + 0 for finite values, which equals v00 bit for bit unless v00 is -0.0 --
          // and an activated (ReLU / ReLU6) value never is.  (wave-uniform branch)
          v = make_float4(v00.x + 0.f, v00.y + 0.f, v00.z + 0.f, v00.w + 0.f);
        } else {
          const float4 v01 = kd_affine_act4(kd_ld4(base + ((int64_t)h0 * a.Wi[t] + w1) * a.C), sc[t], sh[t], a.act[t]);
          const float4 v10 = kd_affine_act4(kd_ld4(base + ((int64_t)h1 * a.Wi[t] + w0) * a.C), sc[t], sh[t], a.act[t]);
          const float4 v11 = kd_affine_act4(kd_ld4(base + ((int64_t)h1 * a.Wi[t] + w1) * a.C), sc[t], sh[t], a.act[t]);
          v.x = lh0 * (lw0 * v00.x + lw1 * v01.x) + lh1 * (lw0 * v10.x + lw1 * v11.x);
          v.y = lh0 * (lw0 * v00.y + lw1 * v01.y) + lh1 * (lw0 * v10.y + lw1 * v11.y);
          v.z = lh0 * (lw0 * v00.z + lw1 * v01.z) + lh1 * (lw0 * v10.z + lw1 * v11.z);
          v.w = lh0 * (lw0 * v00.w + lw1 * v01.w) + lh1 * (lw0 * v10.w + lw1 * v11.w);
        }
        if (t == 0) r = v; else { r.x = v.x + r.x; r.y = v.y + r.y; r.z = v.z + r.z; r.w = v.w + r.w; }
      }
    }
    kd_st4(a.out + p * a.C + c0, r);
  }
}

__device__ __forceinline__ float bl_weight(int o, int i, int in_size, float scale) {
  int i0, i1;
  float l0, l1;
  bl_src(o, in_size, scale, i0, i1, l0, l1);
  return (i0 == i ? l0 : 0.f) + (i1 == i ? l1 : 0.f);
}

// adjoint (gather form, deterministic): gin[b,hi,wi,c] = mask * sum_{ho,wo} W(ho,hi) W(wo,wi) dout[b,ho,wo,c]
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(BlArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mu = kd_zero4(), inv = kd_zero4();
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mu = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const int64_t npix = (int64_t)a.B * a.Hi * a.Wi;
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int wi = (int)(p % a.Wi), hi = (int)((p / a.Wi) % a.Hi), b = (int)(p / ((int64_t)a.Wi * a.Hi));
      int olo = (int)floorf(((float)hi - 0.5f) / a.sh_ - 0.5f) - 1, ohi = (int)ceilf(((float)hi + 1.5f) / a.sh_ - 0.5f) + 1;
      int plo = (int)floorf(((float)wi - 0.5f) / a.sw_ - 0.5f) - 1, phi = (int)ceilf(((float)wi + 1.5f) / a.sw_ - 0.5f) + 1;
      if (hi == 0) olo = 0;             // src < 0 clamps to index 0
      if (wi == 0) plo = 0;
      olo = olo < 0 ? 0 : olo; plo = plo < 0 ? 0 : plo;
      ohi = ohi > a.Ho - 1 ? a.Ho - 1 : ohi; phi = phi > a.Wo - 1 ? a.Wo - 1 : phi;
      float4 acc = kd_zero4();
      if (a.Hi == a.Ho && a.Wi == a.Wo) {              // same size: the adjoint is the identity (fma(1, d, 0) = d: the bits of the general loop)
        acc = kd_ld4(a.dout + p * a.C + c0);
        olo = 1; ohi = 0;                               // (skips the gather loops below)
      }
      for (int oh = olo; oh <= ohi; ++oh) {
        const float wh = bl_weight(oh, hi, a.Hi, a.sh_);
        if (wh == 0.f) continue;
        for (int ow = plo; ow <= phi; ++ow) {
          const float ww = bl_weight(ow, wi, a.Wi, a.sw_);
          if (ww == 0.f) continue;
          const float4 d = kd_ld4(a.dout + (((int64_t)b * a.Ho + oh) * a.Wo + ow) * a.C + c0);
          const float w = wh * ww;
          acc.x = fmaf(w, d.x, acc.x); acc.y = fmaf(w, d.y, acc.y); acc.z = fmaf(w, d.z, acc.z); acc.w = fmaf(w, d.w, acc.w);
        }
      }
      if (a.sc) {
        const float4 xr = kd_ld4(a.in + p * a.C + c0);
        acc.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
        acc.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
        acc.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
        acc.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x = fmaf(acc.x, (xr.x - mu.x) * inv.x, s2.x);
        s2.y = fmaf(acc.y, (xr.y - mu.y) * inv.y, s2.y);
        s2.z = fmaf(acc.z, (xr.z - mu.z) * inv.z, s2.z);
        s2.w = fmaf(acc.w, (xr.w - mu.w) * inv.w, s2.w);
      }
      kd_st4(a.gin + p * a.C + c0, acc);
    }
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

// ---- weighted fusion tail ----------------------------------------------------------------------
struct WfArgs {
  const float* cat; const float* sc; const float* sh;      // raw [M, 2C] (cam | lidar), BN+ReLU coefficients [2C]
  const float* hraw;                                       // attention.0 output (bias included) [M, C]
  const float* w2; const float* b2;                        // attention.2: [2, C], [2]
  float* out; float* wts;                                  // fwd: [M, C], [M, 2]
  const float* dout; float* dcat; float* gh; float* slab;  // bwd: dOut [M,C] -> dcat [M,2C], gh [M,C], slab [grid][3C+4]
  int64_t M; int C; int LP, slots;
};

__device__ __forceinline__ float lp_sum(float v, int LP) {
  for (int o = LP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void weighted_fuse_fwd_kernel(WfArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.LP, slot = tid / a.LP;
  const int c0 = gidx * 4;
  const float4 w20 = kd_ld4(a.w2 + c0), w21 = kd_ld4(a.w2 + a.C + c0);
  const float4 scc = kd_ld4(a.sc + c0), shc = kd_ld4(a.sh + c0), scl = kd_ld4(a.sc + a.C + c0), shl = kd_ld4(a.sh + a.C + c0);
  const float b20 = a.b2[0], b21 = a.b2[1];
  const int64_t iters = (a.M + (int64_t)gridDim.x * a.slots - 1) / ((int64_t)gridDim.x * a.slots);
  for (int64_t it = 0; it < iters; ++it) {                 // uniform trip count: shuffles need all lanes
    const int64_t m = (it * gridDim.x + blockIdx.x) * a.slots + slot;
    const bool ok = m < a.M;
    float4 h = kd_zero4(), cp = kd_zero4(), lp = kd_zero4();
    if (ok) {
      h = kd_affine_act4(kd_ld4(a.hraw + m * a.C + c0), make_float4(1.f, 1.f, 1.f, 1.f), kd_zero4(), KD_ACT_RELU);
      cp = kd_affine_act4(kd_ld4(a.cat + m * 2 * a.C + c0), scc, shc, KD_ACT_RELU);
      lp = kd_affine_act4(kd_ld4(a.cat + m * 2 * a.C + a.C + c0), scl, shl, KD_ACT_RELU);
    }
    float a0 = h.x * w20.x + h.y * w20.y + h.z * w20.z + h.w * w20.w;
    float a1 = h.x * w21.x + h.y * w21.y + h.z * w21.z + h.w * w21.w;
    a0 = lp_sum(a0, a.LP) + b20;
    a1 = lp_sum(a1, a.LP) + b21;
    const float mx = fmaxf(a0, a1);
    const float e0 = expf(a0 - mx), e1 = expf(a1 - mx);
    const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
    if (ok) {
      float4 r;
      r.x = cp.x * w0 + lp.x * w1; r.y = cp.y * w0 + lp.y * w1; r.z = cp.z * w0 + lp.z * w1; r.w = cp.w * w0 + lp.w * w1;
      kd_st4(a.out + m * a.C + c0, r);
      if (gidx == 0) { a.wts[m * 2] = w0; a.wts[m * 2 + 1] = w1; }
    }
  }
}

__global__ __launch_bounds__(256) void weighted_fuse_bwd_kernel(WfArgs a) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.LP, slot = tid / a.LP;
  const int c0 = gidx * 4;
  const float4 w20 = kd_ld4(a.w2 + c0), w21 = kd_ld4(a.w2 + a.C + c0);
  const float4 scc = kd_ld4(a.sc + c0), shc = kd_ld4(a.sh + c0), scl = kd_ld4(a.sc + a.C + c0), shl = kd_ld4(a.sh + a.C + c0);
  float4 dw0 = kd_zero4(), dw1 = kd_zero4(), db1 = kd_zero4(), db2 = kd_zero4();
  const int64_t iters = (a.M + (int64_t)gridDim.x * a.slots - 1) / ((int64_t)gridDim.x * a.slots);
  for (int64_t it = 0; it < iters; ++it) {
    const int64_t m = (it * gridDim.x + blockIdx.x) * a.slots + slot;
    const bool ok = m < a.M;
    float4 h = kd_zero4(), cp = kd_zero4(), lp = kd_zero4(), d = kd_zero4();
    float w0 = 0.f, w1 = 0.f;
    if (ok) {
      h = kd_affine_act4(kd_ld4(a.hraw + m * a.C + c0), make_float4(1.f, 1.f, 1.f, 1.f), kd_zero4(), KD_ACT_RELU);
      cp = kd_affine_act4(kd_ld4(a.cat + m * 2 * a.C + c0), scc, shc, KD_ACT_RELU);
      lp = kd_affine_act4(kd_ld4(a.cat + m * 2 * a.C + a.C + c0), scl, shl, KD_ACT_RELU);
      d = kd_ld4(a.dout + m * a.C + c0);
      w0 = a.wts[m * 2]; w1 = a.wts[m * 2 + 1];
    }
    float g0 = d.x * cp.x + d.y * cp.y + d.z * cp.z + d.w * cp.w;      // dL/dw0 partial
    float g1 = d.x * lp.x + d.y * lp.y + d.z * lp.z + d.w * lp.w;
    g0 = lp_sum(g0, a.LP);
    g1 = lp_sum(g1, a.LP);
    const float dot = w0 * g0 + w1 * g1;
    const float da0 = w0 * (g0 - dot), da1 = w1 * (g1 - dot);              // softmax backward
    if (ok) {
      kd_st4(a.dcat + m * 2 * a.C + c0, make_float4(d.x * w0, d.y * w0, d.z * w0, d.w * w0));
      kd_st4(a.dcat + m * 2 * a.C + a.C + c0, make_float4(d.x * w1, d.y * w1, d.z * w1, d.w * w1));
      float4 gh;
      gh.x = h.x > 0.f ? da0 * w20.x + da1 * w21.x : 0.f;
      gh.y = h.y > 0.f ? da0 * w20.y + da1 * w21.y : 0.f;
      gh.z = h.z > 0.f ? da0 * w20.z + da1 * w21.z : 0.f;
      gh.w = h.w > 0.f ? da0 * w20.w + da1 * w21.w : 0.f;
      kd_st4(a.gh + m * a.C + c0, gh);
      dw0.x = fmaf(da0, h.x, dw0.x); dw0.y = fmaf(da0, h.y, dw0.y); dw0.z = fmaf(da0, h.z, dw0.z); dw0.w = fmaf(da0, h.w, dw0.w);
      dw1.x = fmaf(da1, h.x, dw1.x); dw1.y = fmaf(da1, h.y, dw1.y); dw1.z = fmaf(da1, h.z, dw1.z); dw1.w = fmaf(da1, h.w, dw1.w);
      db1.x += gh.x; db1.y += gh.y; db1.z += gh.z; db1.w += gh.w;
      if (gidx == 0) { db2.x += da0; db2.y += da1; }
    }
  }
  float* out = a.slab + (int64_t)blockIdx.x * (3 * a.C + 4);
  const float4 accs[4] = {dw0, dw1, db1, db2};
  for (int t = 0; t < 4; ++t) {
    __syncthreads();
    kd_st4(red + tid * 4, accs[t]);
    __syncthreads();
    const int width = t < 3 ? a.C : 4;
    for (int c = tid; c < width; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.LP + c / 4) * 4 + (c & 3)];
      out[t * a.C + c] = s;
    }
  }
}

// ---- classifier ---------------------------------------------------------------------------------
struct ClsArgs {
  const float* x; const float* sc; const float* sh; int act;       // deferred [M, Cin]
  const float* w; const float* b;                                   // [NC, Cin], [NC]
  float* logits;                                                    // NCHW [B, NC, HW]
  const float* dlog; const float* mean; const float* invstd;        // bwd
  float* gx; float* partial; float* wslab;                          // [M,Cin]; [grid][2][Cin]; [grid][NC*Cin+4]
  int64_t M; int HW; int Cin; int NC; int groups, slots;
};

__global__ __launch_bounds__(256) void cls_fwd_kernel(ClsArgs a) {
  __shared__ float ws[4 * 64 + 4];
  for (int i = threadIdx.x; i < a.NC * a.Cin; i += 256) ws[i] = a.w[i];
  __syncthreads();
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < a.Cin; c += 4) {
    float4 v = kd_ld4(a.x + m * a.Cin + c);
    if (a.sc) v = kd_affine_act4(v, kd_ld4(a.sc + c), kd_ld4(a.sh + c), a.act);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < a.NC) {
        const float* wj = ws + j * a.Cin + c;
        acc[j] = fmaf(v.w, wj[3], fmaf(v.z, wj[2], fmaf(v.y, wj[1], fmaf(v.x, wj[0], acc[j]))));
      }
  }
  const int64_t b = m / a.HW, hw = m % a.HW;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < a.NC) a.logits[(b * a.NC + j) * a.HW + hw] = acc[j] + (a.b ? a.b[j] : 0.f);
}

__global__ __launch_bounds__(256) void cls_bwd_kernel(ClsArgs a) {
  __shared__ float red[256 * 4];
  __shared__ float ws[4 * 64 + 4];
  for (int i = threadIdx.x; i < a.NC * a.Cin; i += 256) ws[i] = a.w[i];
  __syncthreads();
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4(), dwj[4], dbj = kd_zero4();
#pragma unroll
  for (int j = 0; j < 4; ++j) dwj[j] = kd_zero4();
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mu = kd_zero4(), inv = kd_zero4();
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mu = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    for (int64_t m = (int64_t)blockIdx.x * a.slots + slot; m < a.M; m += (int64_t)gridDim.x * a.slots) {
      const int64_t b = m / a.HW, hw = m % a.HW;
      const float4 xr = kd_ld4(a.x + m * a.Cin + c0);
      const float4 xa = a.sc ? kd_affine_act4(xr, sc, sh, a.act) : xr;
      float4 g = kd_zero4();
      float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < a.NC) {
          dl[j] = a.dlog[(b * a.NC + j) * a.HW + hw];
          const float* wj = ws + j * a.Cin + c0;
          g.x = fmaf(dl[j], wj[0], g.x); g.y = fmaf(dl[j], wj[1], g.y); g.z = fmaf(dl[j], wj[2], g.z); g.w = fmaf(dl[j], wj[3], g.w);
          dwj[j].x = fmaf(dl[j], xa.x, dwj[j].x); dwj[j].y = fmaf(dl[j], xa.y, dwj[j].y);
          dwj[j].z = fmaf(dl[j], xa.z, dwj[j].z); dwj[j].w = fmaf(dl[j], xa.w, dwj[j].w);
        }
      if (gidx == 0) { dbj.x += dl[0]; dbj.y += dl[1]; dbj.z += dl[2]; dbj.w += dl[3]; }
      if (a.sc) {
        g.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
        g.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
        g.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
        g.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, (xr.x - mu.x) * inv.x, s2.x); s2.y = fmaf(g.y, (xr.y - mu.y) * inv.y, s2.y);
        s2.z = fmaf(g.z, (xr.z - mu.z) * inv.z, s2.z); s2.w = fmaf(g.w, (xr.w - mu.w) * inv.w, s2.w);
      }
      kd_st4(a.gx + m * a.Cin + c0, g);
    }
  }
  // block reductions: 2 stat rows, NC weight rows, 1 bias row
  for (int t = 0; t < 7; ++t) {
    if (t >= 2 && t < 6 && t - 2 >= a.NC) continue;
    float4 v = t == 0 ? s1 : t == 1 ? s2 : t < 6 ? dwj[t - 2] : dbj;
    __syncthreads();
    kd_st4(red + tid * 4, v);
    __syncthreads();
    const int width = t < 6 ? a.Cin : 4;
    for (int c = tid; c < width; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      if (t < 2) { if (a.partial) a.partial[((int64_t)blockIdx.x * 2 + t) * a.Cin + c] = s; }
      else if (t < 6) a.wslab[(int64_t)blockIdx.x * (a.NC * a.Cin + 4) + (t - 2) * a.Cin + c] = s;
      else a.wslab[(int64_t)blockIdx.x * (a.NC * a.Cin + 4) + a.NC * a.Cin + c] = s;
    }
  }
}

}  // namespace

extern "C" {

// out (+)= bilinear_resize(act(in*sc+sh)) ; align_corners=False; identity when sizes match.
int kd_bilinear_accum_fwd(const float* in, const float* sc, const float* sh, int act, float* out, int accumulate,
                          int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
  KD_REQUIRE(in && out && B > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bilinear_accum_fwd: bad args");
  const KdCgLayout l = kd_cg_layout((int64_t)B * Ho * Wo, C);
  BlArgs a{in, sc, sh, act, out, accumulate, nullptr, nullptr, nullptr, nullptr, nullptr, B, Hi, Wi, Ho, Wo, C,
           (float)Hi / (float)Ho, (float)Wi / (float)Wo, l.groups, l.slots};
  hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bilinear_accum_fwd");
}

// out = sum over up to three deferred laterals of bilinear_resize(act_i(in_i*sc_i+sh_i)), one pass (in_i == NULL: absent).
int kd_bilinear_sum_fwd(const float* in0, const float* sc0, const float* sh0, int act0, int H0, int W0, const float* in1,
                        const float* sc1, const float* sh1, int act1, int H1, int W1, const float* in2, const float* sc2,
                        const float* sh2, int act2, int H2, int W2, float* out, int B, int Ho, int Wo, int C, void* stream) {
  KD_REQUIRE(in0 && out && B > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bilinear_sum_fwd: bad args");
  const KdCgLayout l = kd_cg_layout((int64_t)B * Ho * Wo, C);
  Bl3Args a{};
  const float* ins[3] = {in0, in1, in2};
  const float* scs[3] = {sc0, sc1, sc2};
  const float* shs[3] = {sh0, sh1, sh2};
  const int acts[3] = {act0, act1, act2}, hs[3] = {H0, H1, H2}, wsz[3] = {W0, W1, W2};
  a.nin = 0;
  for (int t = 0; t < 3; ++t)
    if (ins[t]) {
      KD_REQUIRE(t == a.nin, KD_ERR_ARG, "kd_bilinear_sum_fwd: laterals must be given without gaps");
      a.in[t] = ins[t]; a.sc[t] = scs[t]; a.sh[t] = shs[t]; a.act[t] = acts[t]; a.Hi[t] = hs[t]; a.Wi[t] = wsz[t];
      a.shh[t] = (float)hs[t] / (float)Ho; a.sww[t] = (float)wsz[t] / (float)Wo;
      ++a.nin;
    }
  a.out = out; a.B = B; a.Ho = Ho; a.Wo = Wo; a.C = C; a.groups = l.groups; a.slots = l.slots;
  hipLaunchKernelGGL(bilinear_sum_fwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bilinear_sum_fwd");
}

// gin = act'(.) * adjoint_resize(dout); partial (rows = kd_rowwise_stat_rows(B*Hi*Wi, C)) gets (sum G, sum G*xhat).
int kd_bilinear_bwd(const float* dout, const float* in, const float* sc, const float* sh, int act, const float* mean,
                    const float* invstd, float* gin, float* partial, int B, int Hi, int Wi, int Ho, int Wo, int C,
                    void* stream) {
  KD_REQUIRE(dout && gin && B > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_bilinear_bwd: bad args");
  KD_REQUIRE(!sc || (in && sh && (!partial || (mean && invstd))), KD_ERR_ARG, "kd_bilinear_bwd: mask needs in/sh (+mean/invstd)");
  const KdCgLayout l = kd_cg_layout((int64_t)B * Hi * Wi, C);
  BlArgs a{in, sc, sh, act, nullptr, 0, dout, mean, invstd, gin, sc ? partial : nullptr, B, Hi, Wi, Ho, Wo, C,
           (float)Hi / (float)Ho, (float)Wi / (float)Wo, l.groups, l.slots};
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bilinear_bwd");
}

static int wf_lp(int C) { return (C == 32 || C == 64 || C == 128 || C == 256) ? C / 4 : 0; }

int kd_weighted_fuse_fwd(const float* cat, const float* sc, const float* sh, const float* hraw, const float* w2,
                         const float* b2, float* out, float* wts, int64_t M, int C, void* stream) {
  KD_REQUIRE(cat && sc && sh && hraw && w2 && b2 && out && wts && M > 0, KD_ERR_ARG, "kd_weighted_fuse_fwd: bad args");
  const int LP = wf_lp(C);
  KD_REQUIRE(LP, KD_ERR_SHAPE, "kd_weighted_fuse_fwd: C=%d must be 32/64/128/256", C);
  const int slots = 256 / LP;
  int64_t grid = (M + slots - 1) / slots;
  if (grid > 4096) grid = 4096;
  WfArgs a{cat, sc, sh, hraw, w2, b2, out, wts, nullptr, nullptr, nullptr, nullptr, M, C, LP, slots};
  hipLaunchKernelGGL(weighted_fuse_fwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_weighted_fuse_fwd");
}

static int64_t wf_bwd_grid(int64_t M, int C) {
  const int slots = 256 / (C / 4);
  int64_t grid = (M + slots - 1) / slots;
  return grid > 1024 ? 1024 : grid;
}
size_t kd_weighted_fuse_bwd_ws_bytes(int64_t M, int C) { return (size_t)wf_bwd_grid(M, C) * (3 * C + 4) * sizeof(float); }

// dcat[M,2C] = [dout*w0 | dout*w1] (unmasked), gh[M,C] = dL/d(attention.0 raw output),
// dparams[3C+4] = dW2 (2C) | db1 (C) | db2 (2) | pad.
int kd_weighted_fuse_bwd(const float* dout, const float* cat, const float* sc, const float* sh, const float* hraw,
                         const float* w2, const float* wts, float* dcat, float* gh, float* dparams, int64_t M, int C,
                         void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(dout && cat && sc && sh && hraw && w2 && wts && dcat && gh && dparams && ws && M > 0, KD_ERR_ARG, "kd_weighted_fuse_bwd: bad args");
  const int LP = wf_lp(C);
  KD_REQUIRE(LP, KD_ERR_SHAPE, "kd_weighted_fuse_bwd: C=%d must be 32/64/128/256", C);
  const int64_t grid = wf_bwd_grid(M, C);
  KD_REQUIRE(ws_bytes >= (size_t)grid * (3 * C + 4) * sizeof(float), KD_ERR_WORKSPACE, "kd_weighted_fuse_bwd: workspace too small");
  WfArgs a{cat, sc, sh, hraw, w2, nullptr, nullptr, const_cast<float*>(wts), dout, dcat, gh, (float*)ws, M, C, LP, 256 / LP};
  hipLaunchKernelGGL(weighted_fuse_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  int rc = kd_check_launch("kd_weighted_fuse_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, (int)grid, 3 * C + 4, dparams, (hipStream_t)stream);
}

int kd_cls_conv_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, const float* b,
                    float* logits_nchw, int64_t M, int HW, int Cin, int NC, void* stream) {
  KD_REQUIRE(x && w && logits_nchw && M > 0 && HW > 0 && M % HW == 0, KD_ERR_ARG, "kd_cls_conv_fwd: bad args");
  KD_REQUIRE(Cin % 4 == 0 && Cin <= 64 && NC >= 1 && NC <= 4, KD_ERR_SHAPE, "kd_cls_conv_fwd: Cin=%d NC=%d unsupported", Cin, NC);
  ClsArgs a{x, sc, sh, act, w, b, logits_nchw, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, HW, Cin, NC, 0, 0};
  hipLaunchKernelGGL(cls_fwd_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_cls_conv_fwd");
}

size_t kd_cls_conv_bwd_ws_bytes(int64_t M, int Cin, int NC) {
  return (size_t)kd_cg_layout(M, Cin, 1024).grid * (NC * Cin + 4) * sizeof(float);
}
int64_t kd_cls_conv_bwd_stat_rows(int64_t M, int Cin) { return kd_cg_layout(M, Cin, 1024).grid; }

// gx[M,Cin] = act'(.) * (dlog . W); dwb = dW [NC*Cin] | db [NC] (padded to 4).
int kd_cls_conv_bwd(const float* dlog_nchw, const float* x, const float* sc, const float* sh, int act,
                    const float* mean, const float* invstd, const float* w, float* gx, float* partial, float* dwb,
                    int64_t M, int HW, int Cin, int NC, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(dlog_nchw && x && w && gx && dwb && ws && M > 0 && M % HW == 0, KD_ERR_ARG, "kd_cls_conv_bwd: bad args");
  KD_REQUIRE(Cin % 4 == 0 && Cin <= 64 && NC >= 1 && NC <= 4, KD_ERR_SHAPE, "kd_cls_conv_bwd: Cin=%d NC=%d unsupported", Cin, NC);
  const KdCgLayout l = kd_cg_layout(M, Cin, 1024);
  KD_REQUIRE(ws_bytes >= (size_t)l.grid * (NC * Cin + 4) * sizeof(float), KD_ERR_WORKSPACE, "kd_cls_conv_bwd: workspace too small");
  ClsArgs a{x, sc, sh, act, w, nullptr, nullptr, dlog_nchw, mean, invstd, gx, partial, (float*)ws, M, HW, Cin, NC, l.groups, l.slots};
  hipLaunchKernelGGL(cls_bwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  int rc = kd_check_launch("kd_cls_conv_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, l.grid, NC * Cin + 4, dwb, (hipStream_t)stream);
}

}  // extern "C"
