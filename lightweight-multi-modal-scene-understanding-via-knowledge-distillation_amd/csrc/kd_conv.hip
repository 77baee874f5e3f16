// kd_conv.hip -- the spatial convolutions of the camera path: the 3x3/s2 stem and every depthwise
// 3x3 (stride 1 or 2, pad 1).  All are HBM-bound VALU kernels over NHWC fp32:
//   stem      : camera_encoder.py:63-67      (NCHW image in, NHWC raw out, BN stats in the epilogue)
//   depthwise : camera_encoder.py:30-35, fusion_module.py:25-27,78   (deferred BN+act on load,
//               BN stats in the epilogue); backward = transposed stencil + per-channel weight sums.
#include "kd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// stem: y[b,ho,wo,co] = sum_{ci,kh,kw} x[b,ci,2ho-1+kh,2wo-1+kw] * w[co,ci,kh,kw], Cout == 32.
// One thread = one output pixel x 32 channels (writes 128 contiguous bytes); weights in LDS.
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, float* __restrict__ partial, int B,
                                                       int Cin, int H, int W, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KK = Cin * 9;
  float* ws = sm;                      // [KK][32]  (transposed: tap-major so a tap's 32 weights are contiguous)
  float* red = sm + KK * 32;           // [256][33] stats staging
  for (int i = threadIdx.x; i < KK * 32; i += 256) ws[(i % KK) * 32 + i / KK] = w[i];
  __syncthreads();
  const int64_t npix = (int64_t)B * Ho * Wo;
  float s1 = 0.f, s2 = 0.f;            // this thread's (stat, channel) column sum, see below
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += (int64_t)gridDim.x * 256) {
    const int64_t p = base + threadIdx.x;
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
    if (p < npix) {
      const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
      for (int ci = 0; ci < Cin; ++ci) {
        const float* xp = x + ((int64_t)b * Cin + ci) * H * W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hi = 2 * ho - 1 + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int wi = 2 * wo - 1 + kw;
            float v = 0.f;
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = xp[(int64_t)hi * W + wi];
            const float* wt = ws + (ci * 9 + kh * 3 + kw) * 32;
#pragma unroll
            for (int c = 0; c < 32; ++c) acc[c] = fmaf(v, wt[c], acc[c]);
          }
        }
      }
      float* yp = y + p * 32;
#pragma unroll
      for (int c = 0; c < 32; c += 4) kd_st4(yp + c, make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]));
    }
    if (partial) {
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 32; ++c) red[threadIdx.x * 33 + c] = acc[c];   // rows past npix hold zeros
      __syncthreads();
      if (threadIdx.x < 64) {
        const int c = threadIdx.x & 31, st = threadIdx.x >> 5;
        float s = 0.f;
        for (int r = 0; r < 256; ++r) {
          const float v = red[r * 33 + c];
          s += st ? v * v : v;
        }
        if (st) s2 += s; else s1 += s;
      }
    }
  }
  if (partial && threadIdx.x < 64) {
    const int c = threadIdx.x & 31, st = threadIdx.x >> 5;
    partial[((int64_t)blockIdx.x * 2 + st) * 32 + c] = st ? s2 : s1;
  }
}

// im2col of the stem input, K padded 27 -> 32 (zeros): col[p][ci*9+kh*3+kw].  Only used by the
// stem weight gradient, which then is the ordinary TN wgrad GEMM with K = 32.
__global__ void stem_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int Cin, int H, int W,
                                   int Ho, int Wo, int Kp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * Ho * Wo * Kp;
  if (i >= n) return;
  const int k = (int)(i % Kp);
  const int64_t p = i / Kp;
  float v = 0.f;
  if (k < Cin * 9) {
    const int ci = k / 9, kh = (k % 9) / 3, kw = k % 3;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
    const int hi = 2 * ho - 1 + kh, wi = 2 * wo - 1 + kw;
    if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = x[(((int64_t)b * Cin + ci) * H + hi) * W + wi];
  }
  col[i] = v;
}

// ------------------------------------------------------------------------------------------------
// depthwise 3x3 forward.  Thread = (pixel slot, 4-channel group); grid-stride over output pixels.
struct DwArgs {
  const float* x; const float* sc; const float* sh; int act;     // deferred input (sc == null: plain)
  const float* w;                                                 // [C][9]
  float* y; float* partial;                                       // raw out [B,Ho,Wo,C]; stats slab
  int B, H, W, C, Ho, Wo, stride;
  int groups, slots;
};

__global__ __launch_bounds__(256) void dw_fwd_kernel(DwArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float wreg[4][9];
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
  }
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  if (active) {
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int wo = (int)(p % a.Wo), ho = (int)((p / a.Wo) % a.Ho), b = (int)(p / ((int64_t)a.Wo * a.Ho));
      float4 acc = kd_zero4();
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * a.stride - 1 + kh;
        if (hi < 0 || hi >= a.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wi = wo * a.stride - 1 + kw;
          if (wi < 0 || wi >= a.W) continue;
          float4 v = kd_ld4(a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0);
          if (a.sc) v = kd_affine_act4(v, sc, sh, a.act);
          const int t = kh * 3 + kw;
          acc.x = fmaf(v.x, wreg[0][t], acc.x);
          acc.y = fmaf(v.y, wreg[1][t], acc.y);
          acc.z = fmaf(v.z, wreg[2][t], acc.z);
          acc.w = fmaf(v.w, wreg[3][t], acc.w);
        }
      }
      kd_st4(a.y + p * a.C + c0, acc);
      s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
      s2.x = fmaf(acc.x, acc.x, s2.x); s2.y = fmaf(acc.y, acc.y, s2.y);
      s2.z = fmaf(acc.z, acc.z, s2.z); s2.w = fmaf(acc.w, acc.w, s2.w);
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

// ------------------------------------------------------------------------------------------------
// depthwise backward.  dyeff = kd_bwd_operand(D, Y, al, be, ga [, mask]) is the gradient w.r.t. the
// raw conv output (BN backward folded into the load).
struct DwBwdArgs {
  const float* D; const float* Y; const float* al; const float* be; const float* ga;   // [B,Ho,Wo,C]
  const float* dsc; const float* dsh; int d_act;                                         // optional mask on (D,Y)
  const float* x; const float* sc; const float* sh; int act;                             // deferred conv input
  const float* mean; const float* invstd;                                                // BN of the input (EPI2 stats)
  const float* w;
  float* gx; float* partial;           // data: G_x [B,H,W,C] (masked) + stats slab [grid][2][C]
  float* wslab;                        // weight: slab [grid][C*9]
  int B, H, W, C, Ho, Wo, stride;
  int groups, slots;
};

__device__ __forceinline__ float4 dw_dyeff(const DwBwdArgs& a, int64_t q, int c0, float4 al, float4 be, float4 ga,
                                           float4 dsc, float4 dsh) {
  const float4 d = kd_ld4(a.D + q * a.C + c0);
  if (!a.al) return d;
  const float4 y = kd_ld4(a.Y + q * a.C + c0);
  float4 r;
  r.x = kd_bwd_operand(d.x, y.x, al.x, be.x, ga.x, dsc.x, dsh.x, a.d_act);
  r.y = kd_bwd_operand(d.y, y.y, al.y, be.y, ga.y, dsc.y, dsh.y, a.d_act);
  r.z = kd_bwd_operand(d.z, y.z, al.z, be.z, ga.z, dsc.z, dsh.z, a.d_act);
  r.w = kd_bwd_operand(d.w, y.w, al.w, be.w, ga.w, dsc.w, dsh.w, a.d_act);
  return r;
}

// data gradient: gx[b,hi,wi,c] = mask(z_in) * sum_{kh,kw : (hi+1-kh) % s == 0} dyeff[b,(hi+1-kh)/s,(wi+1-kw)/s,c] * w[c,kh,kw]
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(DwBwdArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float wreg[4][9];
  float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mean = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
  }
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  const int64_t npix = (int64_t)a.B * a.H * a.W;
  if (active) {
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int wi = (int)(p % a.W), hi = (int)((p / a.W) % a.H), b = (int)(p / ((int64_t)a.W * a.H));
      float4 acc = kd_zero4();
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int th = hi + 1 - kh;
        if (th < 0 || th % a.stride) continue;
        const int ho = th / a.stride;
        if (ho >= a.Ho) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tw = wi + 1 - kw;
          if (tw < 0 || tw % a.stride) continue;
          const int wo = tw / a.stride;
          if (wo >= a.Wo) continue;
          const float4 d = dw_dyeff(a, ((int64_t)b * a.Ho + ho) * a.Wo + wo, c0, al, be, ga, dsc, dsh);
          const int t = kh * 3 + kw;
          acc.x = fmaf(d.x, wreg[0][t], acc.x);
          acc.y = fmaf(d.y, wreg[1][t], acc.y);
          acc.z = fmaf(d.z, wreg[2][t], acc.z);
          acc.w = fmaf(d.w, wreg[3][t], acc.w);
        }
      }
      if (a.sc) {                       // mask by the input's activation, collect BN-backward sums
        const float4 xr = kd_ld4(a.x + p * a.C + c0);
        acc.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
        acc.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
        acc.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
        acc.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x = fmaf(acc.x, (xr.x - mean.x) * inv.x, s2.x);
        s2.y = fmaf(acc.y, (xr.y - mean.y) * inv.y, s2.y);
        s2.z = fmaf(acc.z, (xr.z - mean.z) * inv.z, s2.z);
        s2.w = fmaf(acc.w, (xr.w - mean.w) * inv.w, s2.w);
      }
      kd_st4(a.gx + p * a.C + c0, acc);
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

// weight gradient: dw[c,kh,kw] = sum_{b,ho,wo} dyeff[b,ho,wo,c] * xact[b,ho*s-1+kh,wo*s-1+kw,c]
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(DwBwdArgs a) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
  if (active) {
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
  }
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = kd_zero4();
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  if (active) {
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int wo = (int)(p % a.Wo), ho = (int)((p / a.Wo) % a.Ho), b = (int)(p / ((int64_t)a.Wo * a.Ho));
      const float4 d = dw_dyeff(a, p, c0, al, be, ga, dsc, dsh);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * a.stride - 1 + kh;
        if (hi < 0 || hi >= a.H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wi = wo * a.stride - 1 + kw;
          if (wi < 0 || wi >= a.W) continue;
          float4 v = kd_ld4(a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0);
          if (a.sc) v = kd_affine_act4(v, sc, sh, a.act);
          const int t = kh * 3 + kw;
          acc[t].x = fmaf(d.x, v.x, acc[t].x);
          acc[t].y = fmaf(d.y, v.y, acc[t].y);
          acc[t].z = fmaf(d.z, v.z, acc[t].z);
          acc[t].w = fmaf(d.w, v.w, acc[t].w);
        }
      }
    }
  }
  for (int t = 0; t < 9; ++t) {         // reduce the row slots of this block, one tap at a time
    __syncthreads();
    kd_st4(red + tid * 4, acc[t]);
    __syncthreads();
    for (int c = tid; c < a.C; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      a.wslab[(int64_t)blockIdx.x * a.C * 9 + c * 9 + t] = s;
    }
  }
}

}  // namespace

extern "C" {

int64_t kd_stem_stat_rows(int64_t npix) {
  int64_t g = (npix + 255) / 256;
  return g < 1024 ? g : 1024;
}

int kd_stem_conv_fwd(const float* x_nchw, const float* w, float* y_nhwc, float* partial, int B, int Cin, int H, int W,
                     int Cout, void* stream) {
  KD_REQUIRE(x_nchw && w && y_nhwc && B > 0 && H > 0 && W > 0, KD_ERR_ARG, "kd_stem_conv_fwd: bad args");
  KD_REQUIRE(Cout == 32 && Cin >= 1 && Cin <= 4, KD_ERR_SHAPE, "kd_stem_conv_fwd: only Cout=32, Cin<=4 (got %d,%d)", Cout, Cin);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t npix = (int64_t)B * Ho * Wo;
  const int grid = (int)kd_stem_stat_rows(npix);
  const size_t shm = (size_t)(Cin * 9 * 32 + 256 * 33) * sizeof(float);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(256), shm, (hipStream_t)stream, x_nchw, w, y_nhwc, partial, B,
                     Cin, H, W, Ho, Wo);
  return kd_check_launch("kd_stem_conv_fwd");
}

int kd_stem_im2col(const float* x_nchw, float* col, int B, int Cin, int H, int W, int Kp, void* stream) {
  KD_REQUIRE(x_nchw && col && Kp >= Cin * 9 && Kp % 4 == 0, KD_ERR_ARG, "kd_stem_im2col: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t n = (int64_t)B * Ho * Wo * Kp;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                     col, B, Cin, H, W, Ho, Wo, Kp);
  return kd_check_launch("kd_stem_im2col");
}

int64_t kd_dwconv_stat_rows(int64_t npix, int C) { return kd_cg_layout(npix, C).grid; }

int kd_dwconv3x3_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, float* y,
                     float* partial, int B, int H, int W, int C, int stride, void* stream) {
  KD_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0, KD_ERR_ARG, "kd_dwconv3x3_fwd: bad args");
  KD_REQUIRE(C % 4 == 0 && C <= 1024 && (stride == 1 || stride == 2), KD_ERR_SHAPE, "kd_dwconv3x3_fwd: C=%d stride=%d unsupported", C, stride);
  KD_REQUIRE(kd_aligned16(x) && kd_aligned16(y), KD_ERR_ALIGN, "kd_dwconv3x3_fwd: alignment");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const KdCgLayout l = kd_cg_layout((int64_t)B * Ho * Wo, C);
  DwArgs a{x, sc, sh, act, w, y, partial, B, H, W, C, Ho, Wo, stride, l.groups, l.slots};
  hipLaunchKernelGGL(dw_fwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_dwconv3x3_fwd");
}

size_t kd_dwconv_bwd_ws_bytes(int64_t npix_out, int C) {
  return (size_t)kd_cg_layout(npix_out, C).grid * (size_t)C * 9 * sizeof(float);
}

// Backward of y = dwconv3x3(act(x*sc+sh)).  (D, Y, al, be, ga[, dsc, dsh, d_act]) describe dL/dy_raw;
// outputs: gx (masked gradient w.r.t. the activated input, with BN-backward sums in `partial`)
// and dw [C][9].  gx == null skips the data gradient, dw == null the weight gradient.
int kd_dwconv3x3_bwd(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                     const float* dsc, const float* dsh, int d_act, const float* x, const float* sc, const float* sh,
                     int act, const float* mean, const float* invstd, const float* w, float* gx, float* partial,
                     float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(D && x && w && B > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_dwconv3x3_bwd: bad args");
  KD_REQUIRE(!al || (Y && be && ga), KD_ERR_ARG, "kd_dwconv3x3_bwd: al needs Y, be, ga");
  KD_REQUIRE(!sc || (sh && (!partial || (mean && invstd))), KD_ERR_ARG, "kd_dwconv3x3_bwd: sc needs sh (+mean/invstd for stats)");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  if (gx) {
    const KdCgLayout l = kd_cg_layout((int64_t)B * H * W, C);
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, sc ? partial : nullptr,
                nullptr, B, H, W, C, Ho, Wo, stride, l.groups, l.slots};
    hipLaunchKernelGGL(dw_bwd_data_kernel, dim3(l.grid), dim3(256), 0, st, a);
    int rc = kd_check_launch("kd_dwconv3x3_bwd(data)");
    if (rc) return rc;
  }
  if (dw) {
    const KdCgLayout l = kd_cg_layout((int64_t)B * Ho * Wo, C);
    KD_REQUIRE(ws && ws_bytes >= (size_t)l.grid * C * 9 * sizeof(float), KD_ERR_WORKSPACE, "kd_dwconv3x3_bwd: workspace too small");
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, nullptr, nullptr, (float*)ws,
                B, H, W, C, Ho, Wo, stride, l.groups, l.slots};
    hipLaunchKernelGGL(dw_bwd_weight_kernel, dim3(l.grid), dim3(256), 0, st, a);
    int rc = kd_check_launch("kd_dwconv3x3_bwd(weight)");
    if (rc) return rc;
    return kd_slab_reduce_launch((const float*)ws, l.grid, (int64_t)C * 9, dw, st);
  }
  return KD_OK;
}

int64_t kd_dwconv_bwd_stat_rows(int64_t npix_in, int C) { return kd_cg_layout(npix_in, C).grid; }

}  // extern "C"
