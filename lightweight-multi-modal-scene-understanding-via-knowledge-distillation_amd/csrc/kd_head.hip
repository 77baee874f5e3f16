// kd_head.hip -- the "x4" decoder head (LightweightSegmentationHead, fusion_module.py:142-159):
// two ConvTranspose2d(k=4, s=2, p=1, bias=False)+BN+ReLU stages and a 3x3 classifier (bias).
//
// ConvTranspose2d(k4,s2,p1):  out[b, 2ih-1+kh, 2iw-1+kw, co] += x[b,ih,iw,ci] * W[ci,co,kh,kw]
//   = GEMM  col[M_in, Cout*16] = Xeff[M_in, Cin] . W.view(Cin, Cout*16)        (kd_pwconv_gemm, MFMA)
//   + col2im: every output pixel gathers its <= 2x2 contributing taps           (this file)
// backward: im2col of dy_eff -> dcol[M_in, Cout*16], then the ordinary wgrad / dgrad GEMMs.
// Column order n = co*16 + kh*4 + kw == the weight's own memory layout, so W.view(Cin, Cout*16)
// is the dgrad operand as is and its transpose is the forward operand.
#include "kd_common.h"

namespace {

struct C2iArgs {
  const float* col; float* out; float* partial;      // col [B*H*W, Cout*16]; out raw [B,2H,2W,Cout]
  int B, H, W, Cout; int groups, slots;
};

__global__ __launch_bounds__(256) void col2im_fwd_kernel(C2iArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t npix = (int64_t)a.B * Ho * Wo;
  const int N = a.Cout * 16;
  if (active) {
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int ow = (int)(p % Wo), oh = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
      const int kha = (oh + 1) & 1, iha = (oh + 1 - kha) >> 1, ihb = iha - 1;     // taps (kha, iha) and (kha+2, ihb)
      const int kwa = (ow + 1) & 1, iwa = (ow + 1 - kwa) >> 1, iwb = iwa - 1;
      const bool ha = iha < a.H, hb = ihb >= 0, wa = iwa < a.W, wb = iwb >= 0;
      const int ihac = ha ? iha : a.H - 1, ihbc = hb ? ihb : 0, iwac = wa ? iwa : a.W - 1, iwbc = wb ? iwb : 0;
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      const int64_t base = (int64_t)b * a.H * a.W;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* cj = a.col + (c0 + j) * 16;
        const float vaa = cj[(base + (int64_t)ihac * a.W + iwac) * N + kha * 4 + kwa];
        const float vab = cj[(base + (int64_t)ihac * a.W + iwbc) * N + kha * 4 + kwa + 2];
        const float vba = cj[(base + (int64_t)ihbc * a.W + iwac) * N + (kha + 2) * 4 + kwa];
        const float vbb = cj[(base + (int64_t)ihbc * a.W + iwbc) * N + (kha + 2) * 4 + kwa + 2];
        acc[j] = ((ha && wa) ? vaa : 0.f) + ((ha && wb) ? vab : 0.f) + ((hb && wa) ? vba : 0.f) + ((hb && wb) ? vbb : 0.f);
      }
      const float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
      kd_st4(a.out + p * a.Cout + c0, v);
      s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
      s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
    }
  }
  if (a.partial) {
    kd_st4(red + tid * 4, s1);
    kd_st4(red + 1024 + tid * 4, s2);
    __syncthreads();
    for (int i = tid; i < 2 * a.Cout; i += 256) {
      const int st = i / a.Cout, c = i % a.Cout;
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[st * 1024 + (sl * a.groups + c / 4) * 4 + (c & 3)];
      a.partial[((int64_t)blockIdx.x * 2 + st) * a.Cout + c] = s;
    }
  }
}

// dcol[(b,ih,iw), co*16 + kh*4 + kw] = dyeff[b, 2ih-1+kh, 2iw-1+kw, co]   (0 outside the output image)
__global__ __launch_bounds__(256) void convT_im2col_bwd_kernel(const float* __restrict__ D, const float* __restrict__ Y,
                                                               const float* __restrict__ al, const float* __restrict__ be,
                                                               const float* __restrict__ ga, const float* __restrict__ msc,
                                                               const float* __restrict__ msh, int act, float* __restrict__ dcol,
                                                               int B, int H, int W, int Cout) {
  const int64_t n = (int64_t)B * H * W * Cout;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % Cout);
    const int64_t m = i / Cout;
    const int iw = (int)(m % W), ih = (int)((m / W) % H), b = (int)(m / ((int64_t)W * H));
    const float a_ = al[co], b_ = be[co], g_ = ga[co], sc = msc ? msc[co] : 0.f, sh = msh ? msh[co] : 0.f;
    float v[16];
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const int oh = 2 * ih - 1 + kh;
      const bool hok = oh >= 0 && oh < 2 * H;
      const int ohc = oh < 0 ? 0 : (oh >= 2 * H ? 2 * H - 1 : oh);
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        const int ow = 2 * iw - 1 + kw;
        const bool ok = hok && ow >= 0 && ow < 2 * W;
        const int owc = ow < 0 ? 0 : (ow >= 2 * W ? 2 * W - 1 : ow);
        const int64_t q = (((int64_t)b * 2 * H + ohc) * 2 * W + owc) * Cout + co;
        const float e = kd_bwd_operand(D[q], Y[q], a_, b_, g_, sc, sh, act);
        v[kh * 4 + kw] = ok ? e : 0.f;
      }
    }
    float* dst = dcol + (m * Cout + co) * 16;
#pragma unroll
    for (int t = 0; t < 16; t += 4) kd_st4(dst + t, make_float4(v[t], v[t + 1], v[t + 2], v[t + 3]));
  }
}

// ---- 3x3 classifier, Cin (<= 32, multiple of 4) -> NC (<= 4), pad 1, bias; deferred input; NCHW logits
struct C3Args {
  const float* x; const float* sc; const float* sh; int act; const float* w; const float* b; float* logits;
  const float* dlog; const float* mean; const float* invstd; float* gx; float* partial; float* wslab;
  int B, H, W, Cin, NC; int groups, slots;
};

__global__ __launch_bounds__(256) void cls3x3_fwd_kernel(C3Args a) {
  __shared__ float ws[4 * 32 * 9];
  for (int i = threadIdx.x; i < a.NC * a.Cin * 9; i += 256) ws[i] = a.w[i];       // [j][c][t]
  __syncthreads();
  const int64_t npix = (int64_t)a.B * a.H * a.W;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int w_ = (int)(p % a.W), h_ = (int)((p / a.W) % a.H), b = (int)(p / ((int64_t)a.W * a.H));
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = h_ - 1 + kh;
    if (hi < 0 || hi >= a.H) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = w_ - 1 + kw;
      if (wi < 0 || wi >= a.W) continue;
      const float* xp = a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.Cin;
      for (int c = 0; c < a.Cin; c += 4) {
        float4 v = kd_ld4(xp + c);
        if (a.sc) v = kd_affine_act4(v, kd_ld4(a.sc + c), kd_ld4(a.sh + c), a.act);
        const int t = kh * 3 + kw;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (j < a.NC) {
            const float* wj = ws + (j * a.Cin + c) * 9 + t;
            acc[j] = fmaf(v.x, wj[0], fmaf(v.y, wj[9], fmaf(v.z, wj[18], fmaf(v.w, wj[27], acc[j]))));
          }
      }
    }
  }
  const int HW = a.H * a.W;
  const int64_t hw = p % HW;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (j < a.NC) a.logits[((int64_t)b * a.NC + j) * HW + hw] = acc[j] + (a.b ? a.b[j] : 0.f);
}

// data gradient: gx[p, c] = mask * sum_{j,t} dl[j, p - off(t)] * w[j][c][t]  (+ BN-backward sums)
__global__ __launch_bounds__(256) void cls3x3_bwd_data_kernel(C3Args a) {
  __shared__ float red[2 * 256 * 4];
  __shared__ float ws[4 * 32 * 9];
  for (int i = threadIdx.x; i < a.NC * a.Cin * 9; i += 256) ws[i] = a.w[i];
  __syncthreads();
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  const int HW = a.H * a.W;
  const int64_t npix = (int64_t)a.B * HW;
  if (active) {
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mu = kd_zero4(), inv = kd_zero4();
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mu = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
      const int w_ = (int)(p % a.W), h_ = (int)((p / a.W) % a.H), b = (int)(p / HW);
      float4 g = kd_zero4();
      for (int kh = 0; kh < 3; ++kh) {
        const int ho = h_ + 1 - kh;                 // output pixel that used this input at tap kh
        if (ho < 0 || ho >= a.H) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int wo = w_ + 1 - kw;
          if (wo < 0 || wo >= a.W) continue;
          const int t = kh * 3 + kw;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < a.NC) {
              const float d = a.dlog[((int64_t)b * a.NC + j) * HW + (int64_t)ho * a.W + wo];
              const float* wj = ws + (j * a.Cin + c0) * 9 + t;
              g.x = fmaf(d, wj[0], g.x); g.y = fmaf(d, wj[9], g.y); g.z = fmaf(d, wj[18], g.z); g.w = fmaf(d, wj[27], g.w);
            }
        }
      }
      if (a.sc) {
        const float4 xr = kd_ld4(a.x + p * a.Cin + c0);
        g.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act); g.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
        g.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act); g.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x = fmaf(g.x, (xr.x - mu.x) * inv.x, s2.x); s2.y = fmaf(g.y, (xr.y - mu.y) * inv.y, s2.y);
        s2.z = fmaf(g.z, (xr.z - mu.z) * inv.z, s2.z); s2.w = fmaf(g.w, (xr.w - mu.w) * inv.w, s2.w);
      }
      kd_st4(a.gx + p * a.Cin + c0, g);
    }
  }
  if (a.partial) {
    kd_st4(red + tid * 4, s1);
    kd_st4(red + 1024 + tid * 4, s2);
    __syncthreads();
    for (int i = tid; i < 2 * a.Cin; i += 256) {
      const int st = i / a.Cin, c = i % a.Cin;
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[st * 1024 + (sl * a.groups + c / 4) * 4 + (c & 3)];
      a.partial[((int64_t)blockIdx.x * 2 + st) * a.Cin + c] = s;
    }
  }
}

// weight gradient: dw[j][c][t] = sum_p dl[j, p] * xact[p + off(t), c];  db[j] = sum_p dl[j, p]
// one pass per (j, t): thread-private float4 over its 4 channels, block reduction, slab [grid][NC*Cin*9 + 4]
__global__ __launch_bounds__(256) void cls3x3_bwd_weight_kernel(C3Args a) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int c0 = gidx * 4;
  const int HW = a.H * a.W;
  const int64_t npix = (int64_t)a.B * HW;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
  if (active && a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
  float* out = a.wslab + (int64_t)blockIdx.x * (a.NC * a.Cin * 9 + 4);
  for (int j = 0; j < a.NC; ++j) {
    for (int t = 0; t < 10; ++t) {                   // t == 9: the bias row (sum of dl)
      float4 acc = kd_zero4();
      if (active) {
        const int kh = t / 3, kw = t % 3;
        for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
          const float d = a.dlog[((int64_t)(p / HW) * a.NC + j) * HW + p % HW];
          if (t == 9) { if (gidx == 0) acc.x += d; continue; }
          const int w_ = (int)(p % a.W), h_ = (int)((p / a.W) % a.H), b = (int)(p / HW);
          const int hi = h_ - 1 + kh, wi = w_ - 1 + kw;
          if (hi < 0 || hi >= a.H || wi < 0 || wi >= a.W) continue;
          float4 v = kd_ld4(a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.Cin + c0);
          if (a.sc) v = kd_affine_act4(v, sc, sh, a.act);
          acc.x = fmaf(d, v.x, acc.x); acc.y = fmaf(d, v.y, acc.y); acc.z = fmaf(d, v.z, acc.z); acc.w = fmaf(d, v.w, acc.w);
        }
      }
      __syncthreads();
      kd_st4(red + tid * 4, acc);
      __syncthreads();
      if (t < 9) {
        for (int c = tid; c < a.Cin; c += 256) {
          float s = 0.f;
          for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
          out[(j * a.Cin + c) * 9 + t] = s;
        }
      } else if (tid == 0) {
        float s = 0.f;
        for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups) * 4];
        out[a.NC * a.Cin * 9 + j] = s;
      }
    }
  }
}

}  // namespace

extern "C" {

int64_t kd_deconv_stat_rows(int64_t npix_out, int Cout) { return kd_cg_layout(npix_out, Cout).grid; }

// out_raw[B,2H,2W,Cout] = col2im(col[B*H*W, Cout*16]) for ConvTranspose2d(k=4, s=2, p=1); BN stats in `partial`.
int kd_deconv4x4s2_col2im_fwd(const float* col, float* out, float* partial, int B, int H, int W, int Cout, void* stream) {
  KD_REQUIRE(col && out && B > 0 && H > 0 && W > 0 && Cout % 4 == 0 && Cout <= 1024, KD_ERR_ARG, "kd_deconv4x4s2_col2im_fwd: bad args");
  const KdCgLayout l = kd_cg_layout((int64_t)B * 4 * H * W, Cout);
  C2iArgs a{col, out, partial, B, H, W, Cout, l.groups, l.slots};
  hipLaunchKernelGGL(col2im_fwd_kernel, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_deconv4x4s2_col2im_fwd");
}

// dcol[B*H*W, Cout*16] from dy_eff = al*(D*mask) + be*Y + ga at the 2H x 2W output resolution.
int kd_deconv4x4s2_im2col_bwd(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                             const float* msc, const float* msh, int act, float* dcol, int B, int H, int W, int Cout,
                             void* stream) {
  KD_REQUIRE(D && Y && al && be && ga && dcol && B > 0 && Cout > 0, KD_ERR_ARG, "kd_deconv4x4s2_im2col_bwd: bad args");
  KD_REQUIRE(act == KD_ACT_NONE || (msc && msh), KD_ERR_ARG, "kd_deconv4x4s2_im2col_bwd: mask needs sc/sh");
  const int64_t n = (int64_t)B * H * W * Cout;
  int64_t grid = (n + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(convT_im2col_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, D, Y, al, be, ga,
                     msc, msh, act, dcol, B, H, W, Cout);
  return kd_check_launch("kd_deconv4x4s2_im2col_bwd");
}

int kd_cls3x3_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, const float* b,
                  float* logits_nchw, int B, int H, int W, int Cin, int NC, void* stream) {
  KD_REQUIRE(x && w && logits_nchw && B > 0, KD_ERR_ARG, "kd_cls3x3_fwd: bad args");
  KD_REQUIRE(Cin % 4 == 0 && Cin <= 32 && NC >= 1 && NC <= 4, KD_ERR_SHAPE, "kd_cls3x3_fwd: Cin=%d NC=%d unsupported", Cin, NC);
  C3Args a{x, sc, sh, act, w, b, logits_nchw, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, NC, 0, 0};
  const int64_t npix = (int64_t)B * H * W;
  hipLaunchKernelGGL(cls3x3_fwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_cls3x3_fwd");
}

int64_t kd_cls3x3_bwd_stat_rows(int64_t npix, int Cin) { return kd_cg_layout(npix, Cin, 1024).grid; }
size_t kd_cls3x3_bwd_ws_bytes(int64_t npix, int Cin, int NC) {
  return (size_t)kd_cg_layout(npix, Cin, 256).grid * (size_t)(NC * Cin * 9 + 4) * sizeof(float);
}

// gx[M,Cin] (masked, BN-backward sums in `partial`), dwb = dW [NC*Cin*9] | db [NC] (padded to 4)
int kd_cls3x3_bwd(const float* dlog_nchw, const float* x, const float* sc, const float* sh, int act, const float* mean,
                  const float* invstd, const float* w, float* gx, float* partial, float* dwb, int B, int H, int W,
                  int Cin, int NC, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(dlog_nchw && x && w && gx && dwb && ws && B > 0, KD_ERR_ARG, "kd_cls3x3_bwd: bad args");
  KD_REQUIRE(Cin % 4 == 0 && Cin <= 32 && NC >= 1 && NC <= 4, KD_ERR_SHAPE, "kd_cls3x3_bwd: Cin=%d NC=%d unsupported", Cin, NC);
  const int64_t npix = (int64_t)B * H * W;
  hipStream_t st = (hipStream_t)stream;
  {
    const KdCgLayout l = kd_cg_layout(npix, Cin, 1024);
    C3Args a{x, sc, sh, act, w, nullptr, nullptr, dlog_nchw, mean, invstd, gx, sc ? partial : nullptr, nullptr, B, H, W, Cin, NC, l.groups, l.slots};
    hipLaunchKernelGGL(cls3x3_bwd_data_kernel, dim3(l.grid), dim3(256), 0, st, a);
  }
  const KdCgLayout l = kd_cg_layout(npix, Cin, 256);
  const int per = NC * Cin * 9 + 4;
  KD_REQUIRE(ws_bytes >= (size_t)l.grid * per * sizeof(float), KD_ERR_WORKSPACE, "kd_cls3x3_bwd: workspace too small");
  hipError_t e = hipMemsetAsync(ws, 0, (size_t)l.grid * per * sizeof(float), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_cls3x3_bwd: memset failed");
  C3Args a{x, sc, sh, act, w, nullptr, nullptr, dlog_nchw, nullptr, nullptr, nullptr, nullptr, (float*)ws, B, H, W, Cin, NC, l.groups, l.slots};
  hipLaunchKernelGGL(cls3x3_bwd_weight_kernel, dim3(l.grid), dim3(256), 0, st, a);
  int rc = kd_check_launch("kd_cls3x3_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, l.grid, per, dwb, st);
}

}  // extern "C"
