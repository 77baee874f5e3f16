// kd_conv.hip -- the spatial convolutions of the camera path: the 3x3/s2 stem and every depthwise
// 3x3 (stride 1 or 2, pad 1).  All are HBM-bound VALU kernels over NHWC fp32:
//   stem      : camera_encoder.py:63-67      (NCHW image in, NHWC raw out, BN stats in the epilogue)
//   depthwise : camera_encoder.py:30-35, fusion_module.py:25-27,78   (deferred BN+act on load,
//               BN stats in the epilogue); backward = transposed stencil + per-channel weight sums.
#include "kd_common.h"
#include <type_traits>

#include <atomic>
#include <cstdlib>

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

// Round 3 form of the same convolution (same fma chain per output channel, bit-identical raw output): the 9 * CIN input values
// of a pixel are fetched first (independent loads), then every output channel is one fma chain whose weights are wave-uniform
// constants-offset reads of `w` -- scalar loads into SGPRs (s_load_dwordx*), no LDS and no per-fma ds_read; BatchNorm sums are
// kept per thread in registers over all of the thread's pixels and reduced once at the end of the kernel (the first form
// staged every 256-pixel batch through LDS and let 64 threads add 256 values each).
// timing-only probes of dev builds (-DKD_STEM_PROBE=bits; results WRONG by construction): 1 no stores, 2 no input loads, 4 a quarter of the channels
#ifndef KD_STEM_PROBE
#define KD_STEM_PROBE 0
#endif
// FIN (round 4, inference: eval mode without autograd -- the frozen KD teacher, validation): the stem's BatchNorm + activation
// finish, act(fma(raw, sc[c], sh[c])), applied here instead of by a kd_bn_act_apply pass over the [B, 32, H/2, W/2] map (the same
// operations in the same order: identical bits; 0.23 ms per step at 256 frames).  sc / sh are wave-uniform scalar loads.
template <int CIN, bool STATS, bool FIN = false>
__global__ __launch_bounds__(256) void stem_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, float* __restrict__ partial, int B, int H, int W,
                                                        int Ho, int Wo, const float* __restrict__ fsc = nullptr,
                                                        const float* __restrict__ fsh = nullptr, int fact = 0) {
  constexpr int KK = CIN * 9;
  // A thread owns one pixel = 128 contiguous output bytes; stored straight from its registers that is eight 16-byte stores whose
  // 64 lanes hit 64 different lines each (probe: 357 us with the stores, 72 us without: 1.9 TB/s).  So a wave's 64 x 32 results
  // go through a wave-private LDS tile (rows padded to 36 floats: aligned 16-byte accesses in both directions) and leave as eight
  // fully coalesced 1 KB stores: lane l writes bytes [16 l, 16 l + 16) of each KB.  The tile doubles as the statistics staging.
  constexpr int TLD = 36;
  __shared__ __attribute__((aligned(16))) float red[4 * 64 * TLD];
  static_assert(4 * 64 * TLD >= 256 * 33, "the end-of-kernel statistics reduction reuses the tile");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = red + wave * 64 * TLD;
  const int64_t npix = (int64_t)B * Ho * Wo;
  float s1[STATS ? 32 : 1], s2[STATS ? 32 : 1];
#pragma unroll
  for (int c = 0; c < (STATS ? 32 : 1); ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += (int64_t)gridDim.x * 256) {
    const int64_t p_raw = base + threadIdx.x;
    const bool pok = p_raw < npix;
    const int64_t p = pok ? p_raw : npix - 1;                  // (tail lanes recompute the last pixel; nothing of it is stored or summed)
    {
      const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
      float v[KK];
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float* xp = x + ((int64_t)b * CIN + ci) * H * W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hi = 2 * ho - 1 + kh;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int wi = 2 * wo - 1 + kw;
            const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
            const float t = (KD_STEM_PROBE & 2) ? (float)(hi + wi) : xp[(int64_t)(ok ? hi : 0) * W + (ok ? wi : 0)];
            v[ci * 9 + kh * 3 + kw] = ok ? t : 0.f;
          }
        }
      }
#pragma unroll
      for (int c4 = 0; c4 < ((KD_STEM_PROBE & 4) ? 8 : 32); c4 += 4) {
        float a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < KK; ++t) acc = fmaf(v[t], w[(c4 + j) * KK + t], acc);
          a[j] = FIN ? kd_act(kd_affine(acc, fsc[c4 + j], fsh[c4 + j]), fact) : acc;
          if (STATS) { const float q = pok ? acc : 0.f; s1[c4 + j] += q; s2[c4 + j] = fmaf(q, q, s2[c4 + j]); }
        }
        kd_st4(tile + lane * TLD + c4, make_float4(a[0], a[1], a[2], a[3]));
      }
    }
    __builtin_amdgcn_wave_barrier();                            // (LDS is in-order per wave; this only pins the compiler's order)
    const int64_t wbase = base + wave * 64;                     // first pixel of this wave's 64
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pix = 8 * k + (lane >> 3), col = (lane & 7) * 4;
      const float4 o = kd_ld4(tile + pix * TLD + col);
      if (!(KD_STEM_PROBE & 1) && wbase + pix < npix) kd_st4(y + (wbase + pix) * 32 + col, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (STATS) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 32; ++c) red[threadIdx.x * 33 + c] = st ? s2[c] : s1[c];
      __syncthreads();
      const int c = threadIdx.x & 31, seg = threadIdx.x >> 5;            // 8 segments of 32 threads' sums per channel
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += red[(seg * 32 + r) * 33 + c];
      __syncthreads();
      red[seg * 33 + c] = s;
      __syncthreads();
      if (threadIdx.x < 32) {
        float tot = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) tot += red[g8 * 33 + threadIdx.x];
        partial[((int64_t)blockIdx.x * 2 + st) * 32 + threadIdx.x] = tot;
      }
    }
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

// Round 3 form for Kp == 32: one thread gathers the 9 * CIN inputs of a pixel (instead of one thread per matrix element, whose
// index arithmetic cost more than its load) and the wave leaves its 64 x 32 block through the same wave-private LDS tile as
// stem_fwd2_kernel: eight coalesced 1 KB stores.
template <int CIN>
__global__ __launch_bounds__(256) void stem_im2col2_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int Ho, int Wo) {
  constexpr int KK = CIN * 9, TLD = 36;
  static_assert(KK <= 32, "K is padded to 32");
  __shared__ __attribute__((aligned(16))) float tiles[4 * 64 * TLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* tile = tiles + wave * 64 * TLD;
  const int64_t npix = (int64_t)B * Ho * Wo;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += (int64_t)gridDim.x * 256) {
    const int64_t p_raw = base + threadIdx.x;
    const int64_t p = p_raw < npix ? p_raw : npix - 1;
    const int wo = (int)(p % Wo), ho = (int)((p / Wo) % Ho), b = (int)(p / ((int64_t)Wo * Ho));
    float v[32];
#pragma unroll
    for (int t = KK; t < 32; ++t) v[t] = 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float* xp = x + ((int64_t)b * CIN + ci) * H * W;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hi = 2 * ho - 1 + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int wi = 2 * wo - 1 + kw;
          const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
          const float t = xp[(int64_t)(ok ? hi : 0) * W + (ok ? wi : 0)];
          v[ci * 9 + kh * 3 + kw] = ok ? t : 0.f;
        }
      }
    }
#pragma unroll
    for (int c4 = 0; c4 < 32; c4 += 4) kd_st4(tile + lane * TLD + c4, make_float4(v[c4], v[c4 + 1], v[c4 + 2], v[c4 + 3]));
    __builtin_amdgcn_wave_barrier();
    const int64_t wbase = base + wave * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pix = 8 * k + (lane >> 3), c = (lane & 7) * 4;
      const float4 o = kd_ld4(tile + pix * TLD + c);
      if (wbase + pix < npix) kd_st4(col + (wbase + pix) * 32 + c, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// depthwise 3x3: argument blocks (kernels: the sliding-window family further down).
struct DwArgs {
  const float* x; const float* sc; const float* sh; int act;     // deferred input (sc == null: plain)
  const float* w;                                                 // [C][9]
  float* y; float* partial;                                       // raw out [B,Ho,Wo,C]; stats slab
  int B, H, W, C, Ho, Wo, stride;
  int groups, slots, nchunk;     // channel quads per chunk, adjacent columns per block, channel chunks (dw_layout)
  int nt;                                                         // large output: non-temporal stores
};

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
  const float* addend;                 // optional [B,H,W,C]: a second gradient path into the same input (the block's residual),
                                       // added to the data gradient BEFORE the activation mask (fused stride-1 column walk only)
  int B, H, W, C, Ho, Wo, stride;
  int groups, slots, nchunk;     // channel quads per chunk, adjacent columns per block, channel chunks (dw_layout)
  int nt;                              // large gx: non-temporal stores
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

// ------------------------------------------------------------------------------------------------
// Sliding-window depthwise kernels.  A work item is a column segment (b, 16-row segment, w); a thread
// walks down the segment keeping the 3x3 window of ACTIVATED inputs in registers and loading one new
// input row (3 float4) per output row -- 3 loads and 3 activations per output instead of 9 (and 5
// instead of 11 in the weight gradient, 6 instead of 18 in the stride-1 data gradient).
constexpr int DW_SEG = 16;

struct DwRow { float4 l, c, r; };

// activated input row (hi, wi-1..wi+1) of image b, zero outside the image (branch-free)
__device__ __forceinline__ DwRow dw_load_row(const float* __restrict__ x, bool deferred, float4 sc, float4 sh, int act,
                                             int b, int hi, int wi, int H, int W, int C, int c0) {
  const bool hok = hi >= 0 && hi < H;
  const int hic = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
  const float* rowp = x + ((int64_t)b * H + hic) * W * C + c0;
  const bool lok = hok && wi - 1 >= 0, cok = hok, rok = hok && wi + 1 < W;
  const int wl = wi - 1 < 0 ? 0 : wi - 1, wr_ = wi + 1 >= W ? W - 1 : wi + 1;
  float4 l = kd_ld4(rowp + (int64_t)wl * C), c = kd_ld4(rowp + (int64_t)wi * C), r = kd_ld4(rowp + (int64_t)wr_ * C);
  if (deferred) { l = kd_affine_act4(l, sc, sh, act); c = kd_affine_act4(c, sc, sh, act); r = kd_affine_act4(r, sc, sh, act); }
  DwRow o;
  o.l = make_float4(lok ? l.x : 0.f, lok ? l.y : 0.f, lok ? l.z : 0.f, lok ? l.w : 0.f);
  o.c = make_float4(cok ? c.x : 0.f, cok ? c.y : 0.f, cok ? c.z : 0.f, cok ? c.w : 0.f);
  o.r = make_float4(rok ? r.x : 0.f, rok ? r.y : 0.f, rok ? r.z : 0.f, rok ? r.w : 0.f);
  return o;
}

// the same row in two steps, so that the raw loads of the NEXT row are in flight under the arithmetic of this one
__device__ __forceinline__ DwRow dw_load_row_raw(const float* __restrict__ x, int b, int hi, int wi, int H, int W, int C, int c0) {
  const int hic = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
  const float* rowp = x + ((int64_t)b * H + hic) * W * C + c0;
  const int wl = wi - 1 < 0 ? 0 : wi - 1, wr_ = wi + 1 >= W ? W - 1 : wi + 1;
  DwRow o;
  o.l = kd_ld4(rowp + (int64_t)wl * C); o.c = kd_ld4(rowp + (int64_t)wi * C); o.r = kd_ld4(rowp + (int64_t)wr_ * C);
  return o;
}
__device__ __forceinline__ DwRow dw_finish_row(DwRow r, bool deferred, float4 sc, float4 sh, int act, int hi, int wi, int H, int W) {
  const bool hok = hi >= 0 && hi < H;
  const bool lok = hok && wi - 1 >= 0, cok = hok, rok = hok && wi + 1 < W;
  if (deferred) { r.l = kd_affine_act4(r.l, sc, sh, act); r.c = kd_affine_act4(r.c, sc, sh, act); r.r = kd_affine_act4(r.r, sc, sh, act); }
  DwRow o;
  o.l = make_float4(lok ? r.l.x : 0.f, lok ? r.l.y : 0.f, lok ? r.l.z : 0.f, lok ? r.l.w : 0.f);
  o.c = make_float4(cok ? r.c.x : 0.f, cok ? r.c.y : 0.f, cok ? r.c.z : 0.f, cok ? r.c.w : 0.f);
  o.r = make_float4(rok ? r.r.x : 0.f, rok ? r.r.y : 0.f, rok ? r.r.z : 0.f, rok ? r.r.w : 0.f);
  return o;
}

__device__ __forceinline__ void dw_fma_row(float4& acc, const DwRow& r, const float (*w)[9], int kh) {
  acc.x = fmaf(r.l.x, w[0][kh * 3], fmaf(r.c.x, w[0][kh * 3 + 1], fmaf(r.r.x, w[0][kh * 3 + 2], acc.x)));
  acc.y = fmaf(r.l.y, w[1][kh * 3], fmaf(r.c.y, w[1][kh * 3 + 1], fmaf(r.r.y, w[1][kh * 3 + 2], acc.y)));
  acc.z = fmaf(r.l.z, w[2][kh * 3], fmaf(r.c.z, w[2][kh * 3 + 1], fmaf(r.r.z, w[2][kh * 3 + 2], acc.z)));
  acc.w = fmaf(r.l.w, w[3][kh * 3], fmaf(r.c.w, w[3][kh * 3 + 1], fmaf(r.r.w, w[3][kh * 3 + 2], acc.w)));
}

// one slab row per block ROW bx; a block owns the CW = 4 * groups channels from cbase (all of them when nchunk == 1)
__device__ __forceinline__ void dw_block_stats(float* red, float4 s1, float4 s2, float* partial, int C, int groups, int slots, int bx,
                                               int cbase) {
  const int tid = threadIdx.x;
  kd_st4(red + tid * 4, s1);
  kd_st4(red + 1024 + tid * 4, s2);
  __syncthreads();
  const int CW = groups * 4;
  for (int i = tid; i < 2 * CW; i += 256) {
    const int st = i / CW, c = i % CW;
    float s = 0.f;
    for (int sl = 0; sl < slots; ++sl) s += red[st * 1024 + (sl * groups + c / 4) * 4 + (c & 3)];
    partial[((int64_t)bx * 2 + st) * C + cbase + c] = s;
  }
}

template <int STRIDE>
__global__ __launch_bounds__(256) void dw_fwd_sw_kernel(DwArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float wreg[4][9];
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
    const bool deferred = a.sc != nullptr;
    if (deferred) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    const int nseg = (a.Ho + DW_SEG - 1) / DW_SEG;
    const int64_t items = (int64_t)a.B * nseg * a.Wo;
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int wo = (int)(it % a.Wo), sg = (int)((it / a.Wo) % nseg), b = (int)(it / ((int64_t)a.Wo * nseg));
      const int h0 = sg * DW_SEG, h1 = h0 + DW_SEG < a.Ho ? h0 + DW_SEG : a.Ho;
      const int wi = wo * STRIDE;
      DwRow r0 = dw_load_row(a.x, deferred, sc, sh, a.act, b, h0 * STRIDE - 1, wi, a.H, a.W, a.C, c0), r1, r2;
      if (STRIDE == 1) r1 = dw_load_row(a.x, deferred, sc, sh, a.act, b, h0, wi, a.H, a.W, a.C, c0);
      for (int ho = h0; ho < h1; ++ho) {         // (a one-row-ahead prefetch, as in the backward kernels, measured no gain here)
        if (STRIDE == 2) r1 = dw_load_row(a.x, deferred, sc, sh, a.act, b, 2 * ho, wi, a.H, a.W, a.C, c0);
        r2 = dw_load_row(a.x, deferred, sc, sh, a.act, b, ho * STRIDE + 1, wi, a.H, a.W, a.C, c0);
        float4 acc = kd_zero4();
        dw_fma_row(acc, r0, wreg, 0);
        dw_fma_row(acc, r1, wreg, 1);
        dw_fma_row(acc, r2, wreg, 2);
        if (a.nt) kd_st4_nt(a.y + (((int64_t)b * a.Ho + ho) * a.Wo + wo) * a.C + c0, acc);
        else kd_st4(a.y + (((int64_t)b * a.Ho + ho) * a.Wo + wo) * a.C + c0, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2.x = fmaf(acc.x, acc.x, s2.x); s2.y = fmaf(acc.y, acc.y, s2.y);
        s2.z = fmaf(acc.z, acc.z, s2.z); s2.w = fmaf(acc.w, acc.w, s2.w);
        if (STRIDE == 1) { r0 = r1; r1 = r2; } else { r0 = r2; }
      }
    }
  }
  if (a.partial) dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase);
}

// Stride 1, round 4: the same window arithmetic as ONE software pipeline over all the column segments of a thread (the form that took the
// bf16 depthwise kernel from 3.3 to 4.4 TB/s, kd_bf16.hip).  Above, every segment starts cold -- two rows are loaded and waited for before
// the first output row -- and every output row waits for the row it has just requested; 16 waves per CU cover that to ~5.4 TB/s.  Here a
// segment is SEG + 2 row arrivals, the raw row two arrivals ahead is always in flight (during the last two arrivals of a segment: rows
// 0 and 1 of the thread's next segment), and the arrival loop is fully unrolled: no control flow between the loads, exact vmcnt waits,
// the window rotates by renaming.  Same fma order per output as dw_fwd_sw_kernel<1>: the same bits.  Ho must be a multiple of SEG.
template <int STRIDE, int SEG>
__global__ __launch_bounds__(256) void dw_fwd_pipe_kernel(DwArgs a) {
  constexpr int NA = STRIDE * SEG + (STRIDE == 1 ? 2 : 1);     // row arrivals of a segment: input rows STRIDE * h0 - 1 ... STRIDE * (h0 + SEG - 1) + 1
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  const int nseg = a.Ho / SEG;
  const int64_t items = (int64_t)a.B * nseg * a.Wo, stride = (int64_t)nbx * a.slots;
  int64_t it = (int64_t)bx * a.slots + slot;
  if (slot < a.slots && it < items) {
    float wreg[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
    const bool deferred = a.sc != nullptr;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
    if (deferred) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    auto coords = [&](int64_t t, int& b, int& h0, int& wo) __attribute__((always_inline)) {
      wo = (int)(t % a.Wo);
      const int64_t q = t / a.Wo;
      h0 = (int)(q % nseg) * SEG;
      b = (int)(q / nseg);
    };
    // a raw row by 32-bit element offsets from a.x (the host checks the tensor has fewer than 2^31 elements); `hi` passes through an
    // opaque asm so that hipcc computes each arrival's addresses when it issues them (left alone it precomputes all 18 x 3 pointers of a
    // segment: 218 registers, two waves per SIMD)
    auto issue = [&](int bb, int hi, int wi) __attribute__((always_inline)) -> DwRow {
      asm volatile("" : "+v"(hi));
      const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi);
      const uint32_t ro = (uint32_t)((bb * a.H + hic) * a.W) * (uint32_t)a.C + (uint32_t)c0;
      const int wl = wi - 1 < 0 ? 0 : wi - 1, wr_ = wi + 1 >= a.W ? a.W - 1 : wi + 1;
      DwRow o;
      o.l = kd_ld4(a.x + (ro + (uint32_t)(wl * a.C))); o.c = kd_ld4(a.x + (ro + (uint32_t)(wi * a.C))); o.r = kd_ld4(a.x + (ro + (uint32_t)(wr_ * a.C)));
      return o;
    };
    int b, h0, wo;
    coords(it, b, h0, wo);
    DwRow qa = issue(b, STRIDE * h0 - 1, wo * STRIDE), qb = issue(b, STRIDE * h0, wo * STRIDE);
    DwRow r0, r1, r2;
    r1.l = r1.c = r1.r = kd_zero4();
    r2.l = r2.c = r2.r = kd_zero4();
    for (;;) {
      const int64_t itn = it + stride;
      const bool more = itn < items;
      int bn, h0n, won;
      coords(more ? itn : it, bn, h0n, won);
#pragma unroll
      for (int k = 0; k < NA; ++k) {                           // arrival k is input row STRIDE * h0 - 1 + k
        const DwRow c = qa;
        qa = qb;
        if (k + 2 < NA) qb = issue(b, STRIDE * h0 + 1 + k, wo * STRIDE);
        else qb = issue(bn, STRIDE * h0n - 1 + (k + 2 - NA), won * STRIDE);                  // rows 0, 1 of the next segment
        r0 = r1; r1 = r2;
        r2 = dw_finish_row(c, deferred, sc, sh, a.act, STRIDE * h0 - 1 + k, wo * STRIDE, a.H, a.W);
        if (k >= 2 && (STRIDE == 1 || k % 2 == 0)) {           // stride 2: output row h0 + k / 2 - 1 is complete with its third row
          float4 acc = kd_zero4();
          dw_fma_row(acc, r0, wreg, 0);
          dw_fma_row(acc, r1, wreg, 1);
          dw_fma_row(acc, r2, wreg, 2);
          float* dst = a.y + (((int64_t)b * a.Ho + (STRIDE == 1 ? h0 + k - 2 : h0 + k / 2 - 1)) * a.Wo + wo) * a.C + c0;
          if (a.nt) kd_st4_nt(dst, acc); else kd_st4(dst, acc);
          s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
          s2.x = fmaf(acc.x, acc.x, s2.x); s2.y = fmaf(acc.y, acc.y, s2.y);
          s2.z = fmaf(acc.z, acc.z, s2.z); s2.w = fmaf(acc.w, acc.w, s2.w);
        }
        __builtin_amdgcn_sched_barrier(0);                     // arrivals stay in program order (unpinned, hipcc hoists the address
      }                                                        // arithmetic of all 18 arrivals: 218 registers, two waves per SIMD)
      if (!more) break;
      it = itn; b = bn; h0 = h0n; wo = won;
    }
  }
  if (a.partial) dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase);
}

struct DwRaw { float4 d, y; };
// raw (D, Y) at output pixel (ho, wo), indices clamped into the image (the caller zero-selects what was outside)
__device__ __forceinline__ DwRaw dw_dy_raw_s1(const DwBwdArgs& a, int b, int ho, int wo, int c0) {
  const int hoc = ho < 0 ? 0 : (ho >= a.Ho ? a.Ho - 1 : ho), woc = wo < 0 ? 0 : (wo >= a.Wo ? a.Wo - 1 : wo);
  const int64_t q = (((int64_t)b * a.Ho + hoc) * a.Wo + woc) * a.C + c0;
  DwRaw r;
  r.d = kd_ld4(a.D + q);
  r.y = kd_ld4((a.al ? a.Y : a.D) + q);
  return r;
}
__device__ __forceinline__ float4 dw_dy_finish(const DwBwdArgs& a, DwRaw r, bool ok, float4 al, float4 be, float4 ga, float4 dsc,
                                               float4 dsh) {
  float4 v = r.d;
  if (a.al) {
    v.x = kd_bwd_operand(r.d.x, r.y.x, al.x, be.x, ga.x, dsc.x, dsh.x, a.d_act);
    v.y = kd_bwd_operand(r.d.y, r.y.y, al.y, be.y, ga.y, dsc.y, dsh.y, a.d_act);
    v.z = kd_bwd_operand(r.d.z, r.y.z, al.z, be.z, ga.z, dsc.z, dsh.z, a.d_act);
    v.w = kd_bwd_operand(r.d.w, r.y.w, al.w, be.w, ga.w, dsc.w, dsh.w, a.d_act);
  }
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// dyeff row (ho, wo-1..wo+1), zero outside the output image
__device__ __forceinline__ DwRow dw_load_dy_row(const DwBwdArgs& a, float4 al, float4 be, float4 ga, float4 dsc, float4 dsh,
                                                int b, int ho, int wo, int c0) {
  const bool hok = ho >= 0 && ho < a.Ho;
  const int hoc = ho < 0 ? 0 : (ho >= a.Ho ? a.Ho - 1 : ho);
  const int wl = wo - 1 < 0 ? 0 : wo - 1, wr_ = wo + 1 >= a.Wo ? a.Wo - 1 : wo + 1;
  const int64_t base = ((int64_t)b * a.Ho + hoc) * a.Wo;
  const float4 l = dw_dyeff(a, base + wl, c0, al, be, ga, dsc, dsh), c = dw_dyeff(a, base + wo, c0, al, be, ga, dsc, dsh),
               r = dw_dyeff(a, base + wr_, c0, al, be, ga, dsc, dsh);
  const bool lok = hok && wo - 1 >= 0, rok = hok && wo + 1 < a.Wo;
  DwRow o;
  o.l = make_float4(lok ? l.x : 0.f, lok ? l.y : 0.f, lok ? l.z : 0.f, lok ? l.w : 0.f);
  o.c = make_float4(hok ? c.x : 0.f, hok ? c.y : 0.f, hok ? c.z : 0.f, hok ? c.w : 0.f);
  o.r = make_float4(rok ? r.x : 0.f, rok ? r.y : 0.f, rok ? r.z : 0.f, rok ? r.w : 0.f);
  return o;
}

// stride-1 data gradient: gx[hi][wi] = mask * sum_{dh,dw} dy[hi+dh][wi+dw] * w[1-dh][1-dw]
__global__ __launch_bounds__(256) void dw_bwd_data_sw_kernel(DwBwdArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    float wf[4][9];                     // flipped taps: wf[j][(dh+1)*3 + (dw+1)] = w[j][(1-dh)*3 + (1-dw)]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) wf[j][t] = a.w[(c0 + j) * 9 + (8 - t)];
    float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mean = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const int nseg = (a.H + DW_SEG - 1) / DW_SEG;
    const int64_t items = (int64_t)a.B * nseg * a.W;
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int wi = (int)(it % a.W), sg = (int)((it / a.W) % nseg), b = (int)(it / ((int64_t)a.W * nseg));
      const int h0 = sg * DW_SEG, h1 = h0 + DW_SEG < a.H ? h0 + DW_SEG : a.H;
      DwRow r0 = dw_load_dy_row(a, al, be, ga, dsc, dsh, b, h0 - 1, wi, c0);
      DwRow r1 = dw_load_dy_row(a, al, be, ga, dsc, dsh, b, h0, wi, c0), r2;
      // One row AHEAD: the raw dy row hi + 1 (three columns x (D, Y)) and x(hi) are loaded before the store of row
      // hi - 1 is issued -- vmcnt retires in order, so a wait on a load that FOLLOWS a store also waits for the store.
      DwRaw nl = dw_dy_raw_s1(a, b, h0 + 1, wi - 1, c0), nc = dw_dy_raw_s1(a, b, h0 + 1, wi, c0), nr = dw_dy_raw_s1(a, b, h0 + 1, wi + 1, c0);
      float4 xn = kd_zero4();
      if (a.sc) xn = kd_ld4(a.x + (((int64_t)b * a.H + h0) * a.W + wi) * a.C + c0);
      for (int hi = h0; hi < h1; ++hi) {
        {
          const bool hok = hi + 1 < a.Ho;
          r2.l = dw_dy_finish(a, nl, hok && wi - 1 >= 0, al, be, ga, dsc, dsh);
          r2.c = dw_dy_finish(a, nc, hok, al, be, ga, dsc, dsh);
          r2.r = dw_dy_finish(a, nr, hok && wi + 1 < a.Wo, al, be, ga, dsc, dsh);
        }
        const float4 xr = xn;
        nl = dw_dy_raw_s1(a, b, hi + 2, wi - 1, c0); nc = dw_dy_raw_s1(a, b, hi + 2, wi, c0); nr = dw_dy_raw_s1(a, b, hi + 2, wi + 1, c0);
        if (a.sc) xn = kd_ld4(a.x + (((int64_t)b * a.H + (hi + 1 < a.H ? hi + 1 : hi)) * a.W + wi) * a.C + c0);
        float4 acc = kd_zero4();
        dw_fma_row(acc, r0, wf, 0);
        dw_fma_row(acc, r1, wf, 1);
        dw_fma_row(acc, r2, wf, 2);
        const int64_t p = ((int64_t)b * a.H + hi) * a.W + wi;
        if (a.sc) {
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
        if (a.nt) kd_st4_nt(a.gx + p * a.C + c0, acc); else kd_st4(a.gx + p * a.C + c0, acc);
        r0 = r1; r1 = r2;
      }
    }
  }
  if (a.partial) dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase);
}

// stride-1 data AND weight gradient in one pass (the separate kernels read the folded dy and the raw input twice: seven
// tensor passes over HBM; fused: four).  Same column-segment walk as dw_bwd_data_sw_kernel; the weight gradient rides on
// its dy window (see the loop).  The raw centre of each input row doubles as the operand of the activation mask and of
// the BatchNorm-backward sums.  A segment's last dy row and first / last input rows reach into the neighbouring segments:
// every (dy row, input row) pair is still counted exactly once, by the segment that owns the INPUT row.
template <bool ADD>
__global__ __launch_bounds__(256, 2) void dw_bwd_fused_s1_kernel(DwBwdArgs a) {
  __shared__ float red[2 * 256 * 4];
  __shared__ __attribute__((aligned(16))) float wl[9 * 1024];   // flipped taps, [tap][channel]: wl[t][c] = w[c][8 - t] (C <= 1024)
  // ADD: the input BatchNorm's mean / invstd live in LDS too (the addend's registers would otherwise push the kernel into scratch)
  __shared__ __attribute__((aligned(16))) float cf[ADD ? 2 * 1024 : 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  float4 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = kd_zero4();
  for (int i = tid; i < a.C * 9; i += 256) wl[(i % 9) * a.C + i / 9] = a.w[(i / 9) * 9 + (8 - i % 9)];
  if (ADD)
    for (int i = tid; i < a.C; i += 256) { cf[i] = a.mean ? a.mean[i] : 0.f; cf[1024 + i] = a.mean ? a.invstd[i] : 0.f; }
  __syncthreads();
  // (the nine taps live in LDS, not in 36 registers: with the dy window, the input window, the weight-gradient
  // accumulators and the one-row-ahead prefetch the kernel would otherwise spill)
  auto fma_row = [&](float4& acc, const DwRow& r, int kh) {
    const float4 w0 = kd_ld4(wl + (kh * 3 + 0) * a.C + c0), w1 = kd_ld4(wl + (kh * 3 + 1) * a.C + c0), w2 = kd_ld4(wl + (kh * 3 + 2) * a.C + c0);
    acc.x = fmaf(r.l.x, w0.x, fmaf(r.c.x, w1.x, fmaf(r.r.x, w2.x, acc.x)));
    acc.y = fmaf(r.l.y, w0.y, fmaf(r.c.y, w1.y, fmaf(r.r.y, w2.y, acc.y)));
    acc.z = fmaf(r.l.z, w0.z, fmaf(r.c.z, w1.z, fmaf(r.r.z, w2.z, acc.z)));
    acc.w = fmaf(r.l.w, w0.w, fmaf(r.c.w, w1.w, fmaf(r.r.w, w2.w, acc.w)));
  };
  if (active) {
    float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    const bool deferred = a.sc != nullptr;
    if (deferred) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (!ADD && a.mean) { mean = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const int nseg = (a.H + DW_SEG - 1) / DW_SEG;
    const int64_t items = (int64_t)a.B * nseg * a.W;
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int wi = (int)(it % a.W), sg = (int)((it / a.W) % nseg), b = (int)(it / ((int64_t)a.W * nseg));
      const int h0 = sg * DW_SEG, h1 = h0 + DW_SEG < a.H ? h0 + DW_SEG : a.H;
      DwRow r0 = dw_load_dy_row(a, al, be, ga, dsc, dsh, b, h0 - 1, wi, c0);
      DwRow r1 = dw_load_dy_row(a, al, be, ga, dsc, dsh, b, h0, wi, c0), r2;
      // One row AHEAD (in-order vmcnt: a wait on a load that FOLLOWS a store also waits for the store): the raw dy row
      // hi + 1 and the raw input row hi are requested before the store of row hi - 1 is issued.
      DwRaw nl = dw_dy_raw_s1(a, b, h0 + 1, wi - 1, c0), nc = dw_dy_raw_s1(a, b, h0 + 1, wi, c0), nr = dw_dy_raw_s1(a, b, h0 + 1, wi + 1, c0);
      DwRow xn = dw_load_row_raw(a.x, b, h0, wi, a.H, a.W, a.C, c0);
      for (int hi = h0; hi < h1; ++hi) {
        // the residual addend of THIS row: requested first, i.e. older than the next-row prefetches below, so that waiting for
        // it near the end of the iteration leaves those in flight (a row-ahead copy of it would cost the registers that spill)
        float4 ad = kd_zero4();
        if (ADD) ad = kd_ld4(a.addend + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0);
        {
          const bool hok = hi + 1 < a.Ho;
          r2.l = dw_dy_finish(a, nl, hok && wi - 1 >= 0, al, be, ga, dsc, dsh);
          r2.c = dw_dy_finish(a, nc, hok, al, be, ga, dsc, dsh);
          r2.r = dw_dy_finish(a, nr, hok && wi + 1 < a.Wo, al, be, ga, dsc, dsh);
        }
        const float4 xr = xn.c;                                           // raw centre: mask + BatchNorm-backward operand of row hi
        const DwRow xa = dw_finish_row(xn, deferred, sc, sh, a.act, hi, wi, a.H, a.W);
        nl = dw_dy_raw_s1(a, b, hi + 2, wi - 1, c0); nc = dw_dy_raw_s1(a, b, hi + 2, wi, c0); nr = dw_dy_raw_s1(a, b, hi + 2, wi + 1, c0);
        xn = dw_load_row_raw(a.x, b, hi + 1 < a.H ? hi + 1 : hi, wi, a.H, a.W, a.C, c0);
        // ---- weight gradient.  dw[kh][kw] = sum_ho dy(ho) x(ho - 1 + kh, wo - 1 + kw): the input row hi pairs with the dy
        // centres of rows hi + 1 (kh = 0), hi (kh = 1), hi - 1 (kh = 2), all three in the dy window (zero outside the image),
        // so no input window is kept: every input row is activated once and used once. ---------------------------------
#define KD_DW_WG(KH, D)                                                                                            \
  wacc[KH * 3 + 0].x = fmaf(D.x, xa.l.x, wacc[KH * 3 + 0].x); wacc[KH * 3 + 0].y = fmaf(D.y, xa.l.y, wacc[KH * 3 + 0].y); \
  wacc[KH * 3 + 0].z = fmaf(D.z, xa.l.z, wacc[KH * 3 + 0].z); wacc[KH * 3 + 0].w = fmaf(D.w, xa.l.w, wacc[KH * 3 + 0].w); \
  wacc[KH * 3 + 1].x = fmaf(D.x, xa.c.x, wacc[KH * 3 + 1].x); wacc[KH * 3 + 1].y = fmaf(D.y, xa.c.y, wacc[KH * 3 + 1].y); \
  wacc[KH * 3 + 1].z = fmaf(D.z, xa.c.z, wacc[KH * 3 + 1].z); wacc[KH * 3 + 1].w = fmaf(D.w, xa.c.w, wacc[KH * 3 + 1].w); \
  wacc[KH * 3 + 2].x = fmaf(D.x, xa.r.x, wacc[KH * 3 + 2].x); wacc[KH * 3 + 2].y = fmaf(D.y, xa.r.y, wacc[KH * 3 + 2].y); \
  wacc[KH * 3 + 2].z = fmaf(D.z, xa.r.z, wacc[KH * 3 + 2].z); wacc[KH * 3 + 2].w = fmaf(D.w, xa.r.w, wacc[KH * 3 + 2].w);
        KD_DW_WG(0, r2.c) KD_DW_WG(1, r1.c) KD_DW_WG(2, r0.c)
#undef KD_DW_WG
        // ---- data gradient ---------------------------------------------------------------------------------------------------
        float4 acc = kd_zero4();
        fma_row(acc, r0, 0);
        fma_row(acc, r1, 1);
        fma_row(acc, r2, 2);
        const int64_t p = ((int64_t)b * a.H + hi) * a.W + wi;
        if (ADD) { acc.x += ad.x; acc.y += ad.y; acc.z += ad.z; acc.w += ad.w; }
        if (deferred) {
          acc.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
          acc.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
          acc.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
          acc.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
          if (ADD) { mean = kd_ld4(cf + c0); inv = kd_ld4(cf + 1024 + c0); }
          s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
          s2.x = fmaf(acc.x, (xr.x - mean.x) * inv.x, s2.x);
          s2.y = fmaf(acc.y, (xr.y - mean.y) * inv.y, s2.y);
          s2.z = fmaf(acc.z, (xr.z - mean.z) * inv.z, s2.z);
          s2.w = fmaf(acc.w, (xr.w - mean.w) * inv.w, s2.w);
        }
        if (a.nt) kd_st4_nt(a.gx + p * a.C + c0, acc); else kd_st4(a.gx + p * a.C + c0, acc);
        r0 = r1; r1 = r2;
      }
    }
  }
  if (a.partial) { dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase); }
  for (int t = 0; t < 9; ++t) {        // per-block weight-gradient partials, summed in fixed order by kd_slab_reduce
    __syncthreads();
    kd_st4(red + tid * 4, active ? wacc[t] : kd_zero4());
    __syncthreads();
    for (int c = tid; c < a.groups * 4; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      a.wslab[(int64_t)bx * a.C * 9 + (cbase + c) * 9 + t] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// stride-1 backward, tile form: data gradient + BatchNorm-backward sums + weight gradient with every operand element loaded
// from HBM / L2 and transformed ONCE.  (In the column-walk kernels above each thread loads and BN-transforms its own left /
// centre / right neighbours: three times the loads, three times the transform arithmetic, and 250 registers of windows.)
// A workgroup owns a strip of 16 columns x 64 channels and walks down a row segment; per row it stages, cooperatively,
//   dyl : the folded dy  (al*mask(D) + be*Y + ga, zero outside the image), ring of 4 rows x 18 columns,
//   xal : the activated input row and xrl : the raw input row (mask + BatchNorm-backward operand), rings of 2 rows,
// then thread (column, channel quad) reads its 3x3 dy neighbourhood (data gradient) and the activated input row (weight
// gradient: input row hi pairs with the dy centres of rows hi+1 / hi / hi-1, see dw_bwd_fused_s1_kernel) from LDS.
// One barrier per row; registers: 9 taps + 9 weight-gradient accumulators + coefficients.
constexpr int DT_COLS = 16, DT_CQ = 16, DT_LC = DT_COLS + 2;      // strip columns, float4 channel quads per chunk, staged columns
__global__ __launch_bounds__(256) void dw_bwd_tile_s1_kernel(DwBwdArgs a, int nrow_slab, int nchunk) {
  __shared__ __attribute__((aligned(16))) float4 dyl[4][DT_LC][DT_CQ];
  __shared__ __attribute__((aligned(16))) float4 xal[2][DT_LC][DT_CQ];
  __shared__ __attribute__((aligned(16))) float4 xrl[2][DT_LC][DT_CQ];
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x, tx = tid % DT_CQ, ty = tid / DT_CQ;
  const int chunk = blockIdx.x % nchunk, srow = blockIdx.x / nchunk;
  const int c0 = chunk * (DT_CQ * 4) + tx * 4;
  const bool cok = c0 < a.C;                                         // (C need not be a multiple of 64)
  const int cc = cok ? c0 : 0;
  const bool deferred = a.sc != nullptr;
  float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
  if (a.al) { al = kd_ld4(a.al + cc); be = kd_ld4(a.be + cc); ga = kd_ld4(a.ga + cc); }
  if (a.dsc) { dsc = kd_ld4(a.dsc + cc); dsh = kd_ld4(a.dsh + cc); }
  if (deferred) { sc = kd_ld4(a.sc + cc); sh = kd_ld4(a.sh + cc); }
  if (a.mean) { mean = kd_ld4(a.mean + cc); inv = kd_ld4(a.invstd + cc); }
  float4 wf[9];                                                       // flipped taps of the thread's 4 channels
#pragma unroll
  for (int t = 0; t < 9; ++t) wf[t] = make_float4(a.w[(cc + 0) * 9 + 8 - t], a.w[(cc + 1) * 9 + 8 - t], a.w[(cc + 2) * 9 + 8 - t], a.w[(cc + 3) * 9 + 8 - t]);
  float4 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = kd_zero4();
  float4 s1 = kd_zero4(), s2 = kd_zero4();

  const int nstrip = (a.W + DT_COLS - 1) / DT_COLS, nseg = (a.H + DW_SEG - 1) / DW_SEG;
  const int64_t items = (int64_t)a.B * nseg * nstrip;
  // staging roles: element e = tid (and e = 256 + tid for tid < 32) of the 18 x 16 row image
  const int l0c = tid / DT_CQ, l1c = DT_COLS + tid / DT_CQ;           // staged column of the primary / secondary element
  const bool two = tid < 2 * DT_CQ;
  // Staging is split in two so that a row's HBM loads fly during the arithmetic of the row before it: fetch_* issues the raw
  // loads into registers, commit_* (one iteration later) transforms them and writes the LDS image.
  struct RawDy { DwRaw e[2]; bool ok[2]; };
  struct RawX { float4 e[2]; bool ok[2]; };
  auto fetch_dy = [&](int b, int ho, int w0, RawDy& o) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int lc = k ? l1c : l0c, wo = w0 - 1 + lc;
      o.ok[k] = cok && ho >= 0 && ho < a.Ho && wo >= 0 && wo < a.Wo;
      const int hoc = ho < 0 ? 0 : (ho >= a.Ho ? a.Ho - 1 : ho), woc = wo < 0 ? 0 : (wo >= a.Wo ? a.Wo - 1 : wo);
      const int64_t q = (((int64_t)b * a.Ho + hoc) * a.Wo + woc) * a.C + cc;
      if (k == 0 || two) { o.e[k].d = kd_ld4(a.D + q); o.e[k].y = kd_ld4((a.al ? a.Y : a.D) + q); }
    }
  };
  auto commit_dy = [&](const RawDy& o, int slot) {
    dyl[slot][l0c][tx] = dw_dy_finish(a, o.e[0], o.ok[0], al, be, ga, dsc, dsh);
    if (two) dyl[slot][l1c][tx] = dw_dy_finish(a, o.e[1], o.ok[1], al, be, ga, dsc, dsh);
  };
  auto fetch_x = [&](int b, int hi, int w0, RawX& o) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int lc = k ? l1c : l0c, wi = w0 - 1 + lc;
      o.ok[k] = cok && hi >= 0 && hi < a.H && wi >= 0 && wi < a.W;
      const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi), wic = wi < 0 ? 0 : (wi >= a.W ? a.W - 1 : wi);
      if (k == 0 || two) o.e[k] = kd_ld4(a.x + (((int64_t)b * a.H + hic) * a.W + wic) * a.C + cc);
    }
  };
  auto commit_x = [&](const RawX& o, int slot) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (k == 1 && !two) break;
      const int lc = k ? l1c : l0c;
      const float4 xr = o.e[k];
      const float4 xa = deferred ? kd_affine_act4(xr, sc, sh, a.act) : xr;
      const bool ok = o.ok[k];
      xal[slot][lc][tx] = make_float4(ok ? xa.x : 0.f, ok ? xa.y : 0.f, ok ? xa.z : 0.f, ok ? xa.w : 0.f);
      xrl[slot][lc][tx] = xr;
    }
  };
  RawDy pd, pd2;
  RawX px;
  auto item_coords = [&](int64_t t, int& b, int& w0, int& h0) {
    const int strip = (int)(t % nstrip), sg = (int)((t / nstrip) % nseg);
    b = (int)(t / ((int64_t)nstrip * nseg));
    w0 = strip * DT_COLS; h0 = sg * DW_SEG;
  };
  if (srow < items) {
    int b, w0, h0;
    item_coords(srow, b, w0, h0);
    fetch_dy(b, h0 - 1, w0, pd);
    fetch_dy(b, h0, w0, pd2);
  }
  for (int64_t it = srow; it < items; it += nrow_slab) {
    int b, w0, h0;
    item_coords(it, b, w0, h0);
    const int h1 = h0 + DW_SEG < a.H ? h0 + DW_SEG : a.H;
    int bn, w0n, h0n;                                                 // the workgroup's next item (this one again when there is none)
    item_coords(it + nrow_slab < items ? it + nrow_slab : it, bn, w0n, h0n);
    __syncthreads();                                                  // the previous item's readers are done with the rings
    commit_dy(pd, (h0 + 3) & 3);
    commit_dy(pd2, h0 & 3);
    fetch_dy(b, h0 + 1, w0, pd);
    fetch_x(b, h0, w0, px);
    auto row = [&](int hi, auto last_tag) __attribute__((always_inline)) {
      commit_dy(pd, (hi + 1) & 3);
      commit_x(px, hi & 1);
      if constexpr (!decltype(last_tag)::value) {
        fetch_dy(b, hi + 2, w0, pd);                                  // in flight during this row's arithmetic
        fetch_x(b, hi + 1, w0, px);
      } else {                                                        // last row of the item: the NEXT item's first two dy rows instead of
        fetch_dy(bn, h0n - 1, w0n, pd);                               // rows nobody uses -- the next item does not start cold
        fetch_dy(bn, h0n, w0n, pd2);
      }
      __syncthreads();                                                // ring depth 4 / 2: one barrier per row is enough
      const int lc = ty + 1, wi = w0 + ty;
      const float4 (*r0)[DT_CQ] = dyl[(hi + 3) & 3], (*r1)[DT_CQ] = dyl[hi & 3], (*r2)[DT_CQ] = dyl[(hi + 1) & 3];
      const float4 d0l = r0[lc - 1][tx], d0c = r0[lc][tx], d0r = r0[lc + 1][tx];
      const float4 d1l = r1[lc - 1][tx], d1c = r1[lc][tx], d1r = r1[lc + 1][tx];
      const float4 d2l = r2[lc - 1][tx], d2c = r2[lc][tx], d2r = r2[lc + 1][tx];
      const float4 xl = xal[hi & 1][lc - 1][tx], xc = xal[hi & 1][lc][tx], xrr = xal[hi & 1][lc + 1][tx], xr = xrl[hi & 1][lc][tx];
#define KD_FMA4(ACC, A_, B_) ACC.x = fmaf(A_.x, B_.x, ACC.x); ACC.y = fmaf(A_.y, B_.y, ACC.y); ACC.z = fmaf(A_.z, B_.z, ACC.z); ACC.w = fmaf(A_.w, B_.w, ACC.w);
      // weight gradient (same pairing and order as dw_bwd_fused_s1_kernel)
      KD_FMA4(wacc[0], d2c, xl) KD_FMA4(wacc[1], d2c, xc) KD_FMA4(wacc[2], d2c, xrr)
      KD_FMA4(wacc[3], d1c, xl) KD_FMA4(wacc[4], d1c, xc) KD_FMA4(wacc[5], d1c, xrr)
      KD_FMA4(wacc[6], d0c, xl) KD_FMA4(wacc[7], d0c, xc) KD_FMA4(wacc[8], d0c, xrr)
      // data gradient: same fma nesting as dw_fma_row (r, then c, then l; rows 0, 1, 2)
      float4 acc = kd_zero4();
      KD_FMA4(acc, d0r, wf[2]) KD_FMA4(acc, d0c, wf[1]) KD_FMA4(acc, d0l, wf[0])
      KD_FMA4(acc, d1r, wf[5]) KD_FMA4(acc, d1c, wf[4]) KD_FMA4(acc, d1l, wf[3])
      KD_FMA4(acc, d2r, wf[8]) KD_FMA4(acc, d2c, wf[7]) KD_FMA4(acc, d2l, wf[6])
#undef KD_FMA4
      if (cok && wi < a.W) {
        if (deferred) {
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
        float* gp = a.gx + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0;
        if (a.nt) kd_st4_nt(gp, acc); else kd_st4(gp, acc);
      }
    };
    for (int hi = h0; hi < h1 - 1; ++hi) row(hi, std::false_type{});
    row(h1 - 1, std::true_type{});
  }
  // ---- per-workgroup partials: slab row `srow`, this chunk's channels (every (row, chunk) pair is written by exactly one
  // workgroup, zeros included, so the fixed-order slab reductions need no initialisation) ------------------------------------
  const int cbase = chunk * (DT_CQ * 4);
  for (int t = 0; t < 11; ++t) {
    const float4 v = t < 9 ? wacc[t] : (t == 9 ? s1 : s2);
    __syncthreads();
    kd_st4(red + tid * 4, v);
    __syncthreads();
    if (tid < DT_CQ * 4 && cbase + tid < a.C) {
      float s = 0.f;
      for (int yy = 0; yy < DT_COLS; ++yy) s += red[(yy * DT_CQ + tid / 4) * 4 + (tid & 3)];
      const int c = cbase + tid;
      if (t < 9) a.wslab[(int64_t)srow * a.C * 9 + c * 9 + t] = s;
      else if (a.partial) a.partial[((int64_t)srow * 2 + (t - 9)) * a.C + c] = s;
    }
  }
}

// stride-2 data gradient.  Work item = a column of 2x2 input QUADS (b, 8 quad rows, quad column q): the quad with
// top-left input pixel (2a, 2q) receives from exactly the four outputs (a, q), (a, q+1), (a+1, q), (a+1, q+1):
//   gx(2a  , 2q  ) = d00 w11                      gx(2a  , 2q+1) = d01 w10 + d00 w12
//   gx(2a+1, 2q  ) = d10 w01 + d00 w21            gx(2a+1, 2q+1) = d11 w00 + d10 w02 + d01 w20 + d00 w22
// so sliding down the quad rows costs TWO new dy values per four input pixels (the per-pixel form loaded and
// BN-transformed four candidates per pixel: 8x the loads and arithmetic of this one).  Loads are branch-free
// (clamped addresses, zero selects).
__device__ __forceinline__ DwRaw dw_dy_raw(const DwBwdArgs& a, int b, int ho, int wo, int c0) {   // clamped, no branches
  const int hoc = ho < a.Ho ? ho : a.Ho - 1, woc = wo < a.Wo ? wo : a.Wo - 1;
  const int64_t q = (((int64_t)b * a.Ho + hoc) * a.Wo + woc) * a.C + c0;
  DwRaw r;
  r.d = kd_ld4(a.D + q);
  r.y = kd_ld4((a.al ? a.Y : a.D) + q);
  return r;
}
__global__ __launch_bounds__(256) void dw_bwd_data_s2_kernel(DwBwdArgs a) {
  __shared__ float red[2 * 256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    float4 wt[9];                         // wt[t] = tap t of the thread's 4 channels
    {
      float wreg[4][9];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = make_float4(wreg[0][t], wreg[1][t], wreg[2][t], wreg[3][t]);
    }
    float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mean = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const bool masked = a.sc != nullptr;
    const int QH = (a.H + 1) / 2, QW = (a.W + 1) / 2;
    constexpr int QSEG = DW_SEG / 2;
    const int nseg = (QH + QSEG - 1) / QSEG;
    const int64_t items = (int64_t)a.B * nseg * QW;
    // raw x of the quad's four pixels (clamped addresses; pixels outside an odd-sized image are dropped at the store)
    auto load_x = [&](int b, int qa, int q, float4 (&xq)[4]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int hi = 2 * qa + (k >> 1), wi = 2 * q + (k & 1);
        hi = hi < a.H ? hi : a.H - 1; wi = wi < a.W ? wi : a.W - 1;
        xq[k] = kd_ld4(a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0);
      }
    };
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int q = (int)(it % QW), sg = (int)((it / QW) % nseg), b = (int)(it / ((int64_t)QW * nseg));
      const int a0 = sg * QSEG, a1 = a0 + QSEG < QH ? a0 + QSEG : QH;
      const bool w0ok = q < a.Wo, w1ok = q + 1 < a.Wo;
      float4 d00 = dw_dy_finish(a, dw_dy_raw(a, b, a0, q, c0), a0 < a.Ho && w0ok, al, be, ga, dsc, dsh);
      float4 d01 = dw_dy_finish(a, dw_dy_raw(a, b, a0, q + 1, c0), a0 < a.Ho && w1ok, al, be, ga, dsc, dsh);
      // Everything a quad row needs is loaded one iteration AHEAD, i.e. before the previous row's stores are issued:
      // vmcnt retires in order, so a wait on a load that follows a store would also wait for that store.
      DwRaw rn0 = dw_dy_raw(a, b, a0 + 1, q, c0), rn1 = dw_dy_raw(a, b, a0 + 1, q + 1, c0);
      float4 xn[4];
      if (masked) load_x(b, a0, q, xn);
      for (int qa = a0; qa < a1; ++qa) {
        const bool hok = qa + 1 < a.Ho;
        const float4 d10 = dw_dy_finish(a, rn0, hok && w0ok, al, be, ga, dsc, dsh), d11 = dw_dy_finish(a, rn1, hok && w1ok, al, be, ga, dsc, dsh);
        float4 xc[4] = {xn[0], xn[1], xn[2], xn[3]};
        rn0 = dw_dy_raw(a, b, qa + 2, q, c0); rn1 = dw_dy_raw(a, b, qa + 2, q + 1, c0);
        if (masked) load_x(b, qa + 1 < QH ? qa + 1 : qa, q, xn);
        float4 g[4];
        g[0].x = d00.x * wt[4].x; g[0].y = d00.y * wt[4].y; g[0].z = d00.z * wt[4].z; g[0].w = d00.w * wt[4].w;
        g[1].x = fmaf(d01.x, wt[3].x, d00.x * wt[5].x); g[1].y = fmaf(d01.y, wt[3].y, d00.y * wt[5].y);
        g[1].z = fmaf(d01.z, wt[3].z, d00.z * wt[5].z); g[1].w = fmaf(d01.w, wt[3].w, d00.w * wt[5].w);
        g[2].x = fmaf(d10.x, wt[1].x, d00.x * wt[7].x); g[2].y = fmaf(d10.y, wt[1].y, d00.y * wt[7].y);
        g[2].z = fmaf(d10.z, wt[1].z, d00.z * wt[7].z); g[2].w = fmaf(d10.w, wt[1].w, d00.w * wt[7].w);
        g[3].x = fmaf(d11.x, wt[0].x, fmaf(d10.x, wt[2].x, fmaf(d01.x, wt[6].x, d00.x * wt[8].x)));
        g[3].y = fmaf(d11.y, wt[0].y, fmaf(d10.y, wt[2].y, fmaf(d01.y, wt[6].y, d00.y * wt[8].y)));
        g[3].z = fmaf(d11.z, wt[0].z, fmaf(d10.z, wt[2].z, fmaf(d01.z, wt[6].z, d00.z * wt[8].z)));
        g[3].w = fmaf(d11.w, wt[0].w, fmaf(d10.w, wt[2].w, fmaf(d01.w, wt[6].w, d00.w * wt[8].w)));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hi = 2 * qa + (k >> 1), wi = 2 * q + (k & 1);
          const bool ok = hi < a.H && wi < a.W;
          float4 v = g[k];
          if (masked) {
            const float4 xr = xc[k];
            v.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
            v.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
            v.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
            v.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
            if (ok) {
              s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
              s2.x = fmaf(v.x, (xr.x - mean.x) * inv.x, s2.x);
              s2.y = fmaf(v.y, (xr.y - mean.y) * inv.y, s2.y);
              s2.z = fmaf(v.z, (xr.z - mean.z) * inv.z, s2.z);
              s2.w = fmaf(v.w, (xr.w - mean.w) * inv.w, s2.w);
            }
          }
          if (ok) { if (a.nt) kd_st4_nt(a.gx + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0, v); else kd_st4(a.gx + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0, v); }
        }
        d00 = d10; d01 = d11;
      }
    }
  }
  if (a.partial) dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase);
}

// stride-2 data AND weight gradient in one pass over the same quads: the pairs (dy, input pixel) of the weight gradient
//   dw[kh][kw] = sum dy(ho, wo) x(2 ho - 1 + kh, 2 wo - 1 + kw)
// that involve the quad's four pixels are exactly the nine products of the data-gradient formulas above (pixel (0,0): d00 ->
// tap 11; (0,1): d01 -> 10, d00 -> 12; (1,0): d10 -> 01, d00 -> 21; (1,1): d11 -> 00, d10 -> 02, d01 -> 20, d00 -> 22), every
// input pixel belongs to one quad, so each pair is counted once.  The separate kernels read dy, Y and the raw input twice.
__global__ __launch_bounds__(256) void dw_bwd_fused_s2_kernel(DwBwdArgs a) {
  __shared__ float red[2 * 256 * 4];
  float4 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = kd_zero4();
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  if (active) {
    float4 wt[9];                         // wt[t] = tap t of the thread's 4 channels
    {
      float wreg[4][9];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wreg[j][t] = a.w[(c0 + j) * 9 + t];
#pragma unroll
      for (int t = 0; t < 9; ++t) wt[t] = make_float4(wreg[0][t], wreg[1][t], wreg[2][t], wreg[3][t]);
    }
    float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4(), mean = kd_zero4(), inv = kd_zero4();
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    if (a.sc) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    if (a.mean) { mean = kd_ld4(a.mean + c0); inv = kd_ld4(a.invstd + c0); }
    const bool masked = a.sc != nullptr;
    const int QH = (a.H + 1) / 2, QW = (a.W + 1) / 2;
    constexpr int QSEG = DW_SEG / 2;
    const int nseg = (QH + QSEG - 1) / QSEG;
    const int64_t items = (int64_t)a.B * nseg * QW;
    // raw x of the quad's four pixels (clamped addresses; pixels outside an odd-sized image are dropped at the store)
    auto load_x = [&](int b, int qa, int q, float4 (&xq)[4]) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int hi = 2 * qa + (k >> 1), wi = 2 * q + (k & 1);
        hi = hi < a.H ? hi : a.H - 1; wi = wi < a.W ? wi : a.W - 1;
        xq[k] = kd_ld4(a.x + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0);
      }
    };
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int q = (int)(it % QW), sg = (int)((it / QW) % nseg), b = (int)(it / ((int64_t)QW * nseg));
      const int a0 = sg * QSEG, a1 = a0 + QSEG < QH ? a0 + QSEG : QH;
      const bool w0ok = q < a.Wo, w1ok = q + 1 < a.Wo;
      float4 d00 = dw_dy_finish(a, dw_dy_raw(a, b, a0, q, c0), a0 < a.Ho && w0ok, al, be, ga, dsc, dsh);
      float4 d01 = dw_dy_finish(a, dw_dy_raw(a, b, a0, q + 1, c0), a0 < a.Ho && w1ok, al, be, ga, dsc, dsh);
      // Everything a quad row needs is loaded one iteration AHEAD, i.e. before the previous row's stores are issued:
      // vmcnt retires in order, so a wait on a load that follows a store would also wait for that store.
      DwRaw rn0 = dw_dy_raw(a, b, a0 + 1, q, c0), rn1 = dw_dy_raw(a, b, a0 + 1, q + 1, c0);
      float4 xn[4];
      load_x(b, a0, q, xn);
      for (int qa = a0; qa < a1; ++qa) {
        const bool hok = qa + 1 < a.Ho;
        const float4 d10 = dw_dy_finish(a, rn0, hok && w0ok, al, be, ga, dsc, dsh), d11 = dw_dy_finish(a, rn1, hok && w1ok, al, be, ga, dsc, dsh);
        float4 xc[4] = {xn[0], xn[1], xn[2], xn[3]};
        rn0 = dw_dy_raw(a, b, qa + 2, q, c0); rn1 = dw_dy_raw(a, b, qa + 2, q + 1, c0);
        load_x(b, qa + 1 < QH ? qa + 1 : qa, q, xn);
        float4 g[4];
        g[0].x = d00.x * wt[4].x; g[0].y = d00.y * wt[4].y; g[0].z = d00.z * wt[4].z; g[0].w = d00.w * wt[4].w;
        g[1].x = fmaf(d01.x, wt[3].x, d00.x * wt[5].x); g[1].y = fmaf(d01.y, wt[3].y, d00.y * wt[5].y);
        g[1].z = fmaf(d01.z, wt[3].z, d00.z * wt[5].z); g[1].w = fmaf(d01.w, wt[3].w, d00.w * wt[5].w);
        g[2].x = fmaf(d10.x, wt[1].x, d00.x * wt[7].x); g[2].y = fmaf(d10.y, wt[1].y, d00.y * wt[7].y);
        g[2].z = fmaf(d10.z, wt[1].z, d00.z * wt[7].z); g[2].w = fmaf(d10.w, wt[1].w, d00.w * wt[7].w);
        g[3].x = fmaf(d11.x, wt[0].x, fmaf(d10.x, wt[2].x, fmaf(d01.x, wt[6].x, d00.x * wt[8].x)));
        g[3].y = fmaf(d11.y, wt[0].y, fmaf(d10.y, wt[2].y, fmaf(d01.y, wt[6].y, d00.y * wt[8].y)));
        g[3].z = fmaf(d11.z, wt[0].z, fmaf(d10.z, wt[2].z, fmaf(d01.z, wt[6].z, d00.z * wt[8].z)));
        g[3].w = fmaf(d11.w, wt[0].w, fmaf(d10.w, wt[2].w, fmaf(d01.w, wt[6].w, d00.w * wt[8].w)));
        {   // weight gradient: the activated input of the quad (zero outside an odd-sized image)
          float4 xa[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bool ok = 2 * qa + (k >> 1) < a.H && 2 * q + (k & 1) < a.W;
            const float4 t = masked ? kd_affine_act4(xc[k], sc, sh, a.act) : xc[k];
            xa[k] = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
          }
#define KD_WG(T, D, X) wacc[T].x = fmaf(D.x, X.x, wacc[T].x); wacc[T].y = fmaf(D.y, X.y, wacc[T].y); wacc[T].z = fmaf(D.z, X.z, wacc[T].z); wacc[T].w = fmaf(D.w, X.w, wacc[T].w);
          KD_WG(4, d00, xa[0])
          KD_WG(3, d01, xa[1]) KD_WG(5, d00, xa[1])
          KD_WG(1, d10, xa[2]) KD_WG(7, d00, xa[2])
          KD_WG(0, d11, xa[3]) KD_WG(2, d10, xa[3]) KD_WG(6, d01, xa[3]) KD_WG(8, d00, xa[3])
#undef KD_WG
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int hi = 2 * qa + (k >> 1), wi = 2 * q + (k & 1);
          const bool ok = hi < a.H && wi < a.W;
          float4 v = g[k];
          if (masked) {
            const float4 xr = xc[k];
            v.x *= kd_act_mask(kd_affine(xr.x, sc.x, sh.x), a.act);
            v.y *= kd_act_mask(kd_affine(xr.y, sc.y, sh.y), a.act);
            v.z *= kd_act_mask(kd_affine(xr.z, sc.z, sh.z), a.act);
            v.w *= kd_act_mask(kd_affine(xr.w, sc.w, sh.w), a.act);
            if (ok) {
              s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
              s2.x = fmaf(v.x, (xr.x - mean.x) * inv.x, s2.x);
              s2.y = fmaf(v.y, (xr.y - mean.y) * inv.y, s2.y);
              s2.z = fmaf(v.z, (xr.z - mean.z) * inv.z, s2.z);
              s2.w = fmaf(v.w, (xr.w - mean.w) * inv.w, s2.w);
            }
          }
          if (ok) { if (a.nt) kd_st4_nt(a.gx + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0, v); else kd_st4(a.gx + (((int64_t)b * a.H + hi) * a.W + wi) * a.C + c0, v); }
        }
        d00 = d10; d01 = d11;
      }
    }
  }
  if (a.partial) dw_block_stats(red, s1, s2, a.partial, a.C, a.groups, a.slots, bx, cbase);
  for (int t = 0; t < 9; ++t) {        // per-block weight-gradient partials, summed in fixed order by kd_slab_reduce
    __syncthreads();
    kd_st4(red + tid * 4, active ? wacc[t] : kd_zero4());
    __syncthreads();
    for (int c = tid; c < a.groups * 4; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      a.wslab[(int64_t)bx * a.C * 9 + (cbase + c) * 9 + t] = s;
    }
  }
}

// weight gradient with the same input window as the forward
template <int STRIDE>
__global__ __launch_bounds__(256) void dw_bwd_weight_sw_kernel(DwBwdArgs a) {
  __shared__ float red[256 * 4];
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  const bool active = slot < a.slots;
  const int bx = blockIdx.x / a.nchunk, nbx = gridDim.x / a.nchunk, cbase = (blockIdx.x % a.nchunk) * a.groups * 4;
  const int c0 = cbase + gidx * 4;
  float4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = kd_zero4();
  if (active) {
    float4 al = kd_zero4(), be = kd_zero4(), ga = kd_zero4(), dsc = kd_zero4(), dsh = kd_zero4();
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = kd_zero4();
    if (a.al) { al = kd_ld4(a.al + c0); be = kd_ld4(a.be + c0); ga = kd_ld4(a.ga + c0); }
    if (a.dsc) { dsc = kd_ld4(a.dsc + c0); dsh = kd_ld4(a.dsh + c0); }
    const bool deferred = a.sc != nullptr;
    if (deferred) { sc = kd_ld4(a.sc + c0); sh = kd_ld4(a.sh + c0); }
    const int nseg = (a.Ho + DW_SEG - 1) / DW_SEG;
    const int64_t items = (int64_t)a.B * nseg * a.Wo;
    for (int64_t it = (int64_t)bx * a.slots + slot; it < items; it += (int64_t)nbx * a.slots) {
      const int wo = (int)(it % a.Wo), sg = (int)((it / a.Wo) % nseg), b = (int)(it / ((int64_t)a.Wo * nseg));
      const int h0 = sg * DW_SEG, h1 = h0 + DW_SEG < a.Ho ? h0 + DW_SEG : a.Ho;
      const int wi = wo * STRIDE;
      DwRow r0 = dw_load_row(a.x, deferred, sc, sh, a.act, b, h0 * STRIDE - 1, wi, a.H, a.W, a.C, c0), r1, r2;
      if (STRIDE == 1) r1 = dw_load_row(a.x, deferred, sc, sh, a.act, b, h0, wi, a.H, a.W, a.C, c0);
      // the raw operands of row ho + 1 are loaded before row ho is accumulated (27 x 4 FMAs hide the latency)
      DwRow n1, n2;
      if (STRIDE == 2) n1 = dw_load_row_raw(a.x, b, 2 * h0, wi, a.H, a.W, a.C, c0);
      n2 = dw_load_row_raw(a.x, b, h0 * STRIDE + 1, wi, a.H, a.W, a.C, c0);
      DwRaw nd = dw_dy_raw(a, b, h0, wo, c0);
      for (int ho = h0; ho < h1; ++ho) {
        if (STRIDE == 2) r1 = dw_finish_row(n1, deferred, sc, sh, a.act, 2 * ho, wi, a.H, a.W);
        r2 = dw_finish_row(n2, deferred, sc, sh, a.act, ho * STRIDE + 1, wi, a.H, a.W);
        const float4 d = dw_dy_finish(a, nd, true, al, be, ga, dsc, dsh);
        {
          const int hn = ho + 1 < a.Ho ? ho + 1 : ho;      // clamped: the prefetch after the last row is discarded
          if (STRIDE == 2) n1 = dw_load_row_raw(a.x, b, 2 * hn, wi, a.H, a.W, a.C, c0);
          n2 = dw_load_row_raw(a.x, b, hn * STRIDE + 1, wi, a.H, a.W, a.C, c0);
          nd = dw_dy_raw(a, b, hn, wo, c0);
        }
#define KD_DW_WG(KH, R)                                                                                    \
  acc[KH * 3 + 0].x = fmaf(d.x, R.l.x, acc[KH * 3 + 0].x); acc[KH * 3 + 0].y = fmaf(d.y, R.l.y, acc[KH * 3 + 0].y); \
  acc[KH * 3 + 0].z = fmaf(d.z, R.l.z, acc[KH * 3 + 0].z); acc[KH * 3 + 0].w = fmaf(d.w, R.l.w, acc[KH * 3 + 0].w); \
  acc[KH * 3 + 1].x = fmaf(d.x, R.c.x, acc[KH * 3 + 1].x); acc[KH * 3 + 1].y = fmaf(d.y, R.c.y, acc[KH * 3 + 1].y); \
  acc[KH * 3 + 1].z = fmaf(d.z, R.c.z, acc[KH * 3 + 1].z); acc[KH * 3 + 1].w = fmaf(d.w, R.c.w, acc[KH * 3 + 1].w); \
  acc[KH * 3 + 2].x = fmaf(d.x, R.r.x, acc[KH * 3 + 2].x); acc[KH * 3 + 2].y = fmaf(d.y, R.r.y, acc[KH * 3 + 2].y); \
  acc[KH * 3 + 2].z = fmaf(d.z, R.r.z, acc[KH * 3 + 2].z); acc[KH * 3 + 2].w = fmaf(d.w, R.r.w, acc[KH * 3 + 2].w);
        KD_DW_WG(0, r0) KD_DW_WG(1, r1) KD_DW_WG(2, r2)
#undef KD_DW_WG
        if (STRIDE == 1) { r0 = r1; r1 = r2; } else { r0 = r2; }
      }
    }
  }
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
    kd_st4(red + tid * 4, acc[t]);
    __syncthreads();
    for (int c = tid; c < a.groups * 4; c += 256) {
      float s = 0.f;
      for (int sl = 0; sl < a.slots; ++sl) s += red[(sl * a.groups + c / 4) * 4 + (c & 3)];
      a.wslab[(int64_t)bx * a.C * 9 + (cbase + c) * 9 + t] = s;
    }
  }
}

}  // namespace

// stride-1 backward form: 0 separate data / weight kernels, 1 fused column walk, 2 fused tile form, 3 (default) by shape
// (dw_fused_form below: the measurements behind the choice).  KD_DW_FUSED / kd_set_dw_bwd_mode.
static std::atomic<int> g_dw_mode{-1};
static int kd_dw_fused_mode() {
  int m = g_dw_mode.load(std::memory_order_relaxed);
  if (m < 0) {
    const char* e = getenv("KD_DW_FUSED");
    m = (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 3;
    g_dw_mode.store(m, std::memory_order_relaxed);
  }
  return m;
}

// Thread layout of the column-walk kernels.  A block is 256 threads = `slots` adjacent image columns x `groups` channel quads
// (float4 lanes).  Up to 64 quads one block spans all channels; wider tensors (C = 384 / 768: 96 / 192 quads left room for
// two columns / one column per block, so the three-column windows of neighbouring columns were fetched by different
// workgroups -- usually on different XCDs, i.e. through different L2s) are cut into chunks of 32 quads: 8 adjacent columns
// per block again, their overlapping window loads served by the CU's own cache.  blockIdx = row * nchunk + chunk; a block
// ROW owns one statistics / weight-gradient slab row (every chunk writes its own channel range of it).
struct DwLayout { int groups, slots, nchunk, rows, grid; };
static DwLayout dw_layout(int64_t npix, int C) {
  DwLayout l;
  const int quads = C / 4;
  l.nchunk = (quads > 64 && quads % 32 == 0) ? quads / 32 : 1;
  l.groups = quads / l.nchunk;
  l.slots = 256 / l.groups;
  if (l.slots < 1) l.slots = 1;
  const int64_t need = (npix + l.slots - 1) / l.slots;
  const int cap = 2048 / l.nchunk;
  l.rows = (int)(need < cap ? need : cap);
  if (l.rows < 1) l.rows = 1;
  l.grid = l.rows * l.nchunk;
  return l;
}

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
  hipStream_t st = (hipStream_t)stream;
  static const bool old_form = getenv("KD_STEM_FORM") && atoi(getenv("KD_STEM_FORM")) == 1;      // A/B: the LDS-weights form of rounds 1-2
  if (Cin == 3 && !old_form) {
    if (partial) hipLaunchKernelGGL((stem_fwd2_kernel<3, true>), dim3(grid), dim3(256), 0, st, x_nchw, w, y_nhwc, partial, B, H, W, Ho, Wo);
    else hipLaunchKernelGGL((stem_fwd2_kernel<3, false>), dim3(grid), dim3(256), 0, st, x_nchw, w, y_nhwc, partial, B, H, W, Ho, Wo);
    return kd_check_launch("kd_stem_conv_fwd");
  }
  const size_t shm = (size_t)(Cin * 9 * 32 + 256 * 33) * sizeof(float);
  hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(256), shm, st, x_nchw, w, y_nhwc, partial, B,
                     Cin, H, W, Ho, Wo);
  return kd_check_launch("kd_stem_conv_fwd");
}

// inference: y = act(bn(conv(x))) in one kernel (eval-mode coefficients sc / sh of the stem's BatchNorm)
int kd_stem_conv_fwd_infer(const float* x_nchw, const float* w, const float* sc, const float* sh, int act, float* y_nhwc, int B,
                           int Cin, int H, int W, int Cout, void* stream) {
  KD_REQUIRE(x_nchw && w && sc && sh && y_nhwc && B > 0 && H > 0 && W > 0, KD_ERR_ARG, "kd_stem_conv_fwd_infer: bad args");
  KD_REQUIRE(Cout == 32 && Cin == 3, KD_ERR_SHAPE, "kd_stem_conv_fwd_infer: only Cout=32, Cin=3 (got %d,%d)", Cout, Cin);
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int grid = (int)kd_stem_stat_rows((int64_t)B * Ho * Wo);
  hipLaunchKernelGGL((stem_fwd2_kernel<3, false, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w, y_nhwc, nullptr, B, H, W,
                     Ho, Wo, sc, sh, act);
  return kd_check_launch("kd_stem_conv_fwd_infer");
}

int kd_stem_im2col(const float* x_nchw, float* col, int B, int Cin, int H, int W, int Kp, void* stream) {
  KD_REQUIRE(x_nchw && col && Kp >= Cin * 9 && Kp % 4 == 0, KD_ERR_ARG, "kd_stem_im2col: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t n = (int64_t)B * Ho * Wo * Kp;
  if (Cin == 3 && Kp == 32 && kd_aligned16(col)) {
    int64_t grid = ((int64_t)B * Ho * Wo + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(stem_im2col2_kernel<3>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, col, B, H, W, Ho, Wo);
    return kd_check_launch("kd_stem_im2col");
  }
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_nchw,
                     col, B, Cin, H, W, Ho, Wo, Kp);
  return kd_check_launch("kd_stem_im2col");
}

int64_t kd_dwconv_stat_rows(int64_t npix, int C) { return dw_layout(npix, C).rows; }

int kd_dwconv3x3_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, float* y,
                     float* partial, int B, int H, int W, int C, int stride, void* stream) {
  KD_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0, KD_ERR_ARG, "kd_dwconv3x3_fwd: bad args");
  KD_REQUIRE(C % 4 == 0 && C <= 1024 && (stride == 1 || stride == 2), KD_ERR_SHAPE, "kd_dwconv3x3_fwd: C=%d stride=%d unsupported", C, stride);
  KD_REQUIRE(kd_aligned16(x) && kd_aligned16(y), KD_ERR_ALIGN, "kd_dwconv3x3_fwd: alignment");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const DwLayout l = dw_layout((int64_t)B * Ho * Wo, C);
  DwArgs a{x, sc, sh, act, w, y, partial, B, H, W, C, Ho, Wo, stride, l.groups, l.slots, l.nchunk,
           kd_nt_store((size_t)B * Ho * Wo * C * sizeof(float))};
  static const int pipe = [] { const char* e = getenv("KD_DW_FWD_PIPE"); return e ? atoi(e) : 3; }();     // bit 0: stride 1, bit 1: stride 2
  const bool small = (int64_t)B * H * W * C < ((int64_t)1 << 31);      // the pipelined kernel addresses x by 32-bit element offsets
  if (stride == 1 && (pipe & 1) && small && Ho % 16 == 0) hipLaunchKernelGGL((dw_fwd_pipe_kernel<1, 16>), dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (stride == 1 && (pipe & 1) && small && Ho % 8 == 0) hipLaunchKernelGGL((dw_fwd_pipe_kernel<1, 8>), dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (stride == 2 && (pipe & 2) && small && Ho % 8 == 0) hipLaunchKernelGGL((dw_fwd_pipe_kernel<2, 8>), dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (stride == 1) hipLaunchKernelGGL(dw_fwd_sw_kernel<1>, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_fwd_sw_kernel<2>, dim3(l.grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_dwconv3x3_fwd");
}

// (one slab row per workgroup; the fused stride-2 kernel is launched over the INPUT pixels: at most 4 per output pixel)
size_t kd_dwconv_bwd_ws_bytes(int64_t npix_out, int C) {
  return (size_t)dw_layout(4 * npix_out, C).rows * (size_t)C * 9 * sizeof(float);
}

// Backward of y = dwconv3x3(act(x*sc+sh)).  (D, Y, al, be, ga[, dsc, dsh, d_act]) describe dL/dy_raw;
// outputs: gx (masked gradient w.r.t. the activated input, with BN-backward sums in `partial`)
// and dw [C][9].  gx == null skips the data gradient, dw == null the weight gradient.
static int dw_fused_form(int C, int W, int stride) {       // 1 column walk, 2 tile form (stride 1); the default choice by shape
  int dw_mode = kd_dw_fused_mode();
  // by shape (tools/bench_dw at 256 frames, round 3, with the channel-chunked column walk): the tile form wins on every
  // stride-1 shape of the step from 64 channels up (64: 283 vs 323 us, 128: 523 vs 573, 256: 968 vs 1054, 384: 1319 vs 1500, 768 at 32 x 32:
  // 695 vs 784); below a 64-channel chunk it idles lanes (32 channels: 810 vs 575)
  if (dw_mode == 3) dw_mode = (stride == 1 && C >= 64 && C % 64 == 0 && W >= 16) ? 2 : 1;
  return dw_mode;
}

// 1 when kd_dwconv3x3_bwd_add can fold a residual gradient into this shape's one-pass backward (the stride-1 column walk)
int kd_dwconv3x3_bwd_add_supported(int C, int W, int stride) { return stride == 1 && C % 4 == 0 && C <= 1024 && dw_fused_form(C, W, stride) == 1; }

static int dw_bwd_impl(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                       const float* dsc, const float* dsh, int d_act, const float* x, const float* sc, const float* sh,
                       int act, const float* mean, const float* invstd, const float* w, float* gx, float* partial,
                       float* dw, const float* addend, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(D && x && w && B > 0 && C % 4 == 0 && C <= 1024, KD_ERR_ARG, "kd_dwconv3x3_bwd: bad args");
  KD_REQUIRE(!addend || (gx && dw && kd_dwconv3x3_bwd_add_supported(C, W, stride) && kd_aligned16(addend)), KD_ERR_SHAPE,
             "kd_dwconv3x3_bwd_add: a residual addend needs both gradients and the stride-1 column-walk form (C=%d W=%d stride=%d)", C, W, stride);
  KD_REQUIRE(!al || (Y && be && ga), KD_ERR_ARG, "kd_dwconv3x3_bwd: al needs Y, be, ga");
  KD_REQUIRE(!sc || (sh && (!partial || (mean && invstd))), KD_ERR_ARG, "kd_dwconv3x3_bwd: sc needs sh (+mean/invstd for stats)");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  const int dw_mode = dw_fused_form(C, W, stride);
  if (gx && dw && stride == 1 && dw_mode != 0) {               // one pass: data gradient + statistics + weight-gradient partials
    const DwLayout l = dw_layout((int64_t)B * H * W, C);
    KD_REQUIRE(ws && ws_bytes >= (size_t)l.rows * C * 9 * sizeof(float), KD_ERR_WORKSPACE, "kd_dwconv3x3_bwd: workspace too small");
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, sc ? partial : nullptr,
                (float*)ws, addend, B, H, W, C, Ho, Wo, stride, l.groups, l.slots, l.nchunk, kd_nt_store((size_t)B * H * W * C * sizeof(float))};
    if (dw_mode == 2) {                     // tile form: l.rows slab rows (the row count the callers sized their slabs for) x channel chunks
      const int nchunk = (C + DT_CQ * 4 - 1) / (DT_CQ * 4);
      hipLaunchKernelGGL(dw_bwd_tile_s1_kernel, dim3((unsigned)l.rows * nchunk), dim3(256), 0, st, a, l.rows, nchunk);
    } else if (addend) {
      hipLaunchKernelGGL(dw_bwd_fused_s1_kernel<true>, dim3(l.grid), dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL(dw_bwd_fused_s1_kernel<false>, dim3(l.grid), dim3(256), 0, st, a);
    }
    int rc = kd_check_launch("kd_dwconv3x3_bwd(fused)");
    if (rc) return rc;
    return kd_slab_reduce_launch((const float*)ws, l.rows, (int64_t)C * 9, dw, st);
  }
  if (gx && dw && stride == 2 && dw_mode != 0) {               // stride 2: the same fusion over 2x2 input quads
    const DwLayout l = dw_layout((int64_t)B * H * W, C);
    KD_REQUIRE(ws && ws_bytes >= (size_t)l.rows * C * 9 * sizeof(float), KD_ERR_WORKSPACE, "kd_dwconv3x3_bwd: workspace too small");
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, sc ? partial : nullptr,
                (float*)ws, addend, B, H, W, C, Ho, Wo, stride, l.groups, l.slots, l.nchunk, kd_nt_store((size_t)B * H * W * C * sizeof(float))};
    hipLaunchKernelGGL(dw_bwd_fused_s2_kernel, dim3(l.grid), dim3(256), 0, st, a);
    int rc = kd_check_launch("kd_dwconv3x3_bwd(fused, stride 2)");
    if (rc) return rc;
    return kd_slab_reduce_launch((const float*)ws, l.rows, (int64_t)C * 9, dw, st);
  }
  if (gx) {
    const DwLayout l = dw_layout((int64_t)B * H * W, C);
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, sc ? partial : nullptr,
                nullptr, nullptr, B, H, W, C, Ho, Wo, stride, l.groups, l.slots, l.nchunk, kd_nt_store((size_t)B * H * W * C * sizeof(float))};
    if (stride == 1) hipLaunchKernelGGL(dw_bwd_data_sw_kernel, dim3(l.grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dw_bwd_data_s2_kernel, dim3(l.grid), dim3(256), 0, st, a);
    int rc = kd_check_launch("kd_dwconv3x3_bwd(data)");
    if (rc) return rc;
  }
  if (dw) {
    const DwLayout l = dw_layout((int64_t)B * Ho * Wo, C);
    KD_REQUIRE(ws && ws_bytes >= (size_t)l.rows * C * 9 * sizeof(float), KD_ERR_WORKSPACE, "kd_dwconv3x3_bwd: workspace too small");
    DwBwdArgs a{D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, nullptr, nullptr, (float*)ws, nullptr,
                B, H, W, C, Ho, Wo, stride, l.groups, l.slots, l.nchunk, 0};
    if (stride == 1) hipLaunchKernelGGL(dw_bwd_weight_sw_kernel<1>, dim3(l.grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dw_bwd_weight_sw_kernel<2>, dim3(l.grid), dim3(256), 0, st, a);
    int rc = kd_check_launch("kd_dwconv3x3_bwd(weight)");
    if (rc) return rc;
    return kd_slab_reduce_launch((const float*)ws, l.rows, (int64_t)C * 9, dw, st);
  }
  return KD_OK;
}

int kd_dwconv3x3_bwd(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                     const float* dsc, const float* dsh, int d_act, const float* x, const float* sc, const float* sh,
                     int act, const float* mean, const float* invstd, const float* w, float* gx, float* partial,
                     float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream) {
  return dw_bwd_impl(D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, partial, dw, nullptr, B, H, W, C, stride, ws,
                     ws_bytes, stream);
}

// The same backward with a second gradient path into the conv's input -- the residual of an inverted-residual block whose
// first convolution is this one (camera_encoder.py:46-51 with expansion_ratio 1) -- added to the data gradient before the
// activation mask / BatchNorm-backward sums: gx = (conv^T dy + addend) * act'(.), one kernel instead of kernel + add pass.
int kd_dwconv3x3_bwd_add(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                         const float* dsc, const float* dsh, int d_act, const float* x, const float* sc, const float* sh,
                         int act, const float* mean, const float* invstd, const float* w, const float* addend, float* gx,
                         float* partial, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(addend, KD_ERR_ARG, "kd_dwconv3x3_bwd_add: addend is NULL");
  return dw_bwd_impl(D, Y, al, be, ga, dsc, dsh, d_act, x, sc, sh, act, mean, invstd, w, gx, partial, dw, addend, B, H, W, C, stride, ws,
                     ws_bytes, stream);
}

int64_t kd_dwconv_bwd_stat_rows(int64_t npix_in, int C) { return dw_layout(npix_in, C).rows; }

// 0 separate kernels, 1 fused column walk, 2 fused tile form, 3 choose by shape (default); returns the previous mode
int kd_set_dw_bwd_mode(int mode) { (void)kd_dw_fused_mode(); return g_dw_mode.exchange(mode < 0 ? 0 : (mode > 3 ? 3 : mode)); }

}  // extern "C"
