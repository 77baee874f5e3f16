// kd_bf16.hip -- bf16-STORAGE inference path (BASELINE.json configs[1]: "camera+LiDAR concat-fusion forward ... bf16").
//
// A second, separately gated mode next to the fp32 contract: eval-mode forward only, activations live in HBM as bf16
// (NHWC, already normalised + activated: in eval mode BatchNorm coefficients are known before a convolution runs, so every
// unit is ONE kernel: conv -> fma(raw, scale, shift) -> activation (+ residual) -> bf16), accumulation in fp32, weights and
// BatchNorm coefficients stay fp32 in HBM (the GEMM rounds its weight tile to bf16 once, when it parks it in LDS).  The
// 1x1 convolutions are plain bf16 MFMA GEMMs (v_mfma_f32_32x32x16_bf16, ONE product per element instead of the six of the
// fp32-grade split arithmetic) in the weight-resident streaming form of kd_gemm_stream.hip: a lane reads its A fragment
// -- 8 consecutive bf16 of its row, 16 bytes -- straight from HBM as the MFMA operand; no conversion, no LDS for A.
// Reference layers: camera_encoder.py:19-67, fusion_module.py:8-91,162-173, lidar_encoder.py:25-35,57-99.
//
// Accuracy is that of bf16 activations (8-bit mantissa): see tests/test_gpu_bf16.py for the measured logit error and the
// argmax agreement with the fp32 path -- this mode is NOT part of the fp32 parity contract.
#include "kd_common.h"

#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef unsigned short bf16_t;                       // storage type

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {          // round to nearest even, NaN stays NaN
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void unpack8(u32x4 u, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = bf_lo(u[i]); f[2 * i + 1] = bf_hi(u[i]); }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pk_bf16(f[0], f[1]), pk_bf16(f[2], f[3]), pk_bf16(f[4], f[5]), pk_bf16(f[6], f[7])};
}
__device__ __forceinline__ u32x4 ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(bf16_t* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------------------------
// stem: 3x3 / s2 dense conv on the NCHW fp32 image, Cout == 32 (camera_encoder.py:63-67) -> act(bn(.)) as bf16 NHWC.
__global__ __launch_bounds__(256) void stem_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sc,
                                                        const float* __restrict__ sh, int act, bf16_t* __restrict__ y, int B, int Cin,
                                                        int H, int W, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int KK = Cin * 9;
  float* ws = sm;                      // [KK][32] tap-major
  for (int i = threadIdx.x; i < KK * 32; i += 256) ws[(i % KK) * 32 + i / KK] = w[i];
  __syncthreads();
  const int64_t npix = (int64_t)B * Ho * Wo;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
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
    bf16_t* yp = y + p * 32;
#pragma unroll
    for (int c = 0; c < 32; c += 8) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = kd_act(kd_affine(acc[c + j], sc[c + j], sh[c + j]), act);
      st16(yp + c, pack8(o));
    }
  }
}

// Round 3 form (as stem_fwd2_kernel in kd_conv.hip): the pixel's 9 * CIN inputs first, then one fma chain per output channel with
// the weights as wave-uniform scalar loads (no LDS, no per-fma ds_read; the first form needed 256 VGPRs + 66 AGPRs at one wave/SIMD).
template <int CIN>
__global__ __launch_bounds__(256) void stem_bf16_v2_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sc,
                                                           const float* __restrict__ sh, int act, bf16_t* __restrict__ y, int B, int H, int W,
                                                           int Ho, int Wo) {
  constexpr int KK = CIN * 9;
  // a pixel = 64 contiguous output bytes per thread: through a wave-private LDS tile (rows padded to 80 bytes) so that the wave
  // leaves four fully coalesced 1 KB stores instead of four stores that touch 64 lines each (as stem_fwd2_kernel)
  constexpr int TLD = 40;                                                      // bf16 per tile row
  __shared__ __attribute__((aligned(16))) bf16_t tiles[4 * 64 * TLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16_t* tile = tiles + wave * 64 * TLD;
  const int64_t npix = (int64_t)B * Ho * Wo;
  for (int64_t base = (int64_t)blockIdx.x * 256; base < npix; base += (int64_t)gridDim.x * 256) {
    const int64_t p_raw = base + threadIdx.x;
    const int64_t p = p_raw < npix ? p_raw : npix - 1;
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
          const float t = xp[(int64_t)(ok ? hi : 0) * W + (ok ? wi : 0)];
          v[ci * 9 + kh * 3 + kw] = ok ? t : 0.f;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 32; c += 8) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < KK; ++t) acc = fmaf(v[t], w[(c + j) * KK + t], acc);
        o[j] = kd_act(kd_affine(acc, sc[c + j], sh[c + j]), act);
      }
      st16(tile + lane * TLD + c, pack8(o));
    }
    __builtin_amdgcn_wave_barrier();                            // (LDS is in-order per wave; this only pins the compiler's order)
    const int64_t wbase = base + wave * 64;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = 16 * k + (lane >> 2), col = (lane & 3) * 8;
      const u32x4 o = ld16(tile + pix * TLD + col);
      if (wbase + pix < npix) st16(y + (wbase + pix) * 32 + col, o);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// depthwise 3x3 (stride 1 / 2, pad 1) on an ACTIVATED bf16 input; output act(conv * scale + shift) as bf16.
// A thread owns 8 channels (16 bytes) of a column segment and slides a 3-row window down it; its 72 taps in registers.
struct DwBfArgs {
  const bf16_t* x; const float* w; const float* sc; const float* sh; int act; bf16_t* y;
  int B, H, W, C, Ho, Wo, stride, groups, slots;
};
constexpr int DWB_SEG = 16;

struct Row8 { float l[8], c[8], r[8]; };
// one input row's three 16-byte column vectors as loaded (packed bf16) + which of them lie inside the image: the loads are issued
// a whole output row AHEAD of their use (round 4), the zero-select and the unpacking happen when the row enters the window
struct Raw3 { u32x4 l, c, r; bool lok, hok, rok; };

__device__ __forceinline__ Raw3 dwb_issue_row(const DwBfArgs& a, int b, int hi, int wi, int c0) {
  Raw3 o;
  o.hok = hi >= 0 && hi < a.H;
  const int hic = hi < 0 ? 0 : (hi >= a.H ? a.H - 1 : hi);
  const bf16_t* rowp = a.x + ((int64_t)b * a.H + hic) * a.W * a.C + c0;
  o.lok = o.hok && wi - 1 >= 0; o.rok = o.hok && wi + 1 < a.W;
  const int wl = wi - 1 < 0 ? 0 : wi - 1, wr = wi + 1 >= a.W ? a.W - 1 : wi + 1;
  o.l = ld16(rowp + (int64_t)wl * a.C); o.c = ld16(rowp + (int64_t)wi * a.C); o.r = ld16(rowp + (int64_t)wr * a.C);
  return o;
}
__device__ __forceinline__ void dwb_unpack_row(const Raw3& q, Row8& o) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  unpack8(q.lok ? q.l : z, o.l);
  unpack8(q.hok ? q.c : z, o.c);
  unpack8(q.rok ? q.r : z, o.r);
}

template <int STRIDE>
__global__ __launch_bounds__(256) void dw_bf16_kernel(DwBfArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = a.sc[c0 + j]; sh[j] = a.sh[c0 + j]; }
  // A thread's eight channels never change: its 72 taps live in registers for the whole launch.  (Rounds 3-4 kept them in LDS as
  // [tap][C] and read 18 x 16 B per output row: at 32 B of HBM traffic per output the stride-1 launches were bound by those reads and
  // the VALU work together -- 2.6-3.4 TB/s where the stride-2 launches, 80 B per output, ran at 4.2-4.6.)
  float wt[9][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t][j] = a.w[(c0 + j) * 9 + t];
  const int nseg = (a.Ho + DWB_SEG - 1) / DWB_SEG;
  const int64_t items = (int64_t)a.B * nseg * a.Wo;
  auto fma_row = [&](float (&acc)[8], const Row8& r, int kh) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(r.l[j], wt[kh * 3 + 0][j], fmaf(r.c[j], wt[kh * 3 + 1][j], fmaf(r.r[j], wt[kh * 3 + 2][j], acc[j])));
  };
  for (int64_t it = (int64_t)blockIdx.x * a.slots + slot; it < items; it += (int64_t)gridDim.x * a.slots) {
    const int wo = (int)(it % a.Wo), sg = (int)((it / a.Wo) % nseg), b = (int)(it / ((int64_t)a.Wo * nseg));
    const int h0 = sg * DWB_SEG, h1 = h0 + DWB_SEG < a.Ho ? h0 + DWB_SEG : a.Ho;
    const int wi = wo * STRIDE;
    // window rows of output row ho: ho*S - 1, ho*S, ho*S + 1.  Stride 1 slides by one input row per output row, stride 2 by two;
    // the NEW rows of output row ho + 1 are in flight (qa, qb) while row ho is computed and stored.
    Row8 r0, r1, r2;
    {
      const Raw3 t0 = dwb_issue_row(a, b, h0 * STRIDE - 1, wi, c0);
      dwb_unpack_row(t0, r0);
      if (STRIDE == 1) { const Raw3 t1 = dwb_issue_row(a, b, h0, wi, c0); dwb_unpack_row(t1, r1); }
    }
    Raw3 qa = dwb_issue_row(a, b, STRIDE == 1 ? h0 + 1 : 2 * h0, wi, c0);                    // stride 1: row ho + 1; stride 2: row 2 ho
    Raw3 qb = STRIDE == 2 ? dwb_issue_row(a, b, 2 * h0 + 1, wi, c0) : qa;                      // stride 2: row 2 ho + 1
    for (int ho = h0; ho < h1; ++ho) {
      const Raw3 ca = qa, cb = qb;
      // next output row's new input rows: issued before this row's arithmetic and store (a load issued after a store could not be
      // waited for without waiting for the store: in-order vmcnt); rows beyond the image are clamped and zero-selected
      qa = dwb_issue_row(a, b, STRIDE == 1 ? ho + 2 : 2 * ho + 2, wi, c0);
      if (STRIDE == 2) qb = dwb_issue_row(a, b, 2 * ho + 3, wi, c0);
      if (STRIDE == 1) dwb_unpack_row(ca, r2);
      else { dwb_unpack_row(ca, r1); dwb_unpack_row(cb, r2); }
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      fma_row(acc, r0, 0);
      fma_row(acc, r1, 1);
      fma_row(acc, r2, 2);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = kd_act(kd_affine(acc[j], sc[j], sh[j]), a.act);
      st16(a.y + (((int64_t)b * a.Ho + ho) * a.Wo + wo) * a.C + c0, pack8(acc));
      if (STRIDE == 1) { r0 = r1; r1 = r2; } else { r0 = r2; }
    }
  }
}

// Stride 1, round 4: the same arithmetic as ONE software pipeline over all the column segments of a thread.  In the kernel above every
// segment of 16 output rows starts cold (two halo rows are fetched and waited for before the first output row, and the last two
// prefetches of a segment are rows nobody uses): the stride-1 launches ran at 2.6-3.4 TB/s, the stride-2 ones (80 B per output
// against 32) at 4.2-4.6.  Here a segment is SEG + 2 row arrivals; the row two arrivals ahead is always in flight, and during the last
// two arrivals of a segment those are rows 0 and 1 of the thread's NEXT segment.  The arrival loop is fully unrolled (no control flow
// between the loads: the in-order vmcnt waits stay exact), the window rotates by renaming.  Ho must be a multiple of SEG (16 or 8).
template <int SEG>
__global__ __launch_bounds__(256) void dw_bf16_s1_pipe_kernel(DwBfArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = a.sc[c0 + j]; sh[j] = a.sh[c0 + j]; }
  float wt[9][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t][j] = a.w[(c0 + j) * 9 + t];
  const int nseg = a.Ho / SEG;
  const int64_t items = (int64_t)a.B * nseg * a.Wo, stride = (int64_t)gridDim.x * a.slots;
  int64_t it = (int64_t)blockIdx.x * a.slots + slot;
  if (it >= items) return;
  auto coords = [&](int64_t t, int& b, int& h0, int& wo) __attribute__((always_inline)) {
    wo = (int)(t % a.Wo);
    const int64_t q = t / a.Wo;
    h0 = (int)(q % nseg) * SEG;
    b = (int)(q / nseg);
  };
  int b, h0, wo;
  coords(it, b, h0, wo);
  Raw3 qa = dwb_issue_row(a, b, h0 - 1, wo, c0), qb = dwb_issue_row(a, b, h0, wo, c0);
  Row8 r0, r1, r2;
#pragma unroll
  for (int j = 0; j < 8; ++j) { r1.l[j] = r1.c[j] = r1.r[j] = 0.f; r2.l[j] = r2.c[j] = r2.r[j] = 0.f; }
  for (;;) {
    const int64_t itn = it + stride;
    const bool more = itn < items;
    int bn, h0n, won;
    coords(more ? itn : it, bn, h0n, won);
#pragma unroll
    for (int k = 0; k < SEG + 2; ++k) {                        // arrival k is input row h0 - 1 + k
      const Raw3 c = qa;
      qa = qb;
      if (k + 2 < SEG + 2) qb = dwb_issue_row(a, b, h0 + 1 + k, wo, c0);
      else qb = dwb_issue_row(a, bn, h0n - 1 + (k - SEG), won, c0);          // rows 0, 1 of the next segment
      r0 = r1; r1 = r2;
      dwb_unpack_row(c, r2);
      if (k >= 2) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const Row8& rw = kh == 0 ? r0 : (kh == 1 ? r1 : r2);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            acc[j] = fmaf(rw.l[j], wt[kh * 3 + 0][j], fmaf(rw.c[j], wt[kh * 3 + 1][j], fmaf(rw.r[j], wt[kh * 3 + 2][j], acc[j])));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = kd_act(kd_affine(acc[j], sc[j], sh[j]), a.act);
        st16(a.y + (((int64_t)b * a.Ho + (h0 + k - 2)) * a.Wo + wo) * a.C + c0, pack8(acc));
      }
    }
    if (!more) break;
    it = itn; b = bn; h0 = h0n; wo = won;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 1x1 convolution as a bf16 GEMM, weight-resident streaming form.  C[m][n] = act((A[m][:] . W[n][:] + bias[n]) * esc[n] +
// esh[n]) (+ res[m][n]).  AIN 0: A bf16 [M][lda]; 1: A fp32 [M][lda] rounded to bf16 on load (LiDAR BEV grid);
// 3: A = act0(bn0(layer0(point))) computed from the 16-byte point (LiDAR layer 0, K = 64).  EPI 0: bf16 store;
// EPI 4: BEV scatter-max of the (non-negative) result into an fp32 grid by the unsigned bit pattern (cell index per row).
struct GemmBfArgs {
  const void* A; int64_t lda;
  const float* W; const float* bias; const float* esc; const float* esh; int act;
  bf16_t* C; int64_t ldc; const bf16_t* res; int64_t ldres;
  int64_t M; int K, N;
  const int* m_dev;
  const float* l0w; const float* l0b; const float* sc0; const float* sh0; int act0;     // AIN 3
  const int* cell; float* grid; int64_t ldgrid;                                         // EPI 4
};

constexpr int BW = 8;                 // waves per workgroup

__device__ __forceinline__ float xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

#ifndef KD_BF16_PROBE
#define KD_BF16_PROBE 0   // dev builds only: timing probes, results WRONG by construction (1: no C stores, 2: no A loads)
#endif
#ifndef KD_BF16_OCC
#define KD_BF16_OCC 2
#endif
#ifndef KD_BF16_CH
#define KD_BF16_CH 8
#endif
template <int NB, int AIN, int EPI>
__global__ __launch_bounds__(64 * BW, KD_BF16_OCC) void pw_gemm_bf16_kernel(GemmBfArgs g) {
  constexpr int N = 32 * NB, CH = KD_BF16_CH;                         // CH: k-steps (of 16) per register chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Wh = reinterpret_cast<bf16_t*>(smem_raw);                   // [N][K] bf16, 16-byte chunks swizzled
  float* Co = reinterpret_cast<float*>(smem_raw + (size_t)N * g.K * 2);  // AIN 3: [7][K]
  const int K = g.K, CPR = K / 8, NU = K / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool odd = (lane & 1) != 0;
  const int n0 = blockIdx.y * N;
  auto swz = [&](int n) { return CPR % 16 == 0 ? (n & 15) : (CPR % 16 == 8 ? ((n >> 1) & 7) : ((n >> 2) & 3)); };
  for (int i = tid; i < N * K / 4; i += 64 * BW) {
    const int n = i / (K / 4), k4 = i % (K / 4);
    const float4 w = kd_ld4(g.W + (int64_t)(n0 + n) * K + k4 * 4);
    bf16_t* d = Wh + n * K + ((k4 >> 1) ^ swz(n)) * 8 + (k4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(pk_bf16(w.x, w.y), pk_bf16(w.z, w.w));
  }
  if (AIN == 3) {
    for (int k = tid; k < K; k += 64 * BW) {
      Co[k] = g.sc0[k]; Co[K + k] = g.sh0[k];
      const float4 w0 = kd_ld4(g.l0w + k * 4);
      Co[2 * K + k] = w0.x; Co[3 * K + k] = w0.y; Co[4 * K + k] = w0.z; Co[5 * K + k] = w0.w;
      Co[6 * K + k] = g.l0b[k];
    }
  }
  kd_lds_barrier();

  int64_t M = g.M;
  if (g.m_dev) { const int mv = *g.m_dev; M = mv < g.M ? mv : g.M; }
  const int64_t nslab = (M + 31) / 32;
  const int64_t wtot = (int64_t)gridDim.x * BW, wid = (int64_t)blockIdx.x * BW + wave;
  const int fsw = h ^ swz(r);

  float bias[NB], esc[NB], esh[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = n0 + 32 * j + r;
    bias[j] = g.bias ? g.bias[c] : 0.f;
    esc[j] = g.esc[c]; esh[j] = g.esh[c];
  }

  // A fragment of k-step u for this lane: 8 consecutive k of row r starting at 16 u + 8 h
  auto load_frag = [&](int64_t gm, int u) -> u32x4 {
    if (AIN == 0) {
#if KD_BF16_PROBE & 2
      return u32x4{(uint32_t)gm, (uint32_t)u, 0x3f803f80u, 0x3f803f80u};
#endif
      return ld16(reinterpret_cast<const bf16_t*>(g.A) + gm * g.lda + 16 * u + 8 * h);
    } else if (AIN == 1) {
      const float* p = reinterpret_cast<const float*>(g.A) + gm * g.lda + 16 * u + 8 * h;
      const float4 a = kd_ld4(p), b = kd_ld4(p + 4);
      return u32x4{pk_bf16(a.x, a.y), pk_bf16(a.z, a.w), pk_bf16(b.x, b.y), pk_bf16(b.z, b.w)};
    } else {
      return u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto l0_frag = [&](float4 pt, int u) -> u32x4 {                    // AIN 3: layer 0 + BatchNorm + activation of the point
    float v[8];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kb = 16 * u + 8 * h + 4 * e;
      const float4 sc = kd_ld4(Co + kb), sh = kd_ld4(Co + K + kb), wx = kd_ld4(Co + 2 * K + kb), wy = kd_ld4(Co + 3 * K + kb),
                   wz = kd_ld4(Co + 4 * K + kb), ww = kd_ld4(Co + 5 * K + kb), b0 = kd_ld4(Co + 6 * K + kb);
      v[4 * e + 0] = kd_act(kd_affine(kd_l0_raw(pt, make_float4(wx.x, wy.x, wz.x, ww.x), b0.x), sc.x, sh.x), g.act0);
      v[4 * e + 1] = kd_act(kd_affine(kd_l0_raw(pt, make_float4(wx.y, wy.y, wz.y, ww.y), b0.y), sc.y, sh.y), g.act0);
      v[4 * e + 2] = kd_act(kd_affine(kd_l0_raw(pt, make_float4(wx.z, wy.z, wz.z, ww.z), b0.z), sc.z, sh.z), g.act0);
      v[4 * e + 3] = kd_act(kd_affine(kd_l0_raw(pt, make_float4(wx.w, wy.w, wz.w, ww.w), b0.w), sc.w, sh.w), g.act0);
    }
    return pack8(v);
  };

  for (int64_t s = wid; s < nslab; s += wtot) {
    const int64_t m0 = s * 32;
    int64_t gm = m0 + r;
    gm = gm < M ? gm : M - 1;
    float4 pt = kd_zero4();
    if (AIN == 3) pt = kd_ld4(reinterpret_cast<const float*>(g.A) + gm * 4);
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    // k-loop in chunks of CH k-steps: the next chunk's fragments are requested before the current chunk is multiplied
    // (round 4: also requesting the NEXT slab's first chunk during the last chunk of this one -- no exposed latency at the top of a
    // slab -- measured 11 % SLOWER over the 18 launches of the forward, 3.70 -> 4.11 ms: not kept)
    u32x4 cur[CH], nxt[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) cur[i] = (AIN != 3 && i < NU) ? load_frag(gm, i) : u32x4{0u, 0u, 0u, 0u};
    for (int u0 = 0; u0 < NU; u0 += CH) {
#pragma unroll
      for (int i = 0; i < CH; ++i) nxt[i] = (AIN != 3 && u0 + CH + i < NU) ? load_frag(gm, u0 + CH + i) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int u = u0 + i;
        if (u < NU) {
          const u32x4 af = AIN == 3 ? l0_frag(pt, u) : cur[i];
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const u32x4 bf = *reinterpret_cast<const u32x4*>(Wh + (32 * j + r) * K + ((2 * u) ^ fsw) * 8);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) cur[i] = nxt[i];
    }
    // ---- epilogue: register q of a block is row (q & 3) + 8 (q >> 2) + 4 h, column r ------------------------------------
    if (EPI == 0) {
      // neighbouring lanes swap one value of each register pair so that a lane holds two adjacent columns of one row and
      // stores them as one dword (2 bf16): even lane -> (row R, cols c, c + 1), odd lane -> (row R + 1, cols c - 1, c)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
          const int rbase = (q & 3) + 8 * (q >> 2);
          float v0 = kd_act(kd_affine(acc[j][q] + bias[j], esc[j], esh[j]), g.act);
          float v1 = kd_act(kd_affine(acc[j][q + 1] + bias[j], esc[j], esh[j]), g.act);
          float x0 = xor1(v0), x1 = xor1(v1);
          asm volatile("" : "+v"(x0), "+v"(x1));                      // keep both DPP moves out of the divergent selects below
          float lo = odd ? x1 : v0, hi = odd ? v1 : x0;
          const int64_t row = m0 + rbase + 4 * h + (odd ? 1 : 0);
          const int col = n0 + 32 * j + (r & ~1);
#if KD_BF16_PROBE & 1
          if (row < M && lo == 123456.75f) {
#else
          if (row < M) {
#endif
            if (g.res) {
              const uint32_t rr = *reinterpret_cast<const uint32_t*>(g.res + row * g.ldres + col);
              lo += bf_lo(rr); hi += bf_hi(rr);
            }
            *reinterpret_cast<uint32_t*>(g.C + row * g.ldc + col) = pk_bf16(lo, hi);
          }
        }
      }
    } else {
      // EPI 4: scatter-max.  Rows arrive sorted by cell, and registers q = 4 g .. 4 g + 3 of a lane are 4 consecutive rows:
      // a run of equal cells inside the quad is merged in registers before it touches the grid.
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        int cellv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t row = m0 + 8 * gq + 4 * h + i;
          cellv[i] = row < M ? g.cell[row] : -1;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          float run = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = kd_act(kd_affine(acc[j][4 * gq + i] + bias[j], esc[j], esh[j]), g.act);
            const bool first = i == 0 || cellv[i] != cellv[i > 0 ? i - 1 : 0];
            const bool last = i == 3 || cellv[i] != cellv[i < 3 ? i + 1 : i];
            run = (first || v > run) ? v : run;
            if (last && cellv[i] >= 0 && run > 0.f)
              atomicMax(reinterpret_cast<unsigned*>(g.grid + (int64_t)cellv[i] * g.ldgrid + n0 + 32 * j + r), __float_as_uint(run));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 4: second form of the same GEMM for the common case (A bf16, C bf16: AIN 0 / EPI 0), written after timing probes of the first
// (tools/r4_probe_bf16.sh, -DKD_BF16_PROBE): over the 18 launches of a forward 3.59 ms, 2.03 ms without the C stores (and the residual
// loads inside their guards), 2.56 ms without the A loads -- the layers with many output columns ran at 2.9-3.0 TB/s, the residual
// layers (32 -> 32 at 4.2 M rows: 371 us, 86 without its epilogue) waited for one residual dword after the other, and the K = 32 / 64
// layers had 2-4 KB per wave in flight.  What changed:
//   * K is a template parameter (KU k-steps of 16), so the chunk structure of a slab is static and the loop bodies are branch-free;
//   * the A fragments of the NEXT unit are requested before the current unit is multiplied, across slab boundaries; a unit is one
//     chunk of eight k-steps, or for K = 32 / 64 four / two whole slabs (a "super-slab" of 128 / 64 consecutive rows): 8 KB per wave
//     in flight for every K;
//   * a residual tile is requested first thing in its unit (older than the prefetch in the in-order vmcnt queue), in the accumulator
//     layout, so that the sum is rounded to bf16 once as before;
//   * the epilogue goes through a wave-private LDS tile [32 rows][64 columns] (two column blocks per pass): dword writes in the
//     accumulator layout, 16-byte reads along rows, 16-byte global stores -- whole 128-byte lines, 4 store instructions per pass
//     instead of 16 (the fp32 streaming kernels' transposition tile, kd_gemm_stream_kernel.h);
//   * full units run without predicates; the one partial unit of a launch is handled after the loop by the wave it falls to.
// Same products, same fp32 accumulation order per output element, same single rounding: the same bits as the first form
// (tests/test_gpu_bf16.py compares the two).  KD_BF16_V2=0 keeps the first form everywhere.
template <int NB, int KU, bool RES>
__global__ __launch_bounds__(64 * BW, 2) void pw_gemm_bf16_v2_kernel(GemmBfArgs g) {
  constexpr int N = 32 * NB, K = 16 * KU, CPR = K / 8;
  constexpr int SL = KU >= 8 ? 1 : 8 / KU;             // slabs per unit
  constexpr int NCH = (KU + 7) / 8;                    // chunks of eight k-steps per slab (SL == 1)
  constexpr int TWB = NB >= 2 ? 2 : 1;                 // column blocks per epilogue pass
  constexpr int TP = 64 * TWB + 16;                    // tile row pitch, bytes
  constexpr int LPR = 4 * TWB, RPI = 64 / LPR, NST = 32 / RPI;   // 16-byte phase: lanes per row, rows per instruction, instructions per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Wh = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool odd = (lane & 1) != 0;
  const int n0 = blockIdx.y * N;
  unsigned char* tile = smem_raw + (size_t)N * K * 2 + (size_t)wave * 32 * TP;
  auto swz = [&](int n) { return CPR % 16 == 0 ? (n & 15) : (CPR % 16 == 8 ? ((n >> 1) & 7) : ((n >> 2) & 3)); };
  for (int i = tid; i < N * K / 4; i += 64 * BW) {
    const int n = i / (K / 4), k4 = i % (K / 4);
    const float4 w = kd_ld4(g.W + (int64_t)(n0 + n) * K + k4 * 4);
    bf16_t* d = Wh + n * K + ((k4 >> 1) ^ swz(n)) * 8 + (k4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = make_uint2(pk_bf16(w.x, w.y), pk_bf16(w.z, w.w));
  }
  kd_lds_barrier();

  const int M = (int)g.M;
  constexpr int UR = 32 * SL;                          // rows per unit
  const int nunit = (M + UR - 1) / UR, nfull = M / UR;
  const int wtot = (int)gridDim.x * BW, wid = (int)blockIdx.x * BW + wave;
  const int fsw = h ^ swz(r);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A);
  const int lda = (int)g.lda, ldc = (int)g.ldc, ldres = (int)g.ldres;

  float bias[NB], esc[NB], esh[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = n0 + 32 * j + r;
    bias[j] = g.bias ? g.bias[c] : 0.f;
    esc[j] = g.esc[c]; esh[j] = g.esh[c];
  }

  // the eight fragments of a unit: SL > 1: slab k, k-step u -> f[k * KU + u]; SL == 1: chunk c, k-steps 8 c .. 8 c + 7
  auto issue = [&](int S, int c, u32x4 (&f)[8]) __attribute__((always_inline)) {
    const int64_t m0 = (int64_t)S * UR;
    const bf16_t* base = A + m0 * lda + 8 * h;                       // wave-uniform + a lane term that never changes
    const int lim = (int)((int64_t)M - 1 - m0);                     // rows >= M of the partial unit repeat row M - 1
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const int rr = r + 32 * k;
      const int off = (rr < lim ? rr : lim) * lda;
#pragma unroll
      for (int i = 0; i < (SL > 1 ? KU : 8); ++i) {
        const int u = SL > 1 ? i : 8 * c + i;
        if (u < KU) f[SL > 1 ? k * KU + i : i] = ld16(base + off + 16 * u);
      }
    }
  };

  u32x4 cur[8], nxt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cur[i] = u32x4{0u, 0u, 0u, 0u}; nxt[i] = u32x4{0u, 0u, 0u, 0u}; }

  auto unit = [&](int S, auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const int64_t m0u = (int64_t)S * UR;
    uint32_t rr[RES ? SL : 1][RES ? NB : 1][8];
    if constexpr (RES) {                                             // first in the queue: the epilogue must not wait for the prefetch
#pragma unroll
      for (int k = 0; k < SL; ++k)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const bf16_t* rb = g.res + (FULL ? (m0u + 32 * k) * ldres : (int64_t)0) + n0 + 32 * j + (r & ~1);
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            int row = (q & 3) + 8 * (q >> 2) + 4 * h + (odd ? 1 : 0);
            if (!FULL) {                                             // the partial unit: absolute row, clamped to the last row of the tensor
              const int64_t ra = m0u + 32 * k + row;
              rr[k][j][q >> 1] = *reinterpret_cast<const uint32_t*>(rb + (ra < M ? ra : (int64_t)M - 1) * ldres);
            } else {
              rr[k][j][q >> 1] = *reinterpret_cast<const uint32_t*>(rb + row * ldres);
            }
          }
        }
    }
    const int Sn = S + wtot < nunit ? S + wtot : S;                  // past the end: fetch this unit again (never used)
    if constexpr (SL > 1) issue(Sn, 0, nxt);
#pragma unroll
    for (int k = 0; k < SL; ++k) {
      const int64_t m0 = m0u + 32 * k;
      f32x16 acc[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if constexpr (SL == 1) {
          if (c + 1 < NCH) issue(S, c + 1, nxt);
          else issue(Sn, 0, nxt);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int u = SL > 1 ? i : 8 * c + i;
          if ((SL > 1 ? i < KU : true) && u < KU) {
            const u32x4 af = cur[SL > 1 ? k * KU + i : i];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const u32x4 bf = *reinterpret_cast<const u32x4*>(Wh + (32 * j + r) * K + ((2 * u) ^ fsw) * 8);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
            }
          }
        }
        if constexpr (SL == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        }
        __builtin_amdgcn_sched_barrier(0);               // chunks stay in program order: the unrolled chunk loop otherwise lets hipcc hoist
      }                                                  // later chunks' B-fragment reads and run out of registers (K = 384 / 512, NB = 4)
      // ---- epilogue of slab k: TWB column blocks per pass through the tile -------------------------------------------------
#pragma unroll
      for (int p = 0; p < NB / TWB; ++p) {
#pragma unroll
        for (int jj = 0; jj < TWB; ++jj) {
          const int j = p * TWB + jj;
#pragma unroll
          for (int q = 0; q < 16; q += 2) {
            const int rbase = (q & 3) + 8 * (q >> 2);
            float v0 = kd_act(kd_affine(acc[j][q] + bias[j], esc[j], esh[j]), g.act);
            float v1 = kd_act(kd_affine(acc[j][q + 1] + bias[j], esc[j], esh[j]), g.act);
            float x0 = xor1(v0), x1 = xor1(v1);
            asm volatile("" : "+v"(x0), "+v"(x1));
            float lo = odd ? x1 : v0, hi = odd ? v1 : x0;
            if constexpr (RES) { const uint32_t t = rr[k][j][q >> 1]; lo += bf_lo(t); hi += bf_hi(t); }
            const int trow = rbase + 4 * h + (odd ? 1 : 0);
            *reinterpret_cast<uint32_t*>(tile + trow * TP + (32 * jj + (r & ~1)) * 2) = pk_bf16(lo, hi);
          }
        }
        bf16_t* cb = g.C + m0 * ldc + n0 + 32 * TWB * p;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
          const int trow = i * RPI + lane / LPR, ch = lane % LPR;
          const u32x4 v = *reinterpret_cast<const u32x4*>(tile + trow * TP + ch * 16);
          if (FULL || m0 + trow < M) st16(cb + trow * ldc + ch * 8, v);
        }
      }
    }
    if constexpr (SL > 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
  };

  int S = wid;
  if (S < nunit) issue(S, 0, cur);
  for (; S < nfull; S += wtot) unit(S, std::true_type{});
  if (S < nunit) unit(S, std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// FPN sum: out = sum_i bilinear_resize(in_i) (align_corners = False; identity when sizes match), up to 3 bf16 inputs.
struct BlBfArgs {
  const bf16_t* in[3]; int Hi[3], Wi[3]; float sh[3], sw[3]; int nin;
  bf16_t* out; int B, Ho, Wo, C, groups, slots;
};
__device__ __forceinline__ void bl_src(int o, int in_size, float scale, int& i0, int& i1, float& l0, float& l1) {
  float src = scale * ((float)o + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void bilinear_sum_bf16_kernel(BlBfArgs a) {
  const int tid = threadIdx.x;
  const int gidx = tid % a.groups, slot = tid / a.groups;
  if (slot >= a.slots) return;
  const int c0 = gidx * 8;
  const int64_t npix = (int64_t)a.B * a.Ho * a.Wo;
  for (int64_t p = (int64_t)blockIdx.x * a.slots + slot; p < npix; p += (int64_t)gridDim.x * a.slots) {
    const int wo = (int)(p % a.Wo), ho = (int)((p / a.Wo) % a.Ho), b = (int)(p / ((int64_t)a.Wo * a.Ho));
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < a.nin; ++t) {
      int h0, h1, w0, w1;
      float lh0, lh1, lw0, lw1;
      bl_src(ho, a.Hi[t], a.sh[t], h0, h1, lh0, lh1);
      bl_src(wo, a.Wi[t], a.sw[t], w0, w1, lw0, lw1);
      const bf16_t* base = a.in[t] + (int64_t)b * a.Hi[t] * a.Wi[t] * a.C + c0;
      float v00[8], v01[8], v10[8], v11[8];
      unpack8(ld16(base + ((int64_t)h0 * a.Wi[t] + w0) * a.C), v00);
      unpack8(ld16(base + ((int64_t)h0 * a.Wi[t] + w1) * a.C), v01);
      unpack8(ld16(base + ((int64_t)h1 * a.Wi[t] + w0) * a.C), v10);
      unpack8(ld16(base + ((int64_t)h1 * a.Wi[t] + w1) * a.C), v11);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += lh0 * (lw0 * v00[j] + lw1 * v01[j]) + lh1 * (lw0 * v10[j] + lw1 * v11[j]);
    }
    st16(a.out + p * a.C + c0, pack8(acc));
  }
}

// classifier: logits[b][nc][hw] = x[m][:] . w[nc][:] + bias[nc]  (x bf16 [M][Cin], Cin <= 64, NC <= 4) -> fp32 NCHW
__global__ __launch_bounds__(256) void cls_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ logits, int64_t M, int HW, int Cin, int NC) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < Cin; c += 8) {
      float v[8];
      unpack8(ld16(x + m * Cin + c), v);
      for (int nc = 0; nc < NC; ++nc)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[nc] = fmaf(v[j], w[nc * Cin + c + j], acc[nc]);
    }
    const int64_t b = m / HW, hw = m % HW;
    for (int nc = 0; nc < NC; ++nc) logits[(b * NC + nc) * HW + hw] = acc[nc] + bias[nc];
  }
}

int cg8_layout(int64_t rows, int C, int& groups, int& slots) {
  groups = C / 8;
  slots = 256 / groups;
  if (slots < 1) slots = 1;
  int64_t need = (rows + slots - 1) / slots;
  return (int)(need < 2048 ? (need < 1 ? 1 : need) : 2048);
}

// ---------------------------------------------------------------------------------------------------------------------
// weighted-fusion tail (fusion_module.py:115-120): per pixel the two attention logits a_j = h . w2[j] + b2[j] from the
// ReLU'd bf16 attention.0 output h [M][C], softmax over the two, out = w_0 * cam_proj + w_1 * lidar_proj with the two
// projections read from cat [M][2C] (bf16, already normalised + activated).  16-byte lanes: C/8 lanes share a pixel.
__global__ __launch_bounds__(256) void weighted_tail_bf16_kernel(const bf16_t* __restrict__ h, const bf16_t* __restrict__ cat, const float* __restrict__ w2,
                                                                 const float* __restrict__ b2, bf16_t* __restrict__ out, int64_t M, int C, int LP) {
  const int gidx = threadIdx.x % LP, slot = threadIdx.x / LP, slots = 256 / LP;
  const int c0 = gidx * 8;
  float w20[8], w21[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { w20[i] = w2[c0 + i]; w21[i] = w2[C + c0 + i]; }
  const float b20 = b2[0], b21 = b2[1];
  const int64_t iters = (M + (int64_t)gridDim.x * slots - 1) / ((int64_t)gridDim.x * slots);
  for (int64_t it = 0; it < iters; ++it) {               // uniform trip count: the shuffles need every lane
    const int64_t m = (it * gridDim.x + blockIdx.x) * slots + slot;
    const bool ok = m < M;
    const int64_t mm = ok ? m : M - 1;
    float hv[8], cp[8], lp[8];
    unpack8(ld16(h + mm * C + c0), hv);
    unpack8(ld16(cat + mm * 2 * C + c0), cp);
    unpack8(ld16(cat + mm * 2 * C + C + c0), lp);
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a0 = fmaf(hv[i], w20[i], a0); a1 = fmaf(hv[i], w21[i], a1); }
    for (int o = LP >> 1; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o); a1 += __shfl_xor(a1, o); }
    a0 += b20; a1 += b21;
    const float mx = fmaxf(a0, a1);
    const float e0 = expf(a0 - mx), e1 = expf(a1 - mx);
    const float w0 = e0 / (e0 + e1), w1 = e1 / (e0 + e1);
    float r[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = cp[i] * w0 + lp[i] * w1;
    if (ok) st16(out + m * C + c0, pack8(r));
  }
}

template <int NB>
int launch_gemm_bf16(GemmBfArgs& g, int ain, int epi, hipStream_t st) {
  const int ntiles = g.N / (32 * NB);
  int64_t want = (g.M + 32 * BW - 1) / (32 * BW);
  int cap = 128 * KD_BF16_OCC / ntiles; if (cap < 1) cap = 1;
  const dim3 grid((unsigned)(want < cap ? want : cap), ntiles);
  const size_t lds = (size_t)32 * NB * g.K * 2 + (ain == 3 ? (size_t)7 * g.K * 4 : 0);
#define KD_BCASE(A_, E_)                                                                                               \
  if (ain == A_ && epi == E_) {                                                                                        \
    const hipError_t le = hipFuncSetAttribute((const void*)pw_gemm_bf16_kernel<NB, A_, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    KD_REQUIRE(le == hipSuccess, (int)le, "kd_bf16_pwconv: cannot raise the dynamic LDS limit to %zu B: %s", lds, hipGetErrorString(le)); \
    hipLaunchKernelGGL((pw_gemm_bf16_kernel<NB, A_, E_>), grid, dim3(64 * BW), lds, st, g);                            \
    return kd_check_launch("kd_bf16_pwconv");                                                                          \
  }
  KD_BCASE(0, 0) KD_BCASE(1, 0) KD_BCASE(3, 0) KD_BCASE(0, 4)
#undef KD_BCASE
  kd_set_error("kd_bf16_pwconv: unsupported (input kind %d, epilogue %d)", ain, epi);
  return KD_ERR_ARG;
}

constexpr int KD_BF16_V2_NO_INSTANCE = -12345;
template <int NB, int KU>
int launch_gemm_bf16_v2_nk(GemmBfArgs& g, hipStream_t st) {
  const int ntiles = g.N / (32 * NB);
  constexpr int SL = KU >= 8 ? 1 : 8 / KU, TWB = NB >= 2 ? 2 : 1;
  const int64_t want = (g.M + 32 * SL * BW - 1) / (32 * SL * BW);
  int cap = 256 / ntiles; if (cap < 1) cap = 1;
  const dim3 grid((unsigned)(want < cap ? want : cap), ntiles);
  const size_t lds = (size_t)32 * NB * 16 * KU * 2 + (size_t)BW * 32 * (64 * TWB + 16);
  const void* fn = g.res ? (const void*)pw_gemm_bf16_v2_kernel<NB, KU, true> : (const void*)pw_gemm_bf16_v2_kernel<NB, KU, false>;
  const hipError_t le = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  KD_REQUIRE(le == hipSuccess, (int)le, "kd_bf16_pwconv: cannot raise the dynamic LDS limit to %zu B: %s", lds, hipGetErrorString(le));
  if (g.res) hipLaunchKernelGGL((pw_gemm_bf16_v2_kernel<NB, KU, true>), grid, dim3(64 * BW), lds, st, g);
  else hipLaunchKernelGGL((pw_gemm_bf16_v2_kernel<NB, KU, false>), grid, dim3(64 * BW), lds, st, g);
  return kd_check_launch("kd_bf16_pwconv");
}
template <int NB>
int launch_gemm_bf16_v2_n(GemmBfArgs& g, hipStream_t st) {
  switch (g.K / 16) {
    case 2: return launch_gemm_bf16_v2_nk<NB, 2>(g, st);
    case 4: return launch_gemm_bf16_v2_nk<NB, 4>(g, st);
    case 8: return launch_gemm_bf16_v2_nk<NB, 8>(g, st);
    case 12: return launch_gemm_bf16_v2_nk<NB, 12>(g, st);
    case 16: return launch_gemm_bf16_v2_nk<NB, 16>(g, st);
    case 24: return launch_gemm_bf16_v2_nk<NB, 24>(g, st);
    case 32: return launch_gemm_bf16_v2_nk<NB, 32>(g, st);
    case 48: return launch_gemm_bf16_v2_nk<NB, 48>(g, st);
  }
  return KD_BF16_V2_NO_INSTANCE;
}
// the second form where it has an instance: K in {32, 64, 128, 192, 256, 384, 512, 768}, 16-byte-aligned rows of C, fewer than 2^31 rows
int launch_gemm_bf16_v2(GemmBfArgs& g, hipStream_t st) {
  static const int on = [] { const char* e = getenv("KD_BF16_V2"); return e ? atoi(e) : 1; }();
  if (!on || g.K % 16 != 0 || g.M >= ((int64_t)1 << 31) || g.ldc % 8 != 0 || !kd_aligned16(g.C) || g.lda * 128 >= ((int64_t)1 << 31) ||
      g.ldc * 32 >= ((int64_t)1 << 31) || (g.res && g.ldres * 32 >= ((int64_t)1 << 31)))
    return KD_BF16_V2_NO_INSTANCE;
  const int SL = g.K >= 128 ? 1 : 128 / g.K;
  const size_t budget = 160 * 1024;
  for (int NB = 4; NB >= 1; NB >>= 1) {
    if (g.N % (32 * NB) != 0) continue;
    const size_t lds = (size_t)32 * NB * g.K * 2 + (size_t)BW * 32 * (64 * (NB >= 2 ? 2 : 1) + 16);
    if (lds > budget) continue;
    if (g.res && SL * NB * 8 > 32) continue;                         // the residual tile lives in registers for the whole unit
    if (NB == 4) return launch_gemm_bf16_v2_n<4>(g, st);
    if (NB == 2) return launch_gemm_bf16_v2_n<2>(g, st);
    return launch_gemm_bf16_v2_n<1>(g, st);
  }
  return KD_BF16_V2_NO_INSTANCE;
}

}  // namespace

extern "C" {

int kd_bf16_stem(const float* x_nchw, const float* w, const float* sc, const float* sh, int act, void* y, int B, int Cin, int H,
                 int W, int Cout, void* stream) {
  KD_REQUIRE(x_nchw && w && sc && sh && y && B > 0 && Cout == 32 && Cin >= 1 && Cin <= 4, KD_ERR_ARG, "kd_bf16_stem: bad args");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t npix = (int64_t)B * Ho * Wo;
  int64_t grid = (npix + 255) / 256;
  if (grid > 4096) grid = 4096;
  if (Cin == 3) hipLaunchKernelGGL(stem_bf16_v2_kernel<3>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nchw, w, sc, sh, act,
                                   (bf16_t*)y, B, H, W, Ho, Wo);
  else hipLaunchKernelGGL(stem_bf16_kernel, dim3((unsigned)grid), dim3(256), (size_t)Cin * 9 * 32 * sizeof(float), (hipStream_t)stream, x_nchw, w, sc,
                          sh, act, (bf16_t*)y, B, Cin, H, W, Ho, Wo);
  return kd_check_launch("kd_bf16_stem");
}

int kd_bf16_dwconv3x3(const void* x, const float* w, const float* sc, const float* sh, int act, void* y, int B, int H, int W,
                      int C, int stride, void* stream) {
  KD_REQUIRE(x && w && sc && sh && y && B > 0 && C % 8 == 0 && C <= 1024 && (stride == 1 || stride == 2), KD_ERR_ARG, "kd_bf16_dwconv3x3: bad args");
  KD_REQUIRE(kd_aligned16(x) && kd_aligned16(y), KD_ERR_ALIGN, "kd_bf16_dwconv3x3: alignment");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  int groups, slots;
  const int grid = cg8_layout((int64_t)B * ((Ho + DWB_SEG - 1) / DWB_SEG) * Wo, C, groups, slots);
  DwBfArgs a{(const bf16_t*)x, w, sc, sh, act, (bf16_t*)y, B, H, W, C, Ho, Wo, stride, groups, slots};
  static const int pipe = [] { const char* e = getenv("KD_BF16_DW_PIPE"); return e ? atoi(e) : 1; }();
  if (stride == 1 && pipe && Ho % 16 == 0) {
    const int g16 = cg8_layout((int64_t)B * (Ho / 16) * Wo, C, groups, slots);
    a.groups = groups; a.slots = slots;
    hipLaunchKernelGGL(dw_bf16_s1_pipe_kernel<16>, dim3(g16), dim3(256), 0, (hipStream_t)stream, a);
  } else if (stride == 1 && pipe && Ho % 8 == 0) {
    const int g8 = cg8_layout((int64_t)B * (Ho / 8) * Wo, C, groups, slots);
    a.groups = groups; a.slots = slots;
    hipLaunchKernelGGL(dw_bf16_s1_pipe_kernel<8>, dim3(g8), dim3(256), 0, (hipStream_t)stream, a);
  } else if (stride == 1) hipLaunchKernelGGL(dw_bf16_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(dw_bf16_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bf16_dwconv3x3");
}

/* 1x1 convolution / Conv1d(k=1) + eval BatchNorm + activation (+ residual) on bf16 activations.  a_kind 0: A bf16,
 * 1: A fp32 (rounded on load), 3: A = LiDAR points [M,4] with layer 0 (l0w [K][4], l0b, sc0, sh0, act0) recomputed.
 * epi 0: C bf16 [M][ldc] (+ res bf16); epi 4: scatter-max into the fp32 `grid` [cells][ldgrid] by `cell[m]` (rows < 0 skipped;
 * the caller zero-fills the grid; the activation must be non-negative). */
int kd_bf16_pwconv(const void* A, int64_t lda, int a_kind, const float* W, const float* bias, const float* esc, const float* esh,
                   int act, void* C, int64_t ldc, const void* res, int64_t ldres, int epi, int64_t M, int K, int N,
                   const int* m_dev, const float* l0w, const float* l0b, const float* sc0, const float* sh0, int act0,
                   const int* cell, float* grid, int64_t ldgrid, void* stream) {
  KD_REQUIRE(A && W && esc && esh && M > 0 && K >= 16 && N >= 32, KD_ERR_ARG, "kd_bf16_pwconv: bad args");
  KD_REQUIRE(K % 32 == 0 && (K / 8) % 4 == 0 && N % 32 == 0 && K <= 1024, KD_ERR_SHAPE, "kd_bf16_pwconv: K=%d must be a multiple of 32, N=%d of 32", K, N);
  KD_REQUIRE(a_kind == 3 || (lda % (a_kind == 0 ? 8 : 4) == 0 && kd_aligned16(A)), KD_ERR_ALIGN, "kd_bf16_pwconv: A alignment");
  KD_REQUIRE(epi == 4 ? (cell && grid && (act == KD_ACT_RELU || act == KD_ACT_RELU6)) : (C != nullptr && ldc % 2 == 0 && (!res || ldres % 2 == 0)), KD_ERR_ARG,
             "kd_bf16_pwconv: epilogue arguments");
  KD_REQUIRE(a_kind != 3 || (l0w && l0b && sc0 && sh0 && K % 4 == 0), KD_ERR_ARG, "kd_bf16_pwconv: layer-0 arguments");
  GemmBfArgs g{A, lda, W, bias, esc, esh, act, (bf16_t*)C, ldc, (const bf16_t*)res, ldres, M, K, N, m_dev, l0w, l0b, sc0, sh0, act0,
               cell, grid, ldgrid};
  hipStream_t st = (hipStream_t)stream;
  if (a_kind == 0 && epi == 0 && !m_dev) {
    const int rc = launch_gemm_bf16_v2(g, st);
    if (rc != KD_BF16_V2_NO_INSTANCE) return rc;
  }
  // widest column tile whose W image (32 NB x K bf16) fits beside the coefficient tables in 96 KB of LDS
  const size_t budget = 96 * 1024;
  if (N % 128 == 0 && (size_t)128 * K * 2 <= budget) return launch_gemm_bf16<4>(g, a_kind, epi, st);
  if (N % 64 == 0 && (size_t)64 * K * 2 <= budget) return launch_gemm_bf16<2>(g, a_kind, epi, st);
  return launch_gemm_bf16<1>(g, a_kind, epi, st);
}

int kd_bf16_bilinear_sum(const void* in0, int H0, int W0, const void* in1, int H1, int W1, const void* in2, int H2, int W2,
                         void* out, int B, int Ho, int Wo, int C, void* stream) {
  KD_REQUIRE(in0 && out && B > 0 && C % 8 == 0 && C <= 2048, KD_ERR_ARG, "kd_bf16_bilinear_sum: bad args");
  BlBfArgs a{};
  const void* ins[3] = {in0, in1, in2};
  const int hs[3] = {H0, H1, H2}, wsz[3] = {W0, W1, W2};
  a.nin = 0;
  for (int t = 0; t < 3; ++t)
    if (ins[t]) {
      a.in[a.nin] = (const bf16_t*)ins[t]; a.Hi[a.nin] = hs[t]; a.Wi[a.nin] = wsz[t];
      a.sh[a.nin] = (float)hs[t] / (float)Ho; a.sw[a.nin] = (float)wsz[t] / (float)Wo;
      ++a.nin;
    }
  a.out = (bf16_t*)out; a.B = B; a.Ho = Ho; a.Wo = Wo; a.C = C;
  const int grid = cg8_layout((int64_t)B * Ho * Wo, C, a.groups, a.slots);
  hipLaunchKernelGGL(bilinear_sum_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return kd_check_launch("kd_bf16_bilinear_sum");
}

int kd_bf16_cls_conv(const void* x, const float* w, const float* b, float* logits_nchw, int64_t M, int HW, int Cin, int NC, void* stream) {
  KD_REQUIRE(x && w && b && logits_nchw && M > 0 && Cin % 8 == 0 && Cin <= 64 && NC >= 1 && NC <= 4, KD_ERR_ARG, "kd_bf16_cls_conv: bad args");
  int64_t grid = (M + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(cls_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, b, logits_nchw, M, HW, Cin, NC);
  return kd_check_launch("kd_bf16_cls_conv");
}

/* weighted-fusion tail: out [M][C] bf16 = softmax_2(h . w2^T + b2)-weighted sum of the two halves of cat [M][2C] (bf16).
 * h [M][C] bf16 is the ReLU'd attention.0 output; w2 [2][C], b2 [2] fp32. */
int kd_bf16_weighted_tail(const void* h, const void* cat, const float* w2, const float* b2, void* out, int64_t M, int C, void* stream) {
  KD_REQUIRE(h && cat && w2 && b2 && out && M > 0, KD_ERR_ARG, "kd_bf16_weighted_tail: bad args");
  KD_REQUIRE(C == 64 || C == 128 || C == 256 || C == 512, KD_ERR_SHAPE, "kd_bf16_weighted_tail: C=%d must be 64/128/256/512", C);
  KD_REQUIRE(kd_aligned16(h) && kd_aligned16(cat) && kd_aligned16(out), KD_ERR_ALIGN, "kd_bf16_weighted_tail: alignment");
  const int LP = C / 8, slots = 256 / LP;
  int64_t grid = (M + slots - 1) / slots;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(weighted_tail_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, (const bf16_t*)cat, w2, b2,
                     (bf16_t*)out, M, C, LP);
  return kd_check_launch("kd_bf16_weighted_tail");
}

}  // extern "C"
