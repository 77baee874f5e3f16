// kd_wgrad_rs.hip -- the weight gradient of a 1x1 convolution, dW[n][k] = sum_m Deff[m][n] * Aeff[m][k], with ROLE-SPECIALISED
// waves (round 4; the structure kd_lidar_bwd.hip proved on the point MLP).  Reference: ATen convolution_backward's weight
// gradient reached from camera_encoder.py:24,39 and fusion_module.py:12,29,116.
//
// pw_wgrad_kernel (kd_gemm.hip) cuts dW into 64/128-wide output tiles and gives every tile its own workgroups: a layer with six
// tiles reads AND converts the narrow operand six times, every 32-row chunk costs two barriers, and all four waves of a
// workgroup alternate between loading / converting / storing and multiplying (profiles/r03_wgrad_probes.txt: 57 % of the
// family's time was that per-chunk skeleton).  Here one 512-thread workgroup per CU owns a whole [32*NBLK, 32*KBLK] block of
// dW (the full matrix for most layers, a column slice of the wide operand for the 768-wide ones) and walks a slice of the rows:
//   waves 0-3 "vector": all global loads (two register sets: the chunk after next is in flight while the next one is
//     converted), the operand transforms (BatchNorm-backward affine of (G, X), deferred BatchNorm + activation of A), the
//     bf16x3 split, the 8-byte plane stores into the OTHER half of a double-buffered LDS image;
//   waves 4-7 "matrix": transposing fragment reads (ds_read_b64_tr_b16: the reduction index of this GEMM is the row) and
//     MFMAs only; wave (wn, wk) keeps its TNW x TKW accumulator tiles in registers for the whole launch.
// One LDS barrier per chunk of 16 or 32 rows; every operand is read from HBM and converted once per column slice.  Partial
// results of the row slices go to a slab [slices][N][K] summed in fixed order by the reduce kernel (deterministic).
//
// Arithmetic: the pieces and the six piece products per 16-row step of pw_wgrad_kernel<.., SPLIT> (same order inside a step);
// the steps of a row slice are summed in order, slices in slab order -- another (fixed) summation order than the tiled kernel's.
#include "kd_gemm_args.h"

#include <atomic>
#include <cstdlib>

int kd_gemm_split_mode();     // kd_gemm.hip

namespace {

constexpr int RS_THREADS = 512;

__host__ __device__ constexpr int rs_pad(int blk) { return (blk & 1) ? 64 : 32; }   // bf16 of row padding: the four rows of a transposing read fall into four different 64-byte bank quarters
__host__ __device__ constexpr int rs_pitch(int blk) { return 32 * blk + rs_pad(blk); }
// power-of-two rows per pass of the vector waves over a tensor that is `cg` float4 groups wide (256 vector lanes)
__host__ __device__ constexpr int rs_rpp(int cg, int rows) {
  int r = 256 / cg;
  int p = 1;
  while (p * 2 <= r && p * 2 <= rows) p *= 2;
  return p;
}

struct RsGeom {               // runtime part of the configuration (the template carries the per-wave tile shape)
  int WN, WK;                 // matrix-wave grid: WN * WK <= 4 waves busy
  int ncs, split_n;           // column slices of the wide operand (1 = none); split_n: the slices cut N (else K)
  int nrs;                    // row slices
  int rows_per_slice;         // multiple of the chunk height
};

// TNW x TKW: 32x32 accumulator tiles per matrix wave; NBLK = TNW * WN, KBLK = TKW * WK; CHK: 16-row steps per chunk
template <int TNW, int TKW, int WNc, int WKc, int CHK, int DMODE, int AMODE>
__global__ __launch_bounds__(RS_THREADS, 1) void pw_wgrad_rs_kernel(WgradArgs g, RsGeom q) {
  constexpr int NBLK = TNW * WNc, KBLK = TKW * WKc;
  constexpr int CH = 16 * CHK;                                   // rows per chunk
  constexpr int PD = rs_pitch(NBLK), PA = rs_pitch(KBLK);        // bf16 per LDS row
  constexpr int PLD = CH * PD, PLA = CH * PA;                    // bf16 per plane
  constexpr int BUF = 3 * (PLD + PLA);                           // bf16 per buffer
  constexpr int CGD = 8 * NBLK, CGA = 8 * KBLK;                  // float4 column groups
  constexpr int RPD = rs_rpp(CGD, CH), RPA = rs_rpp(CGA, CH);    // rows per pass
  constexpr int ND = CH / RPD, NA = CH / RPA;                    // float4 per lane per chunk and tensor
  constexpr int NDT = DMODE == 2 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block order (speed only): blocks b, b + 8, ... share an L2 and take the column slices of ONE row slice
  int cs, rsl;
  {
    const int grp = 8 * q.ncs, g0 = (int)blockIdx.x / grp, r = (int)blockIdx.x % grp;
    if ((g0 + 1) * 8 <= q.nrs) { rsl = g0 * 8 + (r & 7); cs = r >> 3; }
    else { const int rem = q.nrs - g0 * 8; rsl = g0 * 8 + r % rem; cs = r / rem; }
  }
  const int n0 = q.split_n ? cs * 32 * NBLK : 0, k0 = q.split_n ? 0 : cs * 32 * KBLK;
  const int64_t mbeg = (int64_t)rsl * q.rows_per_slice;
  int64_t mend = mbeg + q.rows_per_slice;
  if (mend > g.M) mend = g.M;
  const int nchunk = mbeg < mend ? (int)((mend - mbeg + CH - 1) / CH) : 0;
  const int nstep = (nchunk + 1) & ~1;                           // both roles run an even number of steps (two register sets)

  if (wave < 4) {
    // ================= vector waves =================
    // lane -> (column group, first row).  Where a pass does not fill the 256 lanes (48 / 96 column groups) the surplus lanes
    // repeat the work of other lanes -- same loads, same values to the same LDS addresses -- rather than idle behind a branch
    // (control flow between memory operations makes the waitcnt pass give up its exact in-order counts)
    constexpr int ACTD = CGD * RPD, ACTA = CGA * RPA;
    static_assert(ACTD >= 128 && ACTA >= 128 && ACTD <= 256 && ACTA <= 256, "lane map");
    const int td = tid < ACTD ? tid : tid - (256 - ACTD), ta = tid < ACTA ? tid : tid - (256 - ACTA);
    const int dcg = td % CGD, drp = td / CGD;
    const int acg = ta % CGA, arp = ta / CGA;
    const int gn = n0 + 4 * dcg, gk = k0 + 4 * acg;
    float4 cd[DMODE == 2 ? 5 : 1], ca[AMODE == 1 ? 2 : 1];
    if (DMODE == 2) {
      cd[0] = kd_ld4(g.al + gn); cd[1] = kd_ld4(g.be + gn); cd[2] = kd_ld4(g.ga + gn);
      cd[3] = kd_ld4(g.msc + gn); cd[4] = kd_ld4(g.msh + gn);
    }
    if (AMODE == 1) { ca[0] = kd_ld4(g.asc + gk); ca[1] = kd_ld4(g.ash + gk); }
    // Addresses are (wave-uniform 64-bit chunk base: SALU) + (32-bit per-lane offset inside the chunk): per-row 64-bit
    // offsets, being loop-invariant, would otherwise be hoisted and held in ~2 registers per load for the whole launch
    const float* Dp = g.D + n0;
    const float* Xp = DMODE == 2 ? g.X + n0 : nullptr;
    const float* Ap = g.A + k0;
    const int ldd_ = (int)g.ldd, ldx_ = (int)g.ldx, lda_ = (int)g.lda;
    float4 rd[2][NDT][ND], ra[2][NA];
    // chunk c of the slice (clamped to the last one: a chunk beyond the end converts to all-zero rows)
    auto issue = [&](int c, float4 (&sd)[NDT][ND], float4 (&sa)[NA]) __attribute__((always_inline)) {
      const int cc = c < nchunk ? c : (nchunk > 0 ? nchunk - 1 : 0);
      const int64_t m0 = mbeg + (int64_t)cc * CH;
      const int lastrow = (int)(mend - 1 - m0);                  // >= CH - 1 except in the slice's tail chunk
      const float* db = Dp + m0 * g.ldd;
      const float* xb = DMODE == 2 ? Xp + m0 * g.ldx : nullptr;
      const float* ab = Ap + m0 * g.lda;
#pragma unroll
      for (int i = 0; i < ND; ++i) {
        const int r = drp + RPD * i, rr = r < lastrow ? r : lastrow;
        sd[0][i] = kd_ld4(db + (unsigned)(rr * ldd_ + 4 * dcg));
        if (DMODE == 2) sd[1][i] = kd_ld4(xb + (unsigned)(rr * ldx_ + 4 * dcg));
      }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int r = arp + RPA * i, rr = r < lastrow ? r : lastrow;
        sa[i] = kd_ld4(ab + (unsigned)(rr * lda_ + 4 * acg));
      }
    };
    auto split_store = [&](float4 v, unsigned short* d, int plane) __attribute__((always_inline)) {
      uint2 hi, mid, lo;
      kd_split3(v, hi, mid, lo);
      *reinterpret_cast<uint2*>(d) = hi;
      *reinterpret_cast<uint2*>(d + plane) = mid;
      *reinterpret_cast<uint2*>(d + 2 * plane) = lo;
    };
    auto convert = [&](int c, const float4 (&sd)[NDT][ND], const float4 (&sa)[NA], int bufi) __attribute__((always_inline)) {
      unsigned short* bd = lds + bufi * BUF;
      unsigned short* ba = bd + 3 * PLD;
      const int64_t m0 = mbeg + (int64_t)c * CH;                 // (c >= nchunk: every row fails the test below -> zeros)
      {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int row = drp + RPD * i;
          float4 v = sd[0][i];
          if (DMODE == 2) {
            const float4 x = sd[1][i];
            v.x = kd_bwd_operand(v.x, x.x, cd[0].x, cd[1].x, cd[2].x, cd[3].x, cd[4].x, g.d_act);
            v.y = kd_bwd_operand(v.y, x.y, cd[0].y, cd[1].y, cd[2].y, cd[3].y, cd[4].y, g.d_act);
            v.z = kd_bwd_operand(v.z, x.z, cd[0].z, cd[1].z, cd[2].z, cd[3].z, cd[4].z, g.d_act);
            v.w = kd_bwd_operand(v.w, x.w, cd[0].w, cd[1].w, cd[2].w, cd[3].w, cd[4].w, g.d_act);
          }
          const bool ok = c < nchunk && m0 + row < mend;
          v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
          split_store(v, bd + row * PD + 4 * dcg, PLD);
          // eight rows per lane (96 column groups): left alone, hipcc interleaves all eight transform + split chains and their
          // temporaries push the two register sets over 256 (22 spills); two rows at a time keep pairs of independent chains
          if (ND >= 8 && (i & 1)) __builtin_amdgcn_sched_barrier(0);
        }
      }
      {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const int row = arp + RPA * i;
          float4 v = sa[i];
          if (AMODE == 1) v = kd_affine_act4(v, ca[0], ca[1], g.a_act);
          const bool ok = c < nchunk && m0 + row < mend;
          v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
          split_store(v, ba + row * PA + 4 * acg, PLA);
        }
      }
    };
    // prologue: chunks 0 and 1 in flight, chunk 0 converted, chunk 2 issued into the freed set
    issue(0, rd[0], ra[0]);
    issue(1, rd[1], ra[1]);
    convert(0, rd[0], ra[0], 0);
    issue(2, rd[0], ra[0]);
    kd_lds_barrier();
    // step s: the matrix waves multiply chunk s (buffer s & 1); convert chunk s + 1 into the other buffer, then refill its set
    // with chunk s + 3.  No branches between memory operations (in-order vmcnt: the compiler's counted waits stay exact).
    for (int s = 0; s < nstep; s += 2) {
      convert(s + 1, rd[1], ra[1], 1);
      issue(s + 3, rd[1], ra[1]);
      kd_lds_barrier();
      convert(s + 2, rd[0], ra[0], 0);
      issue(s + 4, rd[0], ra[0]);
      kd_lds_barrier();
    }
  } else {
    // ================= matrix waves =================
    const int j = wave - 4;
    const int wn = j / WKc, wk = j % WKc;
    const bool busy = j < WNc * WKc;
    f32x16 acc[TNW][TKW];
#pragma unroll
    for (int a = 0; a < TNW; ++a)
#pragma unroll
      for (int b = 0; b < TKW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    constexpr int PAo[6] = {0, 2, 1, 0, 1, 0}, PBo[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (as pw_gemm_kernel)
    kd_lds_barrier();                                                         // (the prologue's)
    for (int s = 0; s < nstep; ++s) {
      if (busy) {
        const unsigned short* bd = lds + (s & 1) * BUF;
        const unsigned short* ba = bd + 3 * PLD;
#pragma unroll
        for (int ks = 0; ks < CHK; ++ks) {
          // the D fragments of the wave's n-blocks stay for the step; the A fragments come one k-block at a time (three planes =
          // 12 registers live instead of 12 * TKW).  Each accumulator tile still receives its six products in the PAo / PBo order.
          bf16x8 d[TNW][3];
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < TNW; ++i) d[i][p] = kd_tr_frag(bd + p * PLD, PD, 16 * ks, (wn * TNW + i) * 32, lane);
#pragma unroll
          for (int ki = 0; ki < TKW; ++ki) {
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = kd_tr_frag(ba + p * PLA, PA, 16 * ks, (wk * TKW + ki) * 32, lane);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
              for (int ni = 0; ni < TNW; ++ni)
                acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[ni][PAo[t]], a[PBo[t]], acc[ni][ki], 0, 0, 0);
          }
        }
      }
      kd_lds_barrier();
    }
    if (busy) {
      float* out = g.slab + (int64_t)rsl * g.N * g.K;
#pragma unroll
      for (int ni = 0; ni < TNW; ++ni)
#pragma unroll
        for (int ki = 0; ki < TKW; ++ki) {
          const int col = k0 + (wk * TKW + ki) * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = n0 + (wn * TNW + ni) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            out[(int64_t)row * g.K + col] = acc[ni][ki][r];
          }
        }
    }
  }
}

std::atomic<int> g_rs_on{-1};       // 0 off, 1 the layers where it measured faster (default), 2 every layer with an instance
int rs_mode() {
  int v = g_rs_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("KD_WGRAD_RS");
    v = (e && e[0] == '0') ? 0 : ((e && e[0] == 'a') ? 2 : 1);
    g_rs_on.store(v, std::memory_order_relaxed);
  }
  return v;
}

bool rs_wide12() {
  // twelve k-blocks per matrix wave for 128 x 384 (one slice) and 128 x 768 (two slices): the narrow operand is converted once / twice
  // instead of twice / four times.  Measured at 256 frames: 220 -> 192 us and 418 -> 364 us (profiles/r04_wgrad_rs_ab.txt); =0 restores six.
  static const bool on = [] { const char* e = getenv("KD_WGRAD_RS_WIDE12"); return !(e && e[0] == '0'); }();
  return on;
}

// ---- configuration table ------------------------------------------------------------------------------------------------
struct RsPlan { int tnw, tkw, wn, wk, chk, ncs, split_n; };

// (N / 32, K / 32) -> plan; false: not covered (tiny layers, odd widths: the tiled kernel serves them)
bool rs_plan(int N, int K, RsPlan& p, bool any = false) {
  if (N % 32 || K % 32) return false;
  int nb = N / 32, kb = K / 32, ncs = 1, split_n = 0;
  // the 768-wide operand is cut into column slices so that a workgroup's block of dW fits its accumulator registers
  if (nb == 24) { nb = 8; ncs = 3; split_n = 1; }
  else if (kb == 24 && nb == 4 && rs_wide12()) { kb = 12; ncs = 2; }
  else if (kb == 24) { kb = 6; ncs = 4; }
  else if (kb == 12 && nb == 4 && rs_wide12()) { }
  else if (kb == 12 && nb <= 4) { kb = 6; ncs = 2; }
  else if (nb == 8 && kb == 8) { nb = 4; ncs = 2; split_n = 1; }
  struct E { int nb, kb, tnw, tkw, wn, wk, chk; };
  static const E tab[] = {
      {6, 1, 3, 1, 2, 1, 2},    // 192 x 32   stage-2 expand
      {12, 2, 3, 2, 4, 1, 1},   // 384 x 64   stage-3 / 4 expand
      {8, 4, 2, 4, 4, 1, 1},    // 768 x 128  stage-5 expand (3 column slices of 256)
      {2, 6, 1, 3, 2, 2, 2},    // 64 x 192 stage-2 project; 64 x 384 stage-3 project (2 slices)
      {4, 6, 1, 6, 4, 1, 2},    // 128 x 384 stage-4 project (2 slices); 128 x 768 stage-5 project (4 slices)
      {4, 12, 1, 12, 4, 1, 1},  // 128 x 384 stage-4 project (one slice); 128 x 768 stage-5 project (2 slices): twelve k-blocks per wave (default)
      {4, 4, 2, 2, 2, 2, 2},    // 128 x 128  FPN laterals / post, fusion projections
      {4, 2, 2, 1, 2, 2, 2},    // 128 x 64   FPN lateral of stage 3
      {2, 4, 1, 2, 2, 2, 2},    // 64 x 128   head block 0
      {4, 8, 2, 4, 2, 2, 1},    // 128 x 256  attention conv; 256 x 256 concat fuse (2 slices)
      {2, 8, 1, 4, 2, 2, 2},    // 64 x 256   head block 0 (concat)
  };
  // Measured per layer at 256 frames against pw_wgrad_kernel (tools/bench_wgrad.py, profiles/r04_wgrad_rs_ab.txt): x1.28 (192 x 32),
  // x1.08-1.10 (384 x 64, 768 x 128, 64 x 384), x1.20 (128 x 384), x1.02 (128 x 768), x1.04 (128 x 256), x1.13 (256 x 256); it LOSES
  // on 64 x 192 (x0.95), 128 x 128 (x0.91), 64 x 256 (x0.97) and ties on 64 x 128 / 128 x 64 -- those stay on the tiled kernel
  // unless kd_set_wgrad_rs(2) / KD_WGRAD_RS=all (tests exercise every instance that way).
  const bool all = any || rs_mode() == 2;
  for (const E& e : tab)
    if (e.nb == nb && e.kb == kb) {
      const bool wins = (nb == 4 && kb == 12) || (nb == 6 && kb == 1) || (nb == 12 && kb == 2) || (nb == 8 && kb == 4) || (nb == 2 && kb == 6 && ncs == 2) ||
                        (nb == 4 && kb == 6) || (nb == 4 && kb == 8);
      if (!wins && !all) return false;
      p = {e.tnw, e.tkw, e.wn, e.wk, e.chk, ncs, split_n};
      return true;
    }
  return false;
}

template <int TNW, int TKW, int WN, int WK, int CHK>
int rs_launch_shape(const WgradArgs& g, const RsGeom& q, hipStream_t st) {
  constexpr int NBLK = TNW * WN, KBLK = TKW * WK;
  constexpr size_t lds = (size_t)2 * 3 * 16 * CHK * (rs_pitch(NBLK) + rs_pitch(KBLK)) * 2;
  static_assert(lds <= 160 * 1024, "the double-buffered plane image must fit the LDS");
  const dim3 grid((unsigned)(q.nrs * q.ncs)), blk(RS_THREADS);
#define KD_RS_CASE(DM_, AM_)                                                                                              \
  if (g.d_mode == DM_ && g.a_mode == AM_) {                                                                               \
    static std::atomic<uint64_t> raised{0};                                                                               \
    const hipError_t e = kd_raise_dynamic_lds((const void*)pw_wgrad_rs_kernel<TNW, TKW, WN, WK, CHK, DM_, AM_>, lds, raised); \
    if (e != hipSuccess) { kd_set_error("kd_wgrad_rs: cannot raise the dynamic LDS limit to %zu B: %s", lds, hipGetErrorString(e)); return -1; } \
    hipLaunchKernelGGL((pw_wgrad_rs_kernel<TNW, TKW, WN, WK, CHK, DM_, AM_>), grid, blk, lds, st, g, q);                  \
    return 1;                                                                                                             \
  }
  KD_RS_CASE(0, 0) KD_RS_CASE(0, 1) KD_RS_CASE(2, 0) KD_RS_CASE(2, 1)
#undef KD_RS_CASE
  return 0;
}

bool rs_enabled() { return rs_mode() != 0; }

int rs_slices(int64_t M, const RsPlan& p) {
  const int ch = 16 * p.chk;
  int nrs = 256 / p.ncs;                                         // one workgroup per CU
  const int64_t chunks = (M + ch - 1) / ch;
  if (nrs > chunks / 8) nrs = (int)(chunks / 8);                 // small batches: at least eight chunks per slice (fewer slab rows to reduce)
  return nrs < 1 ? 1 : nrs;
}

}  // namespace

extern "C" int kd_set_wgrad_rs(int mode) { const int prev = rs_mode(); g_rs_on.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed); return prev; }

// (sized for the role-specialised form whenever the layer HAS one, whatever the switches say now: a workspace allocated before a
// switch flips must still fit)
size_t kd_wgrad_rs_ws_bytes(int64_t M, int N, int K) {
  RsPlan p;
  if (!rs_plan(N, K, p, true)) return 0;
  return (size_t)rs_slices(M, p) * (size_t)N * (size_t)K * sizeof(float);
}

int kd_wgrad_rs_launch(const WgradArgs& g0, size_t ws_bytes, float* dW, hipStream_t st) {
  RsPlan p;
  if (!rs_enabled() || !kd_gemm_split_mode() || (g0.d_mode != 0 && g0.d_mode != 2) || g0.a_mode > 1 || !rs_plan(g0.N, g0.K, p)) return 0;
  WgradArgs g = g0;
  const int ch = 16 * p.chk;
  int nrs = rs_slices(g.M, p);
  const size_t cap = ws_bytes / ((size_t)g.N * g.K * sizeof(float));
  if ((size_t)nrs > cap) nrs = (int)cap;
  if (nrs < 1) return 0;
  const int64_t chunks = ((int64_t)g.M + ch - 1) / ch;
  const int64_t cps = (chunks + nrs - 1) / nrs;
  RsGeom q{p.wn, p.wk, p.ncs, p.split_n, 0, (int)(cps * ch)};
  q.nrs = (int)(((int64_t)g.M + q.rows_per_slice - 1) / q.rows_per_slice);
  int rc = 0;
#define KD_RS_SHAPE(A_, B_, C_, D_, E_) if (p.tnw == A_ && p.tkw == B_ && p.wn == C_ && p.wk == D_ && p.chk == E_) rc = rs_launch_shape<A_, B_, C_, D_, E_>(g, q, st);
  KD_RS_SHAPE(3, 1, 2, 1, 2) KD_RS_SHAPE(3, 2, 4, 1, 1) KD_RS_SHAPE(2, 4, 4, 1, 1) KD_RS_SHAPE(1, 3, 2, 2, 2) KD_RS_SHAPE(1, 6, 4, 1, 2)
  KD_RS_SHAPE(1, 12, 4, 1, 1)
  KD_RS_SHAPE(2, 2, 2, 2, 2) KD_RS_SHAPE(2, 1, 2, 2, 2) KD_RS_SHAPE(1, 2, 2, 2, 2) KD_RS_SHAPE(2, 4, 2, 2, 1) KD_RS_SHAPE(1, 4, 2, 2, 2)
#undef KD_RS_SHAPE
  if (rc <= 0) return rc;
  const int e = kd_check_launch("kd_wgrad_rs");
  if (e) return -1;
  return kd_slab_reduce_launch(g.slab, q.nrs, (int64_t)g.N * g.K, dW, st) == KD_OK ? 1 : -1;
}
