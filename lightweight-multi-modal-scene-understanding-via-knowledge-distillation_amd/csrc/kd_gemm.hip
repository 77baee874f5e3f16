// kd_gemm.hip -- pointwise (1x1) convolution as exact-fp32 MFMA GEMMs on gfx950.
//
// Every dense conv on the hot path except the 3->32 stem is a 1x1 conv, i.e. a row-major GEMM
// over NHWC activations:
//     fwd   : Y[M,N]  = Aeff[M,K] . W[N,K]^T (+bias)           reference: camera_encoder.py:24,39,
//     dgrad : dA[M,K] = dYeff[M,N] . Wt[K,N]^T   (same kernel)            fusion_module.py:12,29,116
//     wgrad : dW[N,K] = dYeff[M,N]^T . Aeff[M,K]  (split over M)          lidar_encoder.py:29,32
// "eff" operands are computed on load from raw tensors (deferred BN+activation, or the BN-backward
// affine of (G, X)), so no normalised / activated tensor is ever written to HBM.
//
// MFMA: v_mfma_f32_32x32x2_f32 -- fp32 in, fp32 accumulate, bit-exact fmaf chain, 157 TFLOP/s
// dense peak on MI355X.  Tiles: 128x128 (or 256x64) per 256-thread workgroup, 4 waves x (2x2) 32x32 MFMA tiles.
#include "kd_gemm_args.h"

#include <atomic>

namespace {

std::atomic<int> g_gemm_split{0};

// Dev build only (-DKD_DBG_PHASES [-DKD_DBG_EPI]): per-phase clock64() totals of wave 0 of every workgroup, read back by
// tools/bench_gemm through kd_dbg_read -- how the epilogue store drains and the per-K-tile serialisation were found.
#ifdef KD_DBG_PHASES
__device__ unsigned long long kd_dbg_counters[8];
#define KD_STAMP(i) do { const long long t_ = clock64(); ph_acc[i] += t_ - ph_t; ph_t = t_; } while (0)
#ifdef KD_DBG_EPI          // bins: 0 barrier | 1 MFMA tail + staging writes | 2 barrier | 3 row loop half 0 | 4 everything before the epilogue | 5 row loop half 1 + stats
#define KD_PH(i) KD_STAMP(((i) == 5) ? 5 : 4)
#define KD_PHE(i) KD_STAMP(i)
#else
#define KD_PH(i) KD_STAMP(i)
#define KD_PHE(i) do {} while (0)
#endif
#else
#define KD_PH(i) do {} while (0)
#define KD_PHE(i) do {} while (0)
#endif


constexpr int BM = 128, BK = 32, LDSLD = 36;   // BM: slab-row granularity; 36-float LDS rows: ds_read_b128 conflict-free

// Tile shape: WM x WN waves of 64x64 each (WM*WN == 4): 128x128 for wide outputs, 256x64 when the
// last (or only) column tile would be at most 64 wide (N = 32, 64, 192, ...) so no MFMA work is
// wasted on padding columns.
//
// SPLIT: the same GEMM on the bf16 matrix pipe.  Every fp32 operand element is cut into three bf16 pieces by
// round-to-nearest (x = hi + mid + lo to 2^-26 relative) when its tile is written to LDS, and a product is the six
// leading piece products
//     hi*hi + (hi*mid + mid*hi) + (hi*lo + lo*hi + mid*mid),      dropped: mid*lo + lo*mid + lo*lo <= 2^-25 |x||y|,
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (32 cycles for K = 16
// against 8 x 64 cycles of v_mfma_f32_32x32x2_f32): 6 x 32 = 192 cycles per 32x32x16 block instead of 512, so the
// near-ridge shapes of this network (AI 14..64 FLOP/B) become HBM-bound instead of matrix-pipe-bound.
// (LDS image of the split operands: see SPROW below.)
// Split LDS image: [plane][row][32 bf16], rows of exactly 64 bytes, no padding, with the four 16-byte chunks of a
// row XOR-swizzled by (row >> 2) & 3.  ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (and
// the same +32): the four rows of a group that share row % 4 then land in four different chunks, so the 16 lanes tile
// the 64 banks exactly once; a 16-lane ds_write_b64 group covers two whole rows = 32 consecutive words.  Both access
// kinds are conflict-free and the image is 20 % smaller than with padded rows (49 KB for 128x128: 3 workgroups / CU).
constexpr int SPROW = 32;                               // bf16 per row
__device__ __forceinline__ int kd_sw_chunk(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 8; }   // element offset

template <int PRO, int EPI, int WM, int WN, bool SPLIT>
// Occupancy: 3 workgroups / CU wherever 168 registers and 3 x 49 KB of LDS allow it (fp32: all but the 256x64 PRO2
// kernel; split: the 128x128 kernels except PRO2, whose two-tensor prologue would spill 46 registers).  The third
// workgroup is worth 8-12 % on the forward kernels: these loops are latency-bound, not pipe-bound (PMC: matrix pipe
// busy ~31 %, VALU ~10 %, LDS ~22 % at 2 workgroups / CU).
// (workgroups per CU by register budget: 3 x 168 registers where the prologue is light; the layer-0-recompute prologue (PRO3) and the
// two- / three-tensor prologues spill at 168 and run 2 x 256)
__global__ __launch_bounds__(256, SPLIT ? ((WM == 2 && PRO != 2 && PRO != 3 && PRO != 4) ? 3 : 2) : ((PRO == 3 || PRO == 4 || (PRO == 2 && WM == 4)) ? 2 : 3)) void pw_gemm_kernel(GemmArgs g) {
  constexpr int BMt = 64 * WM, BNt = 64 * WN;
  constexpr int AF = BMt / 32, BF = BNt / 32;          // float4 loads per thread per K-tile
  constexpr int SMEM_FLOATS = SPLIT ? (BMt + BNt) * 3 * SPROW / 2 : (BMt + BNt) * LDSLD;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* As = smem;
  float* Bs = smem + BMt * LDSLD;
  unsigned short* Ah = reinterpret_cast<unsigned short*>(smem);          // split image: A planes then B planes
  unsigned short* Bh = Ah + 3 * BMt * SPROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WN, wc = wave % WN;
  const int nct = (g.N + BNt - 1) / BNt;
  // XCD-aware tile order (speed only, any placement is correct): workgroups are dealt round-robin
  // over the 8 XCDs, so blocks b, b+8, b+16, ... share an L2.  Give those blocks the nct column
  // tiles of ONE row block, so the A rows are fetched from HBM once and re-read from that XCD's L2
  // (PMC before: FETCH_SIZE 1.3-1.76x the algorithmic bytes for N = 192..768).
  int rowblk, ct;
  {
    const int nrb = (g.M + BMt - 1) / BMt, grp = 8 * nct, g0 = blockIdx.x / grp, r = blockIdx.x % grp;
    if ((g0 + 1) * 8 <= nrb) { rowblk = g0 * 8 + (r & 7); ct = r >> 3; }
    else { const int rem = nrb - g0 * 8; rowblk = g0 * 8 + r % rem; ct = r / rem; }
  }
  const int64_t m0 = (int64_t)rowblk * BMt;
  const int n0 = ct * BNt;
  const int nk = (g.K + BK - 1) / BK;
  if (g.m_dev) {                      // data-dependent M (compacted LiDAR points): no host round trip
    const int mv = *g.m_dev;
    g.M = mv < g.M ? mv : g.M;
    if (m0 >= g.M) return;            // whole workgroup leaves before any barrier
  }
  // (A persistent variant -- one workgroup per resident slot walking tiles, next tile's first loads issued before
  // the epilogue -- was measured 6-12 % SLOWER on the LiDAR shapes and is not kept.)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Loads are BRANCH-FREE: out-of-range rows / columns / k read a clamped (valid) address and are
  // zeroed by a select in `transform`, which runs only when the tile is about to be written to LDS.
  // That keeps all loads of a tile in ONE basic block, issued back to back, in flight under the
  // MFMAs of the previous tile.  Per-channel coefficients depend only on (k0, tid & 7).
  // PF2: a second register stage for the operand tiles (split arithmetic, 128x128 tile, PRO 0/1): the matrix work
  // of one K-tile (~0.6 us) is shorter than the HBM latency under load, so two K-tiles of loads are kept in flight.
  constexpr bool PF2 = false;   // (a second register stage costs the third workgroup per CU, which is worth more)
  // Per-channel coefficients travel WITH their tile (loaded just before it): a coefficient load issued later than
  // a prefetch would sit behind it in the in-order vmcnt queue and drain the prefetch when first used.  Only the
  // fp32 PRO2 kernel (170-register budget at 3 workgroups/CU) fetches its five vectors at the point of use.
  // PRO3 (A = act(bn(layer0(point)))): sc, sh, the four weight rows and the bias of the thread's 4 channels
  constexpr int NCO = (PRO == 2 || PRO == 4) ? 5 : (PRO == 3 ? 7 : 2);
  constexpr bool CO_LATE = PRO == 2 && !SPLIT;
  float4 ra0[AF], rb0[BF], ra1[PF2 ? AF : 1], rb1[PF2 ? BF : 1], rx[PRO >= 2 ? AF : 1], co0[NCO], co1[PF2 ? NCO : 1];
  const int c4 = tid & 7;
  float4 rs[PRO == 4 ? AF : 1];
  int64_t trow[PRO == 4 ? AF : 1];
  bool tval[PRO == 4 ? AF : 1];
  if constexpr (PRO == 4) {           // the thread's rows are the same for every K-tile: one dependent load, up front
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      int64_t gm = m0 + (tid >> 3) + 32 * i;
      gm = gm < g.M ? gm : (int64_t)g.M - 1;
      const int r = g.trows[gm];
      tval[i] = r >= 0;
      trow[i] = (int64_t)(r < 0 ? 0 : r) * g.K;
    }
  }
  auto issue_loads = [&](int kt, float4 (&ra)[AF], float4 (&rb)[BF], float4 (&co)[NCO]) {
    int gk = kt * BK + c4 * 4;
    gk = gk < g.K ? gk : g.K - 4;
    if (PRO == 1 || PRO == 3) { co[0] = kd_ld4(g.p0 + gk); co[1] = kd_ld4(g.p1 + gk); }
    if (PRO == 3) {
#pragma unroll
      for (int j = 0; j < 4; ++j) co[2 + j] = kd_ld4(g.l0w + (gk + j) * 4);
      co[6] = kd_ld4(g.l0b + gk);
    }
    if ((PRO == 2 && !CO_LATE) || PRO == 4) {
      co[0] = kd_ld4(g.p0 + gk); co[1] = kd_ld4(g.p1 + gk); co[2] = kd_ld4(g.p2 + gk); co[3] = kd_ld4(g.p3 + gk); co[4] = kd_ld4(g.p4 + gk);
    }
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      int64_t gm = m0 + (tid >> 3) + 32 * i;
      gm = gm < g.M ? gm : (int64_t)g.M - 1;
      if (PRO == 3) {
        if (kt == 0) rx[i] = kd_ld4(g.A + gm * 4);      // the point: the same row for every K-tile
      } else {
        ra[i] = kd_ld4(g.A + gm * g.lda + gk);
        if (PRO == 2) rx[i] = kd_ld4(g.A2 + gm * g.lda2 + gk);
        if constexpr (PRO == 4) { rx[i] = kd_ld4(g.tmx + trow[i] + gk); rs[i] = kd_ld4(g.tshare + trow[i] + gk); }
      }
    }
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      int gn = n0 + (tid >> 3) + 32 * i;
      gn = gn < g.N ? gn : g.N - 1;
      rb[i] = kd_ld4(g.W + (int64_t)gn * g.K + gk);
    }
  };
  auto transform = [&](int kt, float4 (&ra)[AF], float4 (&rb)[BF], float4 (&co)[NCO]) {
    const bool kok = kt * BK + c4 * 4 < g.K;
    if (CO_LATE) {    // 5 coefficient vectors: cache-resident, fetched here so they do not occupy 20 VGPRs under the MFMAs
      int gk = kt * BK + c4 * 4;
      gk = gk < g.K ? gk : g.K - 4;
      co[0] = kd_ld4(g.p0 + gk); co[1] = kd_ld4(g.p1 + gk); co[2] = kd_ld4(g.p2 + gk); co[3] = kd_ld4(g.p3 + gk); co[4] = kd_ld4(g.p4 + gk);
    }
#pragma unroll
    for (int i = 0; i < AF; ++i) {
      float4 va = ra[i];
      if (PRO == 1) {
        va = kd_affine_act4(va, co[0], co[1], g.pro_act);
      } else if (PRO == 3) {
        const float4 wr[4] = {co[2], co[3], co[4], co[5]};
        va = kd_affine_act4(kd_l0_raw4(rx[i], wr, co[6]), co[0], co[1], g.pro_act);
      } else if (PRO == 2) {
        const float4 x = rx[i];
        va.x = kd_bwd_operand(va.x, x.x, co[0].x, co[1].x, co[2].x, co[3].x, co[4].x, g.pro_act);
        va.y = kd_bwd_operand(va.y, x.y, co[0].y, co[1].y, co[2].y, co[3].y, co[4].y, g.pro_act);
        va.z = kd_bwd_operand(va.z, x.z, co[0].z, co[1].z, co[2].z, co[3].z, co[4].z, g.pro_act);
        va.w = kd_bwd_operand(va.w, x.w, co[0].w, co[1].w, co[2].w, co[3].w, co[4].w, g.pro_act);
      } else if constexpr (PRO == 4) {
        const float4 x = va, mx = rx[i], sv = rs[i];
        const float4 v = kd_affine_act4(x, co[3], co[4], g.pro_act);
        const bool tv = tval[i];
        va.x = kd_bwd_operand((tv && v.x > 0.f && v.x == mx.x) ? sv.x : 0.f, x.x, co[0].x, co[1].x, co[2].x, 0.f, 0.f, KD_ACT_NONE);
        va.y = kd_bwd_operand((tv && v.y > 0.f && v.y == mx.y) ? sv.y : 0.f, x.y, co[0].y, co[1].y, co[2].y, 0.f, 0.f, KD_ACT_NONE);
        va.z = kd_bwd_operand((tv && v.z > 0.f && v.z == mx.z) ? sv.z : 0.f, x.z, co[0].z, co[1].z, co[2].z, 0.f, 0.f, KD_ACT_NONE);
        va.w = kd_bwd_operand((tv && v.w > 0.f && v.w == mx.w) ? sv.w : 0.f, x.w, co[0].w, co[1].w, co[2].w, 0.f, 0.f, KD_ACT_NONE);
      }
      const bool aok = kok && (m0 + (tid >> 3) + 32 * i < g.M);
      ra[i] = make_float4(aok ? va.x : 0.f, aok ? va.y : 0.f, aok ? va.z : 0.f, aok ? va.w : 0.f);
    }
#pragma unroll
    for (int i = 0; i < BF; ++i) {
      const bool bok = kok && (n0 + (tid >> 3) + 32 * i < g.N);
      const float4 vb = rb[i];
      rb[i] = make_float4(bok ? vb.x : 0.f, bok ? vb.y : 0.f, bok ? vb.z : 0.f, bok ? vb.w : 0.f);
    }
  };

#ifdef KD_DBG_PHASES
  long long ph_t = clock64(), ph_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  auto k_tile = [&](int kt, float4 (&ra)[AF], float4 (&rb)[BF], float4 (&co)[NCO]) {
    transform(kt, ra, rb, co);
    KD_PH(0);                          // wait for the tile's loads + BN/act transform
    kd_lds_barrier();
    KD_PH(1);                          // barrier: previous MFMA phase of the slowest wave
    if (SPLIT) {
#pragma unroll
      for (int i = 0; i < AF; ++i) {
        uint2 hi, mid, lo;
        kd_split3(ra[i], hi, mid, lo);
        const int row = (tid >> 3) + 32 * i;
        unsigned short* d = Ah + row * SPROW + kd_sw_chunk(row, c4 >> 1) + (c4 & 1) * 4;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + BMt * SPROW) = mid;
        *reinterpret_cast<uint2*>(d + 2 * BMt * SPROW) = lo;
      }
#pragma unroll
      for (int i = 0; i < BF; ++i) {
        uint2 hi, mid, lo;
        kd_split3(rb[i], hi, mid, lo);
        const int row = (tid >> 3) + 32 * i;
        unsigned short* d = Bh + row * SPROW + kd_sw_chunk(row, c4 >> 1) + (c4 & 1) * 4;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + BNt * SPROW) = mid;
        *reinterpret_cast<uint2*>(d + 2 * BNt * SPROW) = lo;
      }
    } else {
#pragma unroll
      for (int i = 0; i < AF; ++i) kd_st4(As + ((tid >> 3) + 32 * i) * LDSLD + c4 * 4, ra[i]);
#pragma unroll
      for (int i = 0; i < BF; ++i) kd_st4(Bs + ((tid >> 3) + 32 * i) * LDSLD + c4 * 4, rb[i]);
    }
    KD_PH(2);                          // split + LDS stores
    kd_lds_barrier();
    KD_PH(3);                          // barrier after the LDS image
    if (kt + (PF2 ? 2 : 1) < nk) issue_loads(kt + (PF2 ? 2 : 1), ra, rb, co);  // the freed stage refills under this tile's MFMAs
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 a[2][3], b[2][3];
        const int ko = kd_sw_chunk(lane & 31, ks * 2 + (lane >> 5));     // row bits 2..3 == lane bits 2..3 (tile offsets are multiples of 32)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            a[i][p] = *reinterpret_cast<const bf16x8*>(Ah + (p * BMt + wr * 64 + i * 32 + (lane & 31)) * SPROW + ko);
            b[i][p] = *reinterpret_cast<const bf16x8*>(Bh + (p * BNt + wc * 64 + i * 32 + (lane & 31)) * SPROW + ko);
          }
        // smallest terms first; the four accumulators separate two uses of the same one by 3 other MFMAs
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][PA[t]], b[ni][PB[t]], acc[mi][ni], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 a[2], b[2];
      const int ko = kk * 8 + (lane >> 5) * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const f32x4*>(As + (wr * 64 + i * 32 + (lane & 31)) * LDSLD + ko);
        b[i] = *reinterpret_cast<const f32x4*>(Bs + (wc * 64 + i * 32 + (lane & 31)) * LDSLD + ko);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][j], b[ni][j], acc[mi][ni], 0, 0, 0);
    }
  };
  issue_loads(0, ra0, rb0, co0);
  if constexpr (PF2) {
    if (nk > 1) issue_loads(1, ra1, rb1, co1);
    for (int kt = 0; kt < nk; kt += 2) {
      k_tile(kt, ra0, rb0, co0);
      if (kt + 1 < nk) k_tile(kt + 1, ra1, rb1, co1);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) k_tile(kt, ra0, rb0, co0);
  }
  KD_PH(4);                            // MFMA phases (incl. issuing the prefetch)

  // ---- EPI4: eval BatchNorm + ReLU + BEV scatter-max straight from the accumulator tile -----------------------
  // (frozen teacher / validation: the [points, C] output of the last point-MLP layer is never written).  X = int
  // cell index per row (flat (frame, cell) row of the grid, < 0: skip), C = grid [cells][ldc], zero-initialised by the
  // caller.  Values are >= 0 after ReLU, so max == unsigned max of the bit pattern (order-independent, deterministic).
  // Lanes map to CONSECUTIVE channels (4-byte accesses) so one wave's atomics fall in few cache lines, and all
  // current-maximum reads of a half are issued before its first atomic (in-order vmcnt: see the note below).
  if constexpr (EPI == 4) {
    constexpr int TLD4 = BNt + 4, NJ = BNt / 32, NP = WM * 4;         // 8 rows per pass, NP passes per half
    float* T4 = smem;
    const int l32 = tid & 31, r8 = tid >> 5;
    float s4[NJ], h4[NJ], b4[NJ];
    bool ok4[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = n0 + l32 + 32 * j;
      ok4[j] = c < g.N;
      const int cc = ok4[j] ? c : g.N - 1;
      s4[j] = g.esc[cc]; h4[j] = g.esh[cc];
      b4[j] = g.bias ? g.bias[cc] : 0.f;
    }
    const int* cellidx = reinterpret_cast<const int*>(g.X);
    unsigned* grid = reinterpret_cast<unsigned*>(g.C);
    const int cb4 = n0 + l32 < g.N ? n0 + l32 : 0;       // first column of this lane (clamped: partial column tiles)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      kd_lds_barrier();
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          T4[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TLD4 + wc * 64 + ni * 32 + (lane & 31)] = acc[h][ni][r];
      kd_lds_barrier();
      int cell[NP];
      unsigned cur[NP][NJ];
#pragma unroll
      // Each thread owns NP CONSECUTIVE rows (inside one 32-row run of the half): when the caller hands the points
      // over sorted by cell (kd_lidar_cell_sort + kd_lidar_gather_sorted) neighbouring rows share a cell, the running
      // maximum is carried in registers and only the last row of a run touches the grid -- ~5x fewer atomics.
      // Unsorted input is still correct: every row is then its own run.
      for (int i = 0; i < NP; ++i) {
        const int rr = r8 * NP + i;
        int64_t row = m0 + (rr >> 5) * 64 + h * 32 + (rr & 31);
        const bool rok = row < g.M;
        row = rok ? row : (int64_t)g.M - 1;
        const int c = cellidx[row];
        cell[i] = rok ? c : -1;
        const unsigned* src = grid + (int64_t)(c < 0 ? 0 : c) * g.ldc + cb4;
#pragma unroll
        for (int j = 0; j < NJ; ++j) cur[i][j] = src[ok4[j] ? 32 * j : 0];
      }
      float run[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) run[j] = 0.f;
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int rr = r8 * NP + i;
        unsigned* dst = grid + (int64_t)(cell[i] < 0 ? 0 : cell[i]) * g.ldc + cb4;
        const bool first = i == 0 || cell[i] != cell[i > 0 ? i - 1 : 0];
        const bool last = i == NP - 1 || cell[i] != cell[i < NP - 1 ? i + 1 : i];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float v = kd_act(kd_affine(T4[rr * TLD4 + l32 + 32 * j] + b4[j], s4[j], h4[j]), g.epi_act);
          run[j] = (first || v > run[j]) ? v : run[j];
          const unsigned u = __float_as_uint(run[j]);
          if (last && cell[i] >= 0 && ok4[j] && run[j] > 0.f && u > cur[i][j]) atomicMax(dst + 32 * j, u);
        }
      }
    }
    return;
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // The accumulator tile goes through LDS in two halves (the mi = 0 / 1 row groups of every wave) so
  // that C stores and the X / addend loads are whole rows (float4 per lane, BNt*4 contiguous bytes)
  // instead of 4-byte column-strided accesses.
  constexpr bool EPI_BWD = EPI == 2 || EPI == 3;        // dgrad epilogues: multiply by act'(X), BN-backward sums
  constexpr int TLD = BNt + 4;
  constexpr int CPT = BNt / 4, RG = 256 / CPT;        // column groups, row groups; 8 iterations cover WM*32 rows
  static_assert(WM * 32 * TLD <= SMEM_FLOATS, "epilogue staging must fit the operand LDS");
  float* T = smem;
  const int c4e = tid % CPT, rg = tid / CPT;
  const int col = n0 + c4e * 4;
  const bool cok = col < g.N;
  // Per-column vectors: branch-free loads (clamped column, select), then ONE wait here in straight-line code.
  // A load whose first use sits inside a conditional block keeps its registers "pending" for hipcc's waitcnt
  // pass on the other path, and every later use then gets s_waitcnt vmcnt(0) -- which on gfx9 also waits for all
  // earlier STORES: the row loop below used to drain its own store after every row.
  const int colc = cok ? col : g.N - 4;
  float4 bias4 = kd_zero4(), esc = kd_zero4(), esh = kd_zero4(), emean = kd_zero4(), einv = kd_zero4();
  {
    const float* bp = g.bias ? g.bias : g.W;                 // any valid address; the value is dropped by the select
    const float4 bv = kd_ld4(bp + (g.bias ? colc : 0));
    const bool hb = g.bias != nullptr;
    bias4 = make_float4(hb ? bv.x : 0.f, hb ? bv.y : 0.f, hb ? bv.z : 0.f, hb ? bv.w : 0.f);
    if (EPI_BWD) { esc = kd_ld4(g.esc + colc); esh = kd_ld4(g.esh + colc); emean = kd_ld4(g.emean + colc); einv = kd_ld4(g.einv + colc); }
    if (EPI == 5) { esc = kd_ld4(g.esc + colc); esh = kd_ld4(g.esh + colc); }
  }
  float4 ew[EPI == 3 ? 4 : 1], eb = kd_zero4();      // EPI3: X is layer 0 of the point, recomputed for the thread's 4 columns
  if (EPI == 3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ew[j] = kd_ld4(g.l0w + (colc + j) * 4);
    eb = kd_ld4(g.l0b + colc);
  }
  float4 s1 = kd_zero4(), s2 = kd_zero4();
  float4 m1[EPI == 3 ? 4 : 1];                         // EPI3: sum over rows of G0 * point coordinate j, per column
#pragma unroll
  for (int j = 0; j < (EPI == 3 ? 4 : 1); ++j) m1[j] = kd_zero4();
  constexpr int NI = (WM * 32) / RG;                   // rows per thread per half
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    KD_PHE(h == 0 ? 4 : 3);
    kd_lds_barrier();                                 // LDS free: K-loop reads / previous half done
    KD_PHE(0);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        T[(wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * TLD + wc * 64 + ni * 32 + (lane & 31)] = acc[h][ni][r];
    KD_PHE(1);
    kd_lds_barrier();
    KD_PHE(2);
    // All global LOADS of this half are issued before its first store (clamped addresses, no branches), so no
    // wait on a load ever has an older store in front of it in the in-order vmcnt queue.
    float4 xr[(EPI_BWD || EPI == 5) ? NI : 1];
    if (EPI == 5) {                                  // EPI5: the residual is added AFTER BatchNorm + activation
      const bool has = g.addend != nullptr;
      const float* ap = has ? g.addend : g.W;        // any valid address; the value is dropped by the select
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = rg + RG * i;
        int64_t row = m0 + (rr >> 5) * 64 + h * 32 + (rr & 31);
        row = row < g.M ? row : (int64_t)g.M - 1;
        const float4 ad = kd_ld4(ap + (has ? row * g.ldadd + colc : 0));
        xr[i] = make_float4(has ? ad.x : 0.f, has ? ad.y : 0.f, has ? ad.z : 0.f, has ? ad.w : 0.f);
      }
    }
    if (EPI_BWD) {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = rg + RG * i;
        int64_t row = m0 + (rr >> 5) * 64 + h * 32 + (rr & 31);
        row = row < g.M ? row : (int64_t)g.M - 1;
        xr[i] = EPI == 3 ? kd_ld4(g.X + row * 4) : kd_ld4(g.X + row * g.ldx + colc);
      }
    }
    if (EPI != 5 && g.addend) {                      // residual gradient: folded into the staged tile (own elements only)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int rr = rg + RG * i;
        int64_t row = m0 + (rr >> 5) * 64 + h * 32 + (rr & 31);
        row = row < g.M ? row : (int64_t)g.M - 1;
        const float4 ad = kd_ld4(g.addend + row * g.ldadd + colc);
        float4 t = kd_ld4(T + rr * TLD + c4e * 4);
        t.x += ad.x; t.y += ad.y; t.z += ad.z; t.w += ad.w;
        kd_st4(T + rr * TLD + c4e * 4, t);
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = rg + RG * i;
      const int64_t row = m0 + (rr >> 5) * 64 + h * 32 + (rr & 31);
      const bool ok = cok && row < g.M;
      float4 v = kd_ld4(T + rr * TLD + c4e * 4);
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (EPI_BWD) {
        float4 x = xr[i];
        if constexpr (EPI == 3) x = kd_l0_raw4(xr[i], ew, eb);
        v.x *= kd_act_mask(kd_affine(x.x, esc.x, esh.x), g.epi_act);
        v.y *= kd_act_mask(kd_affine(x.y, esc.y, esh.y), g.epi_act);
        v.z *= kd_act_mask(kd_affine(x.z, esc.z, esh.z), g.epi_act);
        v.w *= kd_act_mask(kd_affine(x.w, esc.w, esh.w), g.epi_act);
        if (ok) {
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x = fmaf(v.x, (x.x - emean.x) * einv.x, s2.x);
          s2.y = fmaf(v.y, (x.y - emean.y) * einv.y, s2.y);
          s2.z = fmaf(v.z, (x.z - emean.z) * einv.z, s2.z);
          s2.w = fmaf(v.w, (x.w - emean.w) * einv.w, s2.w);
          if constexpr (EPI == 3) {
            const float pj[4] = {xr[i].x, xr[i].y, xr[i].z, xr[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              m1[j].x = fmaf(v.x, pj[j], m1[j].x); m1[j].y = fmaf(v.y, pj[j], m1[j].y);
              m1[j].z = fmaf(v.z, pj[j], m1[j].z); m1[j].w = fmaf(v.w, pj[j], m1[j].w);
            }
          }
        }
      } else if (EPI == 5) {
        // inference finish: eval-mode BatchNorm + activation (+ residual) applied here, same operation order as the
        // separate kd_bn_act_apply pass (act(fma(raw, sc, sh)) + res on the fp32-rounded raw value): identical bits
        v = kd_affine_act4(v, esc, esh, g.epi_act);
        v.x += xr[i].x; v.y += xr[i].y; v.z += xr[i].z; v.w += xr[i].w;
      } else if (EPI == 1) {
        if (ok) {
          s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
          s2.x = fmaf(v.x, v.x, s2.x); s2.y = fmaf(v.y, v.y, s2.y); s2.z = fmaf(v.z, v.z, s2.z); s2.w = fmaf(v.w, v.w, s2.w);
        }
      }
      if (EPI == 3 ? (ok && g.C != nullptr) : ok) {
        if (g.nt_store) kd_st4_nt(g.C + row * g.ldc + col, v); else kd_st4(g.C + row * g.ldc + col, v);
      }
    }
  }
  if (EPI != 0 && EPI != 5) {
    kd_lds_barrier();
    float* red = smem;                               // [RG row groups][2 stats][BNt columns]
    kd_st4(red + (rg * 2 + 0) * BNt + c4e * 4, s1);
    kd_st4(red + (rg * 2 + 1) * BNt + c4e * 4, s2);
    kd_lds_barrier();
    // the stats slab has one row per 128 matrix rows (kd_pwconv_stat_rows): a 256-row tile fills
    // slab row 2*rowblk and zeroes row 2*rowblk+1
    const int st = tid / BNt, c = tid % BNt;
    if (tid < 2 * BNt && n0 + c < g.N) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < RG; ++k) s += red[(k * 2 + st) * BNt + c];
      const int64_t srow = (int64_t)rowblk * (BMt / 128);
      g.partial[(srow * 2 + st) * g.N + n0 + c] = s;
      if (BMt == 256 && (srow + 1) * 128 < g.M) g.partial[((srow + 1) * 2 + st) * g.N + n0 + c] = 0.f;
    }
    if constexpr (EPI == 3) {
      if (g.m1slab) {                                // same two-level sum for the four G0*point moments: one slab row per TILE
        kd_lds_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) kd_st4(red + (rg * 4 + j) * BNt + c4e * 4, m1[j]);
        kd_lds_barrier();
        for (int i = tid; i < 4 * BNt; i += 256) {
          const int j = i / BNt, c = i % BNt;
          if (n0 + c < g.N) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < RG; ++k) s += red[(k * 4 + j) * BNt + c];
            g.m1slab[((int64_t)rowblk * 4 + j) * g.N + n0 + c] = s;
          }
        }
      }
    }
  }
  KD_PH(5);                            // epilogue
#ifdef KD_DBG_PHASES
  if (tid == 0) {
    atomicAdd(&kd_dbg_counters[6], 1ull);
    for (int i = 0; i < 6; ++i) atomicAdd(&kd_dbg_counters[i], (unsigned long long)ph_acc[i]);
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[n][k] = sum_m Deff[m][n] * Aeff[m][k].  Output tile (64*WN) x (64*WK); WM waves split
// the rows of each staged chunk; the grid additionally splits M into `nsplit` slices whose partial
// tiles go to a slab [nsplit][N][K] summed in fixed order by wgrad_reduce_kernel (deterministic).
// (WgradArgs and kd_tr_frag: kd_gemm_args.h -- shared with the role-specialised form in kd_wgrad_rs.hip)
// timing-only probes of dev builds (-DKD_WG_PROBE=bits; results WRONG by construction): 1 one of the six piece products, 2 every chunk
// loads the slice's FIRST rows (operands stay cache-resident: no HBM latency)
#ifndef KD_WG_PROBE
#define KD_WG_PROBE 0
#endif
template <int WN, int WK, int WM, int DMODE, int AMODE, bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void pw_wgrad_kernel(WgradArgs g) {
  constexpr int KS = WM == 1 ? 2 : 1;                           // SPLIT: 16-row MFMA steps per wave per chunk
  constexpr int TN = 64 * WN, TK = 64 * WK, CH = SPLIT ? 16 * WM * KS : 32 * WM;
  constexpr int DF4 = CH * TN / 1024, AF4 = CH * TK / 1024;     // float4 per thread per chunk
  constexpr int LDN = TN + 32, LDK = TK + 32;                   // SPLIT: bf16 per LDS row
  constexpr int SMEM = SPLIT ? 3 * CH * (LDN + LDK) / 2 : CH * (TN + TK);
  static_assert((WM - 1) * WN * WK * 4096 <= SMEM, "cross-wave reduction must fit the staging LDS");
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* Ds = smem;
  float* As = smem + CH * TN;
  unsigned short* Dh = reinterpret_cast<unsigned short*>(smem);
  unsigned short* Ah = Dh + 3 * CH * LDN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / (WN * WK), wn = (wave % (WN * WK)) / WK, wk = wave % WK;
  const int ntn = (g.N + TN - 1) / TN, ntk = (g.K + TK - 1) / TK;
  // XCD-aware order (speed only, as in pw_gemm_kernel): blocks b, b + 8, b + 16, ... share an L2, so THEY get the ntn * ntk output
  // tiles of one row slice -- its D / A rows then come from HBM once and from that XCD's L2 for the other tiles (round 3; PMC
  // before: the weight-gradient family moved 1.25x its algorithmic bytes, blockIdx % ntiles having dealt a slice's tiles over all XCDs).
  // Selected per launch (g.xcd_order, launch_wgrad): it is faster for few tiles per slice and slower for many.
  int tile, split;
  {
    const int ntiles = ntn * ntk, nsl = (int)gridDim.x / ntiles, grp = 8 * ntiles, g0 = (int)blockIdx.x / grp, r = (int)blockIdx.x % grp;
    if (!g.xcd_order) { tile = (int)blockIdx.x % ntiles; split = (int)blockIdx.x / ntiles; }
    else if ((g0 + 1) * 8 <= nsl) { split = g0 * 8 + (r & 7); tile = r >> 3; }
    else { const int rem = nsl - g0 * 8; split = g0 * 8 + r % rem; tile = r / rem; }
  }
  const int n0 = (tile / ntk) * TN, k0 = (tile % ntk) * TK;
  const int64_t mbeg = (int64_t)split * g.rows_per_split;
  int64_t mend = mbeg + g.rows_per_split;
  if (mend > g.M) mend = g.M;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 rd[DF4], ra[AF4];
  // Branch-free loads (clamped addresses, zero-select in `transform` right before the LDS store), so
  // a chunk's loads sit in one basic block and stay in flight under the previous chunk's MFMAs.
  // A thread's columns are fixed (256 % (TN/4) == 0), so its coefficient vectors are loaded once.
  const int dc4 = tid % (TN / 4), ac4 = tid % (TK / 4);
  const int gn = n0 + dc4 * 4, gk = k0 + ac4 * 4;
  const bool nok = gn < g.N, kok = gk < g.K;
  const int gnc = nok ? gn : g.N - 4, gkc = kok ? gk : g.K - 4;
  float4 cd[DMODE >= 2 ? 5 : 1], ca[AMODE >= 1 ? 2 : 1], rx[DMODE >= 2 ? DF4 : 1], cw[AMODE == 2 ? 4 : 1], cb = kd_zero4();
  float4 rs[DMODE == 3 ? DF4 : 1];
  int tr_cur[DMODE == 3 ? DF4 : 1], tr_nxt[DMODE == 3 ? DF4 : 1];      // grid rows of this / the next chunk's points
  if (DMODE >= 2) {
    cd[0] = kd_ld4(g.al + gnc); cd[1] = kd_ld4(g.be + gnc); cd[2] = kd_ld4(g.ga + gnc);
    cd[3] = kd_ld4(g.msc + gnc); cd[4] = kd_ld4(g.msh + gnc);
  }
  if (AMODE >= 1) { ca[0] = kd_ld4(g.asc + gkc); ca[1] = kd_ld4(g.ash + gkc); }
  if (AMODE == 2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) cw[j] = kd_ld4(g.l0w + (gkc + j) * 4);
    cb = kd_ld4(g.l0b + gkc);
  }
  auto load_rows = [&](int64_t mc, int (&tr)[DMODE == 3 ? DF4 : 1]) {
    if constexpr (DMODE == 3) {
#pragma unroll
      for (int i = 0; i < DF4; ++i) {
        int64_t gm = mc + (tid + 256 * i) / (TN / 4);
        gm = gm < mend ? gm : mend - 1;
        tr[i] = g.trows[gm];
      }
    }
  };
  auto load_chunk = [&](int64_t mc) {
    if (KD_WG_PROBE & 2) mc = mbeg;
    if constexpr (DMODE == 3) {
      // the table rows of THIS chunk were fetched one chunk ago (tr_nxt): no dependent load in the steady state
#pragma unroll
      for (int i = 0; i < DF4; ++i) tr_cur[i] = tr_nxt[i];
      load_rows(mc + CH < mend ? mc + CH : mc, tr_nxt);
    }
#pragma unroll
    for (int i = 0; i < DF4; ++i) {
      int64_t gm = mc + (tid + 256 * i) / (TN / 4);
      gm = gm < mend ? gm : mend - 1;
      rd[i] = kd_ld4(g.D + gm * g.ldd + gnc);
      if (DMODE == 2) rx[i] = kd_ld4(g.X + gm * g.ldx + gnc);
      if constexpr (DMODE == 3) {
        const int64_t o = (int64_t)(tr_cur[i] < 0 ? 0 : tr_cur[i]) * g.N + gnc;
        rx[i] = kd_ld4(g.tmx + o);
        rs[i] = kd_ld4(g.tshare + o);
      }
    }
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      int64_t gm = mc + (tid + 256 * i) / (TK / 4);
      gm = gm < mend ? gm : mend - 1;
      ra[i] = AMODE == 2 ? kd_ld4(g.A + gm * 4) : kd_ld4(g.A + gm * g.lda + gkc);
    }
  };
  auto transform = [&](int64_t mc) {
#pragma unroll
    for (int i = 0; i < DF4; ++i) {
      float4 v = rd[i];
      if (DMODE == 2) {
        const float4 x = rx[i];
        v.x = kd_bwd_operand(v.x, x.x, cd[0].x, cd[1].x, cd[2].x, cd[3].x, cd[4].x, g.d_act);
        v.y = kd_bwd_operand(v.y, x.y, cd[0].y, cd[1].y, cd[2].y, cd[3].y, cd[4].y, g.d_act);
        v.z = kd_bwd_operand(v.z, x.z, cd[0].z, cd[1].z, cd[2].z, cd[3].z, cd[4].z, g.d_act);
        v.w = kd_bwd_operand(v.w, x.w, cd[0].w, cd[1].w, cd[2].w, cd[3].w, cd[4].w, g.d_act);
      } else if constexpr (DMODE == 3) {
        const float4 x = v, mx = rx[i], sv = rs[i];
        const float4 a = kd_affine_act4(x, cd[3], cd[4], g.d_act);
        const bool tv = tr_cur[i] >= 0;
        v.x = kd_bwd_operand((tv && a.x > 0.f && a.x == mx.x) ? sv.x : 0.f, x.x, cd[0].x, cd[1].x, cd[2].x, 0.f, 0.f, KD_ACT_NONE);
        v.y = kd_bwd_operand((tv && a.y > 0.f && a.y == mx.y) ? sv.y : 0.f, x.y, cd[0].y, cd[1].y, cd[2].y, 0.f, 0.f, KD_ACT_NONE);
        v.z = kd_bwd_operand((tv && a.z > 0.f && a.z == mx.z) ? sv.z : 0.f, x.z, cd[0].z, cd[1].z, cd[2].z, 0.f, 0.f, KD_ACT_NONE);
        v.w = kd_bwd_operand((tv && a.w > 0.f && a.w == mx.w) ? sv.w : 0.f, x.w, cd[0].w, cd[1].w, cd[2].w, 0.f, 0.f, KD_ACT_NONE);
      }
      const bool ok = nok && (mc + (tid + 256 * i) / (TN / 4) < mend);
      rd[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
#pragma unroll
    for (int i = 0; i < AF4; ++i) {
      float4 v = ra[i];
      if (AMODE == 1) v = kd_affine_act4(v, ca[0], ca[1], g.a_act);
      if constexpr (AMODE == 2) v = kd_affine_act4(kd_l0_raw4(v, cw, cb), ca[0], ca[1], g.a_act);
      const bool ok = kok && (mc + (tid + 256 * i) / (TK / 4) < mend);
      ra[i] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
  };

  if (mbeg < mend) { load_rows(mbeg, tr_nxt); load_chunk(mbeg); }
  for (int64_t mc = mbeg; mc < mend; mc += CH) {
    transform(mc);
    kd_lds_barrier();
    if (SPLIT) {
#pragma unroll
      for (int i = 0; i < DF4; ++i) {
        const int idx = tid + 256 * i;
        uint2 hi, mid, lo;
        kd_split3(rd[i], hi, mid, lo);
        unsigned short* d = Dh + (idx / (TN / 4)) * LDN + (idx % (TN / 4)) * 4;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + CH * LDN) = mid;
        *reinterpret_cast<uint2*>(d + 2 * CH * LDN) = lo;
      }
#pragma unroll
      for (int i = 0; i < AF4; ++i) {
        const int idx = tid + 256 * i;
        uint2 hi, mid, lo;
        kd_split3(ra[i], hi, mid, lo);
        unsigned short* d = Ah + (idx / (TK / 4)) * LDK + (idx % (TK / 4)) * 4;
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + CH * LDK) = mid;
        *reinterpret_cast<uint2*>(d + 2 * CH * LDK) = lo;
      }
    } else {
#pragma unroll
      for (int i = 0; i < DF4; ++i) {
        const int idx = tid + 256 * i;
        kd_st4(Ds + (idx / (TN / 4)) * TN + (idx % (TN / 4)) * 4, rd[i]);
      }
#pragma unroll
      for (int i = 0; i < AF4; ++i) {
        const int idx = tid + 256 * i;
        kd_st4(As + (idx / (TK / 4)) * TK + (idx % (TK / 4)) * 4, ra[i]);
      }
    }
    kd_lds_barrier();
    if (mc + CH < mend) load_chunk(mc + CH);
    if (SPLIT) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int mrow = (wm * KS + ks) * 16;
        bf16x8 d[2][3], a[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            d[i][p] = kd_tr_frag(Dh + p * CH * LDN, LDN, mrow, wn * 64 + i * 32, lane);
            a[i][p] = kd_tr_frag(Ah + p * CH * LDK, LDK, mrow, wk * 64 + i * 32, lane);
          }
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < ((KD_WG_PROBE & 1) ? 1 : 6); ++t)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)
              acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[ni][PA[t]], a[ki][PB[t]], acc[ni][ki], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int m = wm * 32 + 2 * s + (lane >> 5);
      float d[2], a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        d[i] = Ds[m * TN + wn * 64 + i * 32 + (lane & 31)];
        a[i] = As[m * TK + wk * 64 + i * 32 + (lane & 31)];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
          acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[ni], a[ki], acc[ni][ki], 0, 0, 0);
    }
  }

  if (WM > 1) {                                   // sum the WM row-slices of this tile through LDS
    kd_lds_barrier();
    if (wm > 0) {
      float* dst = smem + (((wm - 1) * WN + wn) * WK + wk) * 4096;
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((ni * 2 + ki) * 16 + r) * 64 + lane] = acc[ni][ki][r];
    }
    kd_lds_barrier();
    if (wm == 0) {
#pragma unroll 1
      for (int o = 1; o < WM; ++o) {         // (one slice at a time: unrolled, hipcc hoists all 3 x 64 LDS reads and spills)
        const float* src = smem + (((o - 1) * WN + wn) * WK + wk) * 4096;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int ki = 0; ki < 2; ++ki)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][ki][r] += src[((ni * 2 + ki) * 16 + r) * 64 + lane];
      }
    }
  }
  if (wm == 0) {
    float* out = g.slab + (int64_t)split * g.N * g.K;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        const int col = k0 + wk * 64 + ki * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = n0 + wn * 64 + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < g.N && col < g.K) out[(int64_t)row * g.K + col] = acc[ni][ki][r];
        }
      }
  }
}

__global__ void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int Cc) {
  // out[c][r] = in[r][c]; weights only (<= 768x256), so no tiling effort
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < R * Cc) out[(int64_t)(i % Cc) * R + i / Cc] = in[i];
}

// every weight of a model in one launch: table[i] = {in, out, R, Cc, first block}; a block looks its matrix up by block index
__global__ void transpose_batch_kernel(const int64_t* __restrict__ table, int n) {
  int m = 0;
  while (m + 1 < n && (int64_t)blockIdx.x >= table[(m + 1) * 5 + 4]) ++m;
  const float* in = reinterpret_cast<const float*>(table[m * 5]);
  float* out = reinterpret_cast<float*>(table[m * 5 + 1]);
  const int R = (int)table[m * 5 + 2], Cc = (int)table[m * 5 + 3];
  const int i = ((int)blockIdx.x - (int)table[m * 5 + 4]) * 256 + threadIdx.x;
  if (i < R * Cc) out[(int64_t)(i % Cc) * R + i / Cc] = in[i];
}

template <int WN, int WK, int WM, bool SPLIT>
int launch_wgrad(WgradArgs& g, size_t ws_bytes, float* dW, hipStream_t st) {
  constexpr int TN = 64 * WN, TK = 64 * WK, CH = SPLIT ? 16 * WM * (WM == 1 ? 2 : 1) : 32 * WM;
  const int ntiles = ((g.N + TN - 1) / TN) * ((g.K + TK - 1) / TK);
  int nsplit = (512 + ntiles - 1) / ntiles;
  int64_t chunks = (g.M + CH - 1) / CH;
  if (nsplit > chunks) nsplit = (int)chunks;
  if (nsplit < 1) nsplit = 1;
  const size_t cap = ws_bytes / ((size_t)g.N * g.K * sizeof(float));
  if ((size_t)nsplit > cap) nsplit = (int)cap;
  KD_REQUIRE(nsplit >= 1, KD_ERR_WORKSPACE, "kd_pwconv_wgrad: workspace too small (%zu B for N=%d K=%d)", ws_bytes, g.N, g.K);
  int64_t cps = (chunks + nsplit - 1) / nsplit;
  g.rows_per_split = (int)(cps * CH);
  nsplit = (int)((g.M + g.rows_per_split - 1) / g.rows_per_split);
  const dim3 grid(ntiles * nsplit), blk(256);
  // measured on the same box (tools/bench_wgrad.py, 256 frames, KD_WGRAD_XCD=0|1): the XCD-aware order wins 1-8 % where a row slice has
  // three output tiles (64 <-> 384 layers) and LOSES 29-34 % where it has six (128 <-> 768: 462 -> 598 us), so it is chosen by tile count
  static const int xcd_env = [] { const char* e = getenv("KD_WGRAD_XCD"); return e ? atoi(e) : -1; }();
  g.xcd_order = xcd_env >= 0 ? xcd_env : (ntiles > 1 && ntiles <= 3);
  if (g.d_mode == 3) hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 3, 1, SPLIT>), grid, blk, 0, st, g);
  else if (g.d_mode == 2 && g.a_mode == 2) hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 2, 2, SPLIT>), grid, blk, 0, st, g);
  else if (g.d_mode == 2 && g.a_mode == 1) hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 2, 1, SPLIT>), grid, blk, 0, st, g);
  else if (g.d_mode == 2) hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 2, 0, SPLIT>), grid, blk, 0, st, g);
  else if (g.a_mode == 1) hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 0, 1, SPLIT>), grid, blk, 0, st, g);
  else hipLaunchKernelGGL((pw_wgrad_kernel<WN, WK, WM, 0, 0, SPLIT>), grid, blk, 0, st, g);
  return kd_slab_reduce_launch(g.slab, nsplit, (int64_t)g.N * g.K, dW, st);
}


int64_t stat_rows_for(int64_t M, int K, int N, int pro, int epi, bool add) {
  if (g_gemm_split.load(std::memory_order_relaxed)) {
    const int r = kd_gemm_stream_stat_rows(M, K, N, pro, epi, add);
    if (r > 0) return r;
  }
  return (M + BM - 1) / BM;
}

// partial_rows: the row count the CALLER sized its statistics slab for (and will hand to kd_bn_finalize_train /
// kd_bn_bwd_finalize).  The two kernel forms write different row counts (one per workgroup / one per 128 matrix rows), and
// which form runs depends on process-wide switches that may have moved since the caller asked: a mismatch is an error
// here, never a silently short or stale reduction downstream (round 2's 3e-4 systematic gradient error).
int gemm_launch(GemmArgs& g, int pro, int epi, hipStream_t st, int64_t partial_rows = -1) {
  if (g.partial && (epi == 1 || epi == 2 || epi == 3)) {
    const int64_t want = stat_rows_for(g.M, g.K, g.N, pro, epi, g.addend != nullptr);
    KD_REQUIRE(partial_rows == want, KD_ERR_ARG,
               "GEMM statistics slab: caller sized %lld rows, this launch (M=%d K=%d N=%d pro=%d epi=%d, %s kernel) writes %lld "
               "-- query kd_pwconv_stat_rows_for() in the same arithmetic / streaming mode as the launch",
               (long long)partial_rows, g.M, g.K, g.N, pro, epi,
               want == (g.M + BM - 1) / BM ? "tiled" : "streaming", (long long)want);
  }
  const dim3 blk(256);
  g.nt_store = kd_nt_store((size_t)g.M * g.N * sizeof(float));   // (non-temporal LOADS of A measured neutral: not kept)
  // 256x64 tiles when the last column tile would be <= 64 wide (N = 32, 64, 192, ...)
  // (the table-form scatter gradient, PRO4, only ever has N = 128 output columns -- the point MLP's hidden width: square tiles only;
  // its 256x64 instance needed 124 bytes of scratch and is not built)
  const bool tall = pro != 4 && ((g.N - 1) % 128) < 64;
  const int64_t M = g.M;
  const int64_t ntile = tall ? ((M + 255) / 256) * ((g.N + 63) / 64) : ((M + 127) / 128) * ((g.N + 127) / 128);
  const bool split = g_gemm_split.load(std::memory_order_relaxed) != 0;
  if (split) {        // skinny shapes whose weights fit in LDS: the weight-resident streaming kernels (kd_gemm_stream.hip)
    const int rc = kd_gemm_stream_launch(g, pro, epi, st);
    if (rc < 0) return KD_ERR_ARG;
    if (rc > 0) return KD_OK;
  }
  const dim3 grid((unsigned)ntile);
#define KD_GEMM_CASE(P_, E_)                                                                       \
  if (pro == P_ && epi == E_) {                                                                    \
    if (split) {                                                                                   \
      if (tall) hipLaunchKernelGGL((pw_gemm_kernel<P_, E_, 4, 1, true>), grid, blk, 0, st, g);     \
      else hipLaunchKernelGGL((pw_gemm_kernel<P_, E_, 2, 2, true>), grid, blk, 0, st, g);          \
    } else {                                                                                       \
      if (tall) hipLaunchKernelGGL((pw_gemm_kernel<P_, E_, 4, 1, false>), grid, blk, 0, st, g);    \
      else hipLaunchKernelGGL((pw_gemm_kernel<P_, E_, 2, 2, false>), grid, blk, 0, st, g);         \
    }                                                                                              \
  }
  KD_GEMM_CASE(0, 0) KD_GEMM_CASE(0, 1) KD_GEMM_CASE(0, 2)
  KD_GEMM_CASE(1, 0) KD_GEMM_CASE(1, 1) KD_GEMM_CASE(1, 2)
  KD_GEMM_CASE(2, 0) KD_GEMM_CASE(2, 1) KD_GEMM_CASE(2, 2)
  KD_GEMM_CASE(3, 0) KD_GEMM_CASE(3, 1) KD_GEMM_CASE(2, 3)      // LiDAR layer 0 recomputed from the points
  KD_GEMM_CASE(1, 4)                                              // last point-MLP layer + BEV scatter-max (eval)
  KD_GEMM_CASE(0, 5) KD_GEMM_CASE(1, 5)                           // inference: eval BatchNorm + act (+ residual) in the epilogue
#undef KD_GEMM_CASE
  if (pro == 4 && epi == 2) {                                     // last point-MLP layer: dgrad with G rebuilt from the tables
    if (split) hipLaunchKernelGGL((pw_gemm_kernel<4, 2, 2, 2, true>), grid, blk, 0, st, g);
    else hipLaunchKernelGGL((pw_gemm_kernel<4, 2, 2, 2, false>), grid, blk, 0, st, g);
  }
  return kd_check_launch("kd_pwconv_gemm");
}

int wgrad_launch(WgradArgs& g, size_t ws_bytes, float* dW, hipStream_t st) {
  const int N = g.N, K = g.K;
  if (g_gemm_split.load(std::memory_order_relaxed)) {
    const int rc = kd_wgrad_rs_launch(g, ws_bytes, dW, st);      // the role-specialised form where the layer has one (kd_wgrad_rs.hip)
    if (rc < 0) return KD_ERR_ARG;
    if (rc > 0) return KD_OK;
    if (N > 64 && K > 64) return launch_wgrad<2, 2, 1, true>(g, ws_bytes, dW, st);
    if (N > 64) return launch_wgrad<2, 1, 2, true>(g, ws_bytes, dW, st);
    if (K > 64) return launch_wgrad<1, 2, 2, true>(g, ws_bytes, dW, st);
    return launch_wgrad<1, 1, 4, true>(g, ws_bytes, dW, st);
  }
  if (N > 64 && K > 64) return launch_wgrad<2, 2, 1, false>(g, ws_bytes, dW, st);
  if (N > 64) return launch_wgrad<2, 1, 2, false>(g, ws_bytes, dW, st);
  if (K > 64) return launch_wgrad<1, 2, 2, false>(g, ws_bytes, dW, st);
  return launch_wgrad<1, 1, 4, false>(g, ws_bytes, dW, st);
}

}  // namespace

int kd_gemm_split_mode() { return g_gemm_split.load(std::memory_order_relaxed); }

extern "C" {

// 0: v_mfma_f32_32x32x2_f32 (exact fp32 products); 1: bf16x6 split products on v_mfma_f32_32x32x16_bf16.
// Process-wide switch for the forward / dgrad / wgrad GEMMs; returns the previous value.
int kd_set_gemm_split(int on) { return g_gemm_split.exchange(on ? 1 : 0); }
#ifdef KD_DBG_PHASES
int kd_dbg_read(unsigned long long* out, int reset) {
  hipDeviceSynchronize();
  hipMemcpyFromSymbol(out, HIP_SYMBOL(kd_dbg_counters), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(kd_dbg_counters), z, sizeof(z)); }
  return 0;
}
#endif

// Rows of the BN-statistics slab a GEMM over M rows writes ([rows][2][N] floats).
int64_t kd_pwconv_stat_rows(int64_t M) { return (M + BM - 1) / BM; }
// ... for the launch the dispatcher will actually make for (K, N, pro, epi) in the current arithmetic: the streaming
// kernels write one row per workgroup (<= 256), the tiled kernels one per 128 matrix rows.  Callers size the slab AND tell
// kd_bn_finalize_train / kd_bn_bwd_finalize how many rows to sum with this number.
int64_t kd_pwconv_stat_rows_for(int64_t M, int K, int N, int pro, int epi, int with_addend) { return stat_rows_for(M, K, N, pro, epi, with_addend != 0); }

// Forward / dgrad GEMM.  See include/kd_hip.h for the argument contract.
int kd_pwconv_gemm(const float* A, int64_t lda, const float* A2, int64_t lda2, int pro, int pro_act, const float* p0,
                   const float* p1, const float* p2, const float* p3, const float* p4, const float* W,
                   const float* bias, float* C, int64_t ldc, const float* addend, int64_t ldadd, int epi,
                   const float* X, int64_t ldx, const float* esc, const float* esh, const float* emean,
                   const float* einv, int epi_act, float* partial, int64_t partial_rows, int64_t M, int K, int N,
                   const int* m_dev, void* stream) {
  KD_REQUIRE(A && W && C && M > 0 && K > 0 && N > 0, KD_ERR_ARG, "kd_pwconv_gemm: null pointer or empty shape");
  KD_REQUIRE(M < (int64_t)1 << 31, KD_ERR_SHAPE, "kd_pwconv_gemm: M=%lld too large", (long long)M);
  KD_REQUIRE(K % 4 == 0 && lda % 4 == 0, KD_ERR_SHAPE, "kd_pwconv_gemm: K=%d and lda=%lld must be multiples of 4", K, (long long)lda);
  KD_REQUIRE(N % 4 == 0 && ldc % 4 == 0 && (!addend || ldadd % 4 == 0) && (!X || ldx % 4 == 0), KD_ERR_SHAPE,
             "kd_pwconv_gemm: N=%d and the output-side row strides must be multiples of 4", N);
  KD_REQUIRE(kd_aligned16(A) && kd_aligned16(W) && kd_aligned16(C) && kd_aligned16(addend) && kd_aligned16(X) &&
             kd_aligned16(bias), KD_ERR_ALIGN, "kd_pwconv_gemm: A/W/C/addend/X/bias must be 16-byte aligned");
  KD_REQUIRE(pro >= 0 && pro <= 2 && ((epi >= 0 && epi <= 2) || epi == 5), KD_ERR_ARG, "kd_pwconv_gemm: bad pro/epi");
  if (epi == 5) KD_REQUIRE(pro <= 1 && esc && esh && kd_aligned16(esc) && kd_aligned16(esh), KD_ERR_ARG, "kd_pwconv_gemm: EPI5 needs pro 0/1 and sc, sh");
  if (pro == 1) KD_REQUIRE(p0 && p1 && kd_aligned16(p0) && kd_aligned16(p1), KD_ERR_ARG, "kd_pwconv_gemm: PRO1 needs sc/sh");
  if (pro == 2) {
    KD_REQUIRE(A2 && p0 && p1 && p2 && lda2 % 4 == 0 && kd_aligned16(A2), KD_ERR_ARG, "kd_pwconv_gemm: PRO2 needs X, al, be, ga");
    KD_REQUIRE(pro_act == KD_ACT_NONE || (p3 && p4), KD_ERR_ARG, "kd_pwconv_gemm: PRO2 mask needs sc/sh");
  }
  if (epi == 1 || epi == 2) KD_REQUIRE(partial, KD_ERR_ARG, "kd_pwconv_gemm: stats epilogue needs a partial slab");
  if (epi == 2) KD_REQUIRE(X && esc && esh && emean && einv, KD_ERR_ARG, "kd_pwconv_gemm: EPI2 needs X, sc, sh, mean, invstd");
  if (pro == 2 && !p3) { p3 = p0; p4 = p0; }          // mask disabled (act none): any valid vector will do
  KD_REQUIRE(K >= 4, KD_ERR_SHAPE, "kd_pwconv_gemm: K must be >= 4");
  GemmArgs g{A, lda, A2, lda2, p0, p1, p2, p3, p4, pro, pro_act, W, bias, C, ldc, addend, ldadd,
             X, ldx, esc, esh, emean, einv, epi_act, partial, (int)M, K, N, m_dev, nullptr, nullptr};
  return gemm_launch(g, pro, epi, (hipStream_t)stream, partial_rows);
}

size_t kd_pwconv_wgrad_ws_bytes(int64_t M, int N, int K) {
  const int tn = N > 64 ? 128 : 64, tk = K > 64 ? 128 : 64;
  const int ntiles = ((N + tn - 1) / tn) * ((K + tk - 1) / tk);
  const int nsplit = (512 + ntiles - 1) / ntiles;
  const size_t tiled = (size_t)nsplit * (size_t)N * (size_t)K * sizeof(float), rs = kd_wgrad_rs_ws_bytes(M, N, K);
  return tiled > rs ? tiled : rs;
}

int kd_pwconv_wgrad(const float* D, int64_t ldd, const float* X, int64_t ldx, int d_mode, int d_act, const float* al,
                    const float* be, const float* ga, const float* msc, const float* msh, const float* A,
                    int64_t lda, int a_mode, int a_act, const float* asc, const float* ash, float* dW, int64_t M,
                    int N, int K, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(D && A && dW && ws && M > 0 && N > 0 && K > 0, KD_ERR_ARG, "kd_pwconv_wgrad: null pointer or empty shape");
  KD_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && lda % 4 == 0, KD_ERR_SHAPE, "kd_pwconv_wgrad: N, K, ld must be multiples of 4");
  KD_REQUIRE(kd_aligned16(D) && kd_aligned16(A) && kd_aligned16(ws), KD_ERR_ALIGN, "kd_pwconv_wgrad: 16-byte alignment");
  if (d_mode == 2) KD_REQUIRE(X && al && be && ga && ldx % 4 == 0, KD_ERR_ARG, "kd_pwconv_wgrad: d_mode 2 needs X, al, be, ga");
  if (d_mode == 2 && d_act != KD_ACT_NONE) KD_REQUIRE(msc && msh, KD_ERR_ARG, "kd_pwconv_wgrad: mask needs sc/sh");
  if (a_mode == 1) KD_REQUIRE(asc && ash, KD_ERR_ARG, "kd_pwconv_wgrad: a_mode 1 needs sc/sh");
  KD_REQUIRE(M < (int64_t)1 << 31, KD_ERR_SHAPE, "kd_pwconv_wgrad: M too large");
  if (d_mode == 2 && !msc) { msc = al; msh = al; }       // mask disabled (act none): any valid vector will do
  WgradArgs g{D, ldd, X, ldx, al, be, ga, msc, msh, d_mode, d_act, A, lda, asc, ash, a_mode, a_act, (float*)ws,
              (int)M, N, K, 0, nullptr, nullptr};
  return wgrad_launch(g, ws_bytes, dW, (hipStream_t)stream);
}


// ---- LiDAR point-MLP layer 1 with layer 0 recomputed from the points (never materialised) -------------------
// forward: C[M,N] = act0(bn0(l0(pts))) . W1^T + bias1; epi 0 store, 1 store + BN statistics
int kd_lidar_l1_fwd(const float* pts, const float* w0, const float* b0, const float* sc0, const float* sh0, int act0,
                    const float* W1, const float* bias1, float* C, int64_t ldc, int epi, float* partial,
                    int64_t partial_rows, int64_t M, int K, int N, const int* m_dev, void* stream) {
  KD_REQUIRE(pts && w0 && b0 && sc0 && sh0 && W1 && C && M > 0 && K >= 4 && N > 0, KD_ERR_ARG, "kd_lidar_l1_fwd: bad args");
  KD_REQUIRE(M < (int64_t)1 << 31 && K % 4 == 0 && N % 4 == 0 && ldc % 4 == 0, KD_ERR_SHAPE, "kd_lidar_l1_fwd: K, N, ldc must be multiples of 4");
  KD_REQUIRE(kd_aligned16(pts) && kd_aligned16(w0) && kd_aligned16(b0) && kd_aligned16(sc0) && kd_aligned16(sh0) && kd_aligned16(W1) &&
             kd_aligned16(C) && kd_aligned16(bias1), KD_ERR_ALIGN, "kd_lidar_l1_fwd: 16-byte alignment");
  KD_REQUIRE((epi == 0 || epi == 1) && (epi == 0 || partial), KD_ERR_ARG, "kd_lidar_l1_fwd: epi must be 0, or 1 with a statistics slab");
  GemmArgs g{pts, 4, nullptr, 0, sc0, sh0, nullptr, nullptr, nullptr, 3, act0, W1, bias1, C, ldc, nullptr, 0,
             nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, partial, (int)M, K, N, m_dev, w0, b0};
  return gemm_launch(g, 3, epi, (hipStream_t)stream, partial_rows);
}

// data gradient: G0[M,K0] = (dy1_eff[M,N1] . W1) * act0'(bn0(l0(pts))) with the BN0-backward sums in `partial`;
// dy1_eff = al*(G*mask(Y1*msc+msh)) + be*Y1 + ga; Wt = W1 as stored ([N1][K0]) transposed by the caller ([K0][N1]).
// (one slab row per tile of the tiled kernel, or per wave of the streaming kernel: the larger of the two)
size_t kd_lidar_l1_dgrad_ws_bytes(int64_t M, int K0) {
  const int64_t tiled = (M + 127) / 128, waves = (M + 31) / 32 < 2048 ? (M + 31) / 32 + 8 : 2048;
  return (size_t)(tiled > waves ? tiled : waves) * 4 * K0 * sizeof(float);
}
// rows of the BatchNorm-backward slab (`partial`, [rows][2][K0]) kd_lidar_l1_dgrad writes for this shape
int64_t kd_lidar_l1_dgrad_stat_rows(int64_t M, int N1, int K0) { return kd_pwconv_stat_rows_for(M, N1, K0, 2, 3, 0); }
// ... and kd_lidar_l2_dgrad ([rows][2][K1])
int64_t kd_lidar_l2_dgrad_stat_rows(int64_t M, int N2, int K1) { return kd_pwconv_stat_rows_for(M, N2, K1, 4, 2, 0); }

// m1_out (optional, [4][K0]) = sum_m G0[m][:] * pts[m][j]: with it (and its workspace m1_ws of
// kd_lidar_l1_dgrad_ws_bytes) the layer-0 weight gradient needs G0 only through these moments, and G0 may be NULL.
int kd_lidar_l1_dgrad(const float* G, int64_t ldg, const float* Y1, int64_t ldy, const float* al, const float* be,
                      const float* ga, const float* msc, const float* msh, int mact, const float* Wt, float* G0,
                      int64_t ldg0, const float* pts, const float* w0, const float* b0, const float* sc0,
                      const float* sh0, const float* mean0, const float* invstd0, int act0, float* partial,
                      int64_t partial_rows, float* m1_out, void* m1_ws, size_t m1_ws_bytes, int64_t M, int N1, int K0,
                      void* stream) {
  KD_REQUIRE(G && Y1 && al && be && ga && Wt && (G0 || m1_out) && pts && w0 && b0 && sc0 && sh0 && mean0 && invstd0 && partial && M > 0,
             KD_ERR_ARG, "kd_lidar_l1_dgrad: bad args");
  KD_REQUIRE(!m1_out || (m1_ws && m1_ws_bytes >= kd_lidar_l1_dgrad_ws_bytes(M, K0)), KD_ERR_WORKSPACE,
             "kd_lidar_l1_dgrad: moment workspace missing or too small");
  KD_REQUIRE(mact == KD_ACT_NONE || (msc && msh), KD_ERR_ARG, "kd_lidar_l1_dgrad: mask needs sc/sh");
  KD_REQUIRE(M < (int64_t)1 << 31 && N1 % 4 == 0 && K0 % 4 == 0 && ldg % 4 == 0 && ldy % 4 == 0 && ldg0 % 4 == 0, KD_ERR_SHAPE,
             "kd_lidar_l1_dgrad: channel counts and strides must be multiples of 4");
  if (!msc) { msc = al; msh = al; }
  GemmArgs g{G, ldg, Y1, ldy, al, be, ga, msc, msh, 2, mact, Wt, nullptr, G0, ldg0, nullptr, 0,
             pts, 4, sc0, sh0, mean0, invstd0, act0, partial, (int)M, N1, K0, nullptr, w0, b0,
             nullptr, nullptr, nullptr, m1_out ? (float*)m1_ws : nullptr};
  const int sr = g_gemm_split.load(std::memory_order_relaxed) ? kd_gemm_stream_stat_rows(M, N1, K0, 2, 3, false) : 0;
  const int rc = gemm_launch(g, 2, 3, (hipStream_t)stream, partial_rows);
  if (rc || !m1_out) return rc;
  // one slab row per wave of the streaming kernel, else per output tile: 256 rows when the output is at most 64 wide
  // (see gemm_launch), else 128
  const bool tall = ((K0 - 1) % 128) < 64;
  const int nrb = sr > 0 ? sr : (int)(tall ? (M + 255) / 256 : (M + 127) / 128);
  return kd_slab_reduce_tall_launch((float*)m1_ws, nrb, (int64_t)4 * K0, m1_out, (hipStream_t)stream);
}

// weight gradient dW1[N,K] = dy1_eff[M,N]^T . act0(bn0(l0(pts)))[M,K]
int kd_lidar_l1_wgrad(const float* D, int64_t ldd, const float* X, int64_t ldx, int d_act, const float* al,
                      const float* be, const float* ga, const float* msc, const float* msh, const float* pts,
                      const float* w0, const float* b0, const float* sc0, const float* sh0, int act0, float* dW,
                      int64_t M, int N, int K, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(D && X && al && be && ga && pts && w0 && b0 && sc0 && sh0 && dW && ws && M > 0, KD_ERR_ARG, "kd_lidar_l1_wgrad: bad args");
  KD_REQUIRE(d_act == KD_ACT_NONE || (msc && msh), KD_ERR_ARG, "kd_lidar_l1_wgrad: mask needs sc/sh");
  KD_REQUIRE(M < (int64_t)1 << 31 && N % 4 == 0 && K % 4 == 0 && ldd % 4 == 0 && ldx % 4 == 0, KD_ERR_SHAPE, "kd_lidar_l1_wgrad: N, K, ld must be multiples of 4");
  if (!msc) { msc = al; msh = al; }
  WgradArgs g{D, ldd, X, ldx, al, be, ga, msc, msh, 2, d_act, pts, 4, sc0, sh0, 2, act0, (float*)ws, (int)M, N, K, 0, w0, b0};
  return wgrad_launch(g, ws_bytes, dW, (hipStream_t)stream);
}

// Last point-MLP layer in eval mode fused with the BEV scatter-max: grid[cell_idx[p], :] = max(grid, act2(bn2(
// act1(bn1(A[p])) . W2^T + bias2))) for rows p < *m_dev; the [points, N] layer output is never written.
int kd_lidar_l2_fwd_scatter(const float* A, int64_t lda, const float* sc1, const float* sh1, int act1, const float* W2,
                            const float* bias2, const float* sc2, const float* sh2, int act2, const int* cell_idx,
                            float* grid, int64_t ncells, int64_t M, int K, int N, const int* m_dev, void* stream) {
  KD_REQUIRE(A && sc1 && sh1 && W2 && sc2 && sh2 && cell_idx && grid && ncells > 0 && M > 0 && K >= 4 && N > 0, KD_ERR_ARG,
             "kd_lidar_l2_fwd_scatter: bad args");
  KD_REQUIRE(M < (int64_t)1 << 31 && K % 4 == 0 && lda % 4 == 0 && N % 4 == 0, KD_ERR_SHAPE, "kd_lidar_l2_fwd_scatter: K, N, lda must be multiples of 4");
  KD_REQUIRE(act2 == KD_ACT_RELU || act2 == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_l2_fwd_scatter: needs a non-negative activation");
  KD_REQUIRE(kd_aligned16(A) && kd_aligned16(W2) && kd_aligned16(sc1) && kd_aligned16(sh1), KD_ERR_ALIGN, "kd_lidar_l2_fwd_scatter: 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grid, 0, (size_t)ncells * N * sizeof(float), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_l2_fwd_scatter: memset failed: %s", hipGetErrorString(e));
  GemmArgs g{A, lda, nullptr, 0, sc1, sh1, nullptr, nullptr, nullptr, 1, act1, W2, bias2, grid, N, nullptr, 0,
             reinterpret_cast<const float*>(cell_idx), 0, sc2, sh2, nullptr, nullptr, act2, nullptr, (int)M, K, N, m_dev,
             nullptr, nullptr};
  return gemm_launch(g, 1, 4, st);
}

// ---- last point-MLP layer, training backward, with the scatter-max gradient G never materialised ---------------
// (rows sorted by cell: kd_lidar_sort_points; tables from kd_lidar_seg_max_fwd (grid) / kd_lidar_seg_share_bwd (share))
//   G[m][c]   = (rows[m] >= 0 && v > 0 && v == grid[rows[m]][c]) ? share[rows[m]][c] : 0,  v = act2(Y2[m][c]*sc2[c] + sh2[c])
//   dy2_eff   = al*G + be*Y2 + ga                                   (BatchNorm-2 backward folded in)
// data gradient: G1[M,K1] = (dy2_eff . W2) * act1'(Y1*sc1+sh1) with the BN1-backward sums in `partial` (Wt = W2^T [K1][N2])
int kd_lidar_l2_dgrad(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share, const float* al,
                      const float* be, const float* ga, const float* sc2, const float* sh2, int act2, const float* Wt,
                      float* G1, int64_t ldg1, const float* Y1, int64_t ldy1, const float* sc1, const float* sh1,
                      const float* mean1, const float* invstd1, int act1, float* partial, int64_t partial_rows, int64_t M,
                      int N2, int K1, void* stream) {
  KD_REQUIRE(Y2 && rows && grid && share && al && be && ga && sc2 && sh2 && Wt && G1 && Y1 && sc1 && sh1 && mean1 && invstd1 && partial && M > 0,
             KD_ERR_ARG, "kd_lidar_l2_dgrad: bad args");
  KD_REQUIRE(M < (int64_t)1 << 31 && N2 % 4 == 0 && K1 % 4 == 0 && ldy2 % 4 == 0 && ldg1 % 4 == 0 && ldy1 % 4 == 0 && N2 >= 4, KD_ERR_SHAPE,
             "kd_lidar_l2_dgrad: channel counts and strides must be multiples of 4");
  KD_REQUIRE(kd_aligned16(Y2) && kd_aligned16(grid) && kd_aligned16(share) && kd_aligned16(Wt) && kd_aligned16(G1) && kd_aligned16(Y1),
             KD_ERR_ALIGN, "kd_lidar_l2_dgrad: 16-byte alignment");
  GemmArgs g{Y2, ldy2, nullptr, 0, al, be, ga, sc2, sh2, 4, act2, Wt, nullptr, G1, ldg1, nullptr, 0,
             Y1, ldy1, sc1, sh1, mean1, invstd1, act1, partial, (int)M, N2, K1, nullptr, nullptr, nullptr, grid, share, rows};
  return gemm_launch(g, 4, 2, (hipStream_t)stream, partial_rows);
}

// weight gradient dW2[N2,K1] = dy2_eff[M,N2]^T . act1(Y1*sc1+sh1)[M,K1]
int kd_lidar_l2_wgrad(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share, const float* al,
                      const float* be, const float* ga, const float* sc2, const float* sh2, int act2, const float* Y1,
                      int64_t ldy1, const float* sc1, const float* sh1, int act1, float* dW, int64_t M, int N2, int K1,
                      void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(Y2 && rows && grid && share && al && be && ga && sc2 && sh2 && Y1 && sc1 && sh1 && dW && ws && M > 0, KD_ERR_ARG,
             "kd_lidar_l2_wgrad: bad args");
  KD_REQUIRE(M < (int64_t)1 << 31 && N2 % 4 == 0 && K1 % 4 == 0 && ldy2 % 4 == 0 && ldy1 % 4 == 0, KD_ERR_SHAPE,
             "kd_lidar_l2_wgrad: N, K, ld must be multiples of 4");
  KD_REQUIRE(kd_aligned16(Y2) && kd_aligned16(grid) && kd_aligned16(share) && kd_aligned16(Y1) && kd_aligned16(ws), KD_ERR_ALIGN,
             "kd_lidar_l2_wgrad: 16-byte alignment");
  WgradArgs g{Y2, ldy2, nullptr, 0, al, be, ga, sc2, sh2, 3, act2, Y1, ldy1, sc1, sh1, 1, act1, (float*)ws,
              (int)M, N2, K1, 0, nullptr, nullptr, grid, share, rows};
  return wgrad_launch(g, ws_bytes, dW, (hipStream_t)stream);
}

// out[c][r] = in[r][c] -- used once per step per weight to get W^T for the dgrad GEMM.
int kd_transpose(const float* in, float* out, int R, int Cc, void* stream) {
  KD_REQUIRE(in && out && R > 0 && Cc > 0, KD_ERR_ARG, "kd_transpose: bad args");
  hipLaunchKernelGGL(transpose_kernel, dim3((R * Cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, R, Cc);
  return kd_check_launch("kd_transpose");
}

// kd_transpose for n matrices at once.  table: device int64 [n][5] = {in pointer, out pointer, R, Cc, index of the matrix's first
// 256-element block}, first-block indices ascending from 0; nblocks = their total (every matrix owns ceil(R*Cc/256) blocks).
int kd_transpose_batch(const int64_t* table, int n, int nblocks, void* stream) {
  KD_REQUIRE(table && n > 0 && nblocks >= n, KD_ERR_ARG, "kd_transpose_batch: bad args");
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, table, n);
  return kd_check_launch("kd_transpose_batch");
}

}  // extern "C"
