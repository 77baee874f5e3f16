// kd_gemm_stream_kernel.h -- the weight-resident STREAMING 1x1-convolution GEMM kernel (split bf16x3 arithmetic, gfx950).
// Included by the translation units that instantiate it (kd_gemm_stream.hip: forward shapes, kd_gemm_stream_bwd0.hip /
// kd_gemm_stream_bwd2.hip: the data-gradient shapes); the dispatch logic lives in kd_gemm_stream.hip.
//
// The dense layers of this network are skinny: M = B*H*W or B*N_points rows in the millions, K and N in 32..768
// (reference: camera_encoder.py:24,39, fusion_module.py:12,29,116, lidar_encoder.py:29,32).  A 128x128-tile kernel
// spends such a GEMM in prologues and epilogues: K / 32 = 1..4 K-steps per tile, two barriers each, operands staged
// through LDS, the result staged through LDS again.  Here the roles are turned round:
//
//   * W (at most ~150 KB as three bf16 planes) is split ONCE per workgroup and stays in LDS for the whole launch, in the
//     [n][k] order the MFMA B operand wants, 16-byte chunks XOR-swizzled so every ds_read_b128 is conflict-free;
//   * each WAVE owns 32-row slabs of the activation matrix, taken round-robin from one global stream (wave w of the
//     launch handles slabs w, w + W, w + 2W, ...): no barrier after the prologue, no LDS for A at all; a workgroup is
//     8 waves (two per SIMD) on one CU, so one wave's loads and epilogue hide behind its partner's MFMAs;
//   * a lane loads its own MFMA A-operand bytes straight from HBM (lane (r, h) of v_mfma_f32_32x32x16_bf16 holds
//     A[row r][k = 16u + 8h .. +7]: two float4 of row r), applies the operand transform -- deferred BatchNorm +
//     activation (PRO1), LiDAR layer 0 recomputed from the 16-byte point (PRO3), the BatchNorm-backward operand
//     al*(D*mask) + be*X + ga from TWO streamed tensors (PRO2), the same with the scatter-max gradient rebuilt from the
//     per-cell tables (PRO4: three streamed tensors) -- and cuts the eight values into three bf16x8 fragments in registers.  That VALU work sits
//     in the shadow of the MFMAs of the previous k-step;
//   * a slab is walked in K-CHUNKS of 32*KC columns, one chunk of every streamed tensor per register set.  With two
//     register sets (DB) the loads of chunk c + 1 are issued BEFORE the k-loop of chunk c and have its 1 500-3 000
//     matrix-pipe cycles to land (the loaded HBM latency is ~4 000 cycles; the rest is covered by the partner wave);
//     with one set the next unit's loads are issued after the k-loop and fly during the epilogue only.  Two sets of 32
//     columns cost the registers of one set of 64: that is what lets the two-tensor data-gradient prologue fit the
//     256-register budget of two waves per SIMD, and it runs the K = 128 forward layers 5-15 % faster than one
//     whole-slab load (profiles/r02_stream_vs_tiled.txt);
//   * the accumulator tile goes to HBM directly: register q of a 32x32 block is one 128-byte row segment per half
//     wave (full-rate dword stores); the data-gradient epilogue (EPI2) reads the raw tensor whose activation is
//     differentiated in the same layout.  BatchNorm statistics are per-lane column sums kept in registers ACROSS the slabs
//     of the wave, combined over the workgroup's eight waves through LDS at the end and written once per launch: the statistics
//     slab has one row per workgroup (<= 256) instead of one per 128 matrix rows (160 000 for the LiDAR layers), deterministic
//     because the slab -> wave map and the order of the final sum are fixed.
//
// Arithmetic: identical to pw_gemm_kernel<.., SPLIT = true> (same pieces, same six products in the same order per
// k-step, k-steps in order), so the raw outputs are bit-identical to the tiled kernel's; statistics are summed in a
// different (fixed) order.
#pragma once
#include "kd_gemm_args.h"

#include <type_traits>

namespace kd_stream {

typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// Dev build only (-DKD_STREAM_DBG): per-phase s_memtime totals over all waves, read back by tools/bench_stream through
// kd_stream_dbg_read: [0] load wait + first conversion, [1] k-loop, [2] next loads + epilogue, [3] units, [4] s_memrealtime
#ifdef KD_STREAM_DBG
extern __device__ unsigned long long kd_stream_dbg[8];
#define KD_SSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define KD_SSTAMP(i) do {} while (0)
#endif

constexpr int SW = 8;                 // waves per workgroup: two per SIMD share the resident W planes

// 16-byte chunk swizzle of a W row of CPR chunks (CPR = K / 8): the 16 lanes of a ds_read_b128 group read the same
// logical chunk q of 16 different rows n; their physical chunks must fall into 16 different 16-byte slots of the
// 256-byte bank row (MI355X_MICROARCH.md, LDS).
template <int CPR>
__device__ __forceinline__ int sw_key(int n) {
  if constexpr (CPR % 16 == 0) return n & 15;
  else if constexpr (CPR % 16 == 8) return (n >> 1) & 7;
  else { static_assert(CPR == 4, "K must be 32 or a multiple of 64"); return (n >> 2) & 3; }
}

// KD_STREAM_TSTORE (default 1): a finished 32 x 32 accumulator block leaves through a wave-private LDS tile (rows padded to 36 floats;
// LDS is in-order per wave: no barrier) as four 16-byte-per-lane stores of eight whole 128-byte row segments each, instead of
// sixteen dword stores (two 128-byte segments each).  =0 builds the dword form (A/B: profiles/r04_stream_tstore_ab.txt).
#ifndef KD_STREAM_TSTORE
#define KD_STREAM_TSTORE 1
#endif
constexpr bool STREAM_TSTORE = KD_STREAM_TSTORE != 0;
// Which instances use it, from the in-step per-launch A/B (same box, tools/r4_ab_stream.sh, profiles/r04_stream_tstore_ab.txt): the
// 128-wide column tiles gain 2-6 % (128 -> 128: 236 -> 224 us, LiDAR layer 2: 4581 -> 4428 us) and so do the 32 -> 192 expand launches
// (three 64-wide tiles: 787 -> 748 us); the narrow outputs (N = 32 / 64 in one tile) and the layer-0-recompute launch lose 2-8 %
// (their dword stores already write whole 128-byte segments; the transposition only adds LDS traffic); the mask epilogue (EPI 2)
// sits at the 256-register limit and spills with the tile.
constexpr bool stream_tstore(int kb, int nb, int pro, int epi) {
  return STREAM_TSTORE && epi != 2 && pro != 3 && (nb == 4 || (nb == 2 && kb == 1));
}
constexpr int STREAM_TR = 32 * 36;                                      // floats per wave-private transposition tile

constexpr int stream_nco(int pro) { return pro == 1 ? 2 : (pro == 3 ? 7 : ((pro == 2 || pro == 4) ? 5 : 0)); }
constexpr size_t stream_lds_bytes(int K, int N, int pro) {
  return (size_t)3 * N * K * 2 + (size_t)(stream_nco(pro) > 0 ? stream_nco(pro) : 1) * K * 4 + (STREAM_TSTORE ? (size_t)8 * STREAM_TR * 4 : 0);
}

// KB = K / 32, KC = chunk width / 32 (divides KB), NB = column-tile width / 32.
// DB: two register sets per streamed tensor -- the loads of chunk c + 1 are issued BEFORE the k-loop of chunk c.
// ADD: g.addend is added (EPI5: after BatchNorm + activation; otherwise to the raw result, before bias and mask).
template <int KB, int KC, int NB, int PRO, int EPI, bool DB = false, bool ADD = false>
__global__ __launch_bounds__(64 * SW, 2) void pw_stream_kernel(GemmArgs g) {
  constexpr int K = 32 * KB, N = 32 * NB, CPR = K / 8;
  constexpr int NCH = KB / KC, NUC = 2 * KC;                             // chunks per slab; 16-wide k-steps per chunk
  static_assert(KB % KC == 0, "the chunk width must divide K");
  constexpr int WPL = N * K;                                             // bf16 per plane
  constexpr int NCO = stream_nco(PRO);
  constexpr int NT = PRO == 2 ? 2 : (PRO == 4 ? 3 : 1);                  // tensors streamed on the A side
  constexpr bool EPI_BWD = EPI == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Wh = reinterpret_cast<unsigned short*>(smem_raw);      // [3][N][K] bf16, swizzled
  float* Co = reinterpret_cast<float*>(smem_raw + 3 * WPL * 2);         // [NCO][K] coefficient tables
  float* trt = Co + (NCO > 0 ? NCO : 1) * K + (threadIdx.x >> 6) * STREAM_TR;   // this wave's transposition tile (STREAM_TSTORE)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * N;                                         // column tile (N_total > N: several tiles)

  // ---- prologue: W tile -> three bf16 planes in LDS, coefficient tables ------------------------------------------------
  for (int i = tid; i < N * K / 4; i += 64 * SW) {
    const int n = i / (K / 4), k4 = i % (K / 4);
    const float4 w = kd_ld4(g.W + (int64_t)(n0 + n) * g.K + k4 * 4);
    uint2 hi, mid, lo;
    kd_split3(w, hi, mid, lo);
    unsigned short* d = Wh + n * K + ((k4 >> 1) ^ sw_key<CPR>(n)) * 8 + (k4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + WPL) = mid;
    *reinterpret_cast<uint2*>(d + 2 * WPL) = lo;
  }
  if (NCO > 0) {
    for (int k = tid; k < K; k += 64 * SW) {
      Co[k] = g.p0[k];
      Co[K + k] = g.p1[k];
      if (PRO == 3) {
        const float4 w0 = kd_ld4(g.l0w + k * 4);
        Co[2 * K + k] = w0.x; Co[3 * K + k] = w0.y; Co[4 * K + k] = w0.z; Co[5 * K + k] = w0.w;
        Co[6 * K + k] = g.l0b[k];
      }
      if (PRO == 2 || PRO == 4) { Co[2 * K + k] = g.p2[k]; Co[3 * K + k] = g.p3[k]; Co[4 * K + k] = g.p4[k]; }
    }
  }
  kd_lds_barrier();

  int64_t M = g.M;
  if (g.m_dev) { const int mv = *g.m_dev; M = mv < g.M ? mv : g.M; }
  const int64_t nslab = (M + 31) / 32;
  const int64_t wtot = (int64_t)gridDim.x * SW, wid = (int64_t)blockIdx.x * SW + wave;

  // per-lane column constants: column of block j is n0 + 32 j + r (N_total is a multiple of the tile width: no column tail)
  constexpr int NE = (EPI == 5 || EPI_BWD) ? NB : 1;
  float bias[NB], esc[NE], esh[NE], emean[EPI_BWD ? NB : 1], einv[EPI_BWD ? NB : 1];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = n0 + 32 * j + r;
    bias[j] = g.bias ? g.bias[c] : 0.f;
    if (EPI == 5 || EPI_BWD) { esc[j] = g.esc[c]; esh[j] = g.esh[c]; }
    if (EPI_BWD) { emean[j] = g.emean[c]; einv[j] = g.einv[c]; }
  }
  constexpr int NS = (EPI == 1 || EPI_BWD) ? NB : 1;
  float s1[NS], s2[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  constexpr int NA = PRO == 3 ? 1 : 2 * NUC;                             // float4 registers per streamed tensor per chunk
  // (plain cache policy on purpose: a lane reads 32 bytes of a 128-byte line per k-step, four instructions touch each line --
  // with the non-temporal hint every one of them went back to HBM: 4x the read traffic, measured)
  // Addresses are (wave-uniform 64-bit base) + (per-lane 32-bit offset that never changes): the base is SALU arithmetic and
  // the loads / stores take it as their scalar operand, so streaming costs no VALU address math.
  const int lda_ = PRO == 3 ? 4 : (int)g.lda;
  const int a_lane = r * lda_ + (PRO == 3 ? 0 : 8 * h);                 // floats
  const int a2_lane = PRO == 2 ? r * (int)g.lda2 + 8 * h : 0;
  // PRO4: the table row of this lane's matrix row (fixed for the slab), fetched one slab ahead
  int trow_cur = 0, trow_nxt = 0;
  auto fetch_trow = [&](int64_t s) {
    int64_t gm = s * 32 + r;
    gm = gm < M ? gm : M - 1;
    return g.trows[gm];
  };
  auto load_unit = [&](int64_t s, int c, f4v (&ra)[NT][NA]) {
    const int64_t m0 = s * 32;
    const bool tail = m0 + 32 > M;                                       // tail slab: clamp (rows >= M are never stored / counted)
    const int64_t gm = m0 + r < M ? m0 + r : M - 1;
    const float* p = tail ? g.A + gm * lda_ + (PRO == 3 ? 0 : 8 * h) : g.A + m0 * lda_ + a_lane;
    p += PRO == 3 ? 0 : 32 * KC * c;
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[0][i] = *reinterpret_cast<const f4v*>(p + 16 * (i >> 1) + 4 * (i & 1));
    if constexpr (PRO == 2) {
      const float* p2 = (tail ? g.A2 + gm * g.lda2 + 8 * h : g.A2 + m0 * g.lda2 + a2_lane) + 32 * KC * c;
#pragma unroll
      for (int i = 0; i < NA; ++i) ra[1][i] = *reinterpret_cast<const f4v*>(p2 + 16 * (i >> 1) + 4 * (i & 1));
    }
    if constexpr (PRO == 4) {
      const int tr = c == 0 ? trow_nxt : trow_cur;                       // chunk 0 of a slab is loaded while the previous slab is current
      const int64_t to = (int64_t)(tr < 0 ? 0 : tr) * K + 8 * h + 32 * KC * c;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        ra[1][i] = *reinterpret_cast<const f4v*>(g.tmx + to + 16 * (i >> 1) + 4 * (i & 1));
        ra[2][i] = *reinterpret_cast<const f4v*>(g.tshare + to + 16 * (i >> 1) + 4 * (i & 1));
      }
    }
  };

  // One pair (two consecutive k) of the A fragment of k-step u of the chunk (ug = its index in the slab): transform, cut
  // into three bf16 pieces.  Pair p covers k = 16 ug + 8 h + 2 p, + 1.  cf: the coefficient float4s of this half k-step
  // (pairs 2e, 2e + 1), read from LDS when the even pair is converted.
  struct Coef { float4 c[NCO > 0 ? NCO : 1]; };
  auto sel = [](const float4& v, int o, int i) { return o ? (i ? v.w : v.z) : (i ? v.y : v.x); };
  auto conv_pair = [&](const f4v (&ra)[NT][NA], const float* Cp, int ug, int u, int p, Coef& cf, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
    const int e = p >> 1, o = (p & 1) * 2;
    if (NCO > 0 && (p & 1) == 0) {
      const int kb = 16 * ug + 8 * h + 4 * e;
#pragma unroll
      for (int t = 0; t < NCO; ++t) cf.c[t] = kd_ld4(Cp + t * K + kb);
    }
    float v[2];
    if constexpr (PRO == 3) {
      const float4 pt = make_float4(ra[0][0][0], ra[0][0][1], ra[0][0][2], ra[0][0][3]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 w = make_float4(sel(cf.c[2], o, i), sel(cf.c[3], o, i), sel(cf.c[4], o, i), sel(cf.c[5], o, i));
        v[i] = kd_act(kd_affine(kd_l0_raw(pt, w, sel(cf.c[6], o, i)), sel(cf.c[0], o, i), sel(cf.c[1], o, i)), g.pro_act);
      }
    } else {
      const f4v x = ra[0][2 * u + e];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        v[i] = x[o + i];
        if constexpr (PRO == 1) v[i] = kd_act(kd_affine(v[i], sel(cf.c[0], o, i), sel(cf.c[1], o, i)), g.pro_act);
        if constexpr (PRO == 2)
          v[i] = kd_bwd_operand(v[i], ra[1][2 * u + e][o + i], sel(cf.c[0], o, i), sel(cf.c[1], o, i), sel(cf.c[2], o, i), sel(cf.c[3], o, i),
                                sel(cf.c[4], o, i), g.pro_act);
        if constexpr (PRO == 4) {
          const float y = v[i], mx = ra[1][2 * u + e][o + i], sv = ra[2][2 * u + e][o + i];
          const float a = kd_act(kd_affine(y, sel(cf.c[3], o, i), sel(cf.c[4], o, i)), g.pro_act);
          v[i] = kd_bwd_operand((trow_cur >= 0 && a > 0.f && a == mx) ? sv : 0.f, y, sel(cf.c[0], o, i), sel(cf.c[1], o, i), sel(cf.c[2], o, i),
                                0.f, 0.f, KD_ACT_NONE);
        }
      }
    }
    kd_split_pair(v[0], v[1], ph, pm, pl);
  };
  struct Frag { uint32_t p[3][4]; };   // the three bf16x8 planes of one operand fragment
  auto plane = [](const Frag& f, int i) { const u32x4 v = {f.p[i][0], f.p[i][1], f.p[i][2], f.p[i][3]}; return __builtin_bit_cast(bf16x8, v); };
  auto conv_all = [&](const f4v (&ra)[NT][NA], const float* Cp, int ug, Frag& a) {
    Coef cf;
#pragma unroll
    for (int p = 0; p < 4; ++p) conv_pair(ra, Cp, ug, 0, p, cf, a.p[0][p], a.p[1][p], a.p[2][p]);
  };
  const int fsw = h ^ sw_key<CPR>(r);                                    // physical chunk of k-step ug = (2 ug) ^ fsw
  auto load_b = [&](const unsigned short* Wp, int ug, int j, Frag& b) {
    const unsigned short* bp = Wp + (32 * j + r) * K + ((2 * ug) ^ fsw) * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(bp + p * WPL);
      b.p[p][0] = v[0]; b.p[p][1] = v[1]; b.p[p][2] = v[2]; b.p[p][3] = v[3];
    }
  };

  // ---- one chunk: NUC k-steps x NB column blocks, software-pipelined: while the six MFMAs of step (u, j) run, the B
  // fragment of the next step is read from LDS and a share of the NEXT k-step's A fragment is converted. -----------------
  auto compute_chunk = [&](const f4v (&ra)[NT][NA], int c, Frag& a_cur, Frag& b_cur, f32x16 (&acc)[NB]) {
    // W fragments and coefficient tables do not depend on the slab: left alone, the compiler hoists their LDS reads out
    // of the stream loop and tries to keep the whole of W in registers.  An opaque zero pins them inside the chunk.
    int pin = 0;
    asm volatile("" : "+v"(pin));
    const unsigned short* Wp = Wh + pin;
    const float* Cp = Co + pin;
    const int ug0 = c * NUC, ugn = (c == NCH - 1) ? 0 : (c + 1) * NUC;   // first k-step of this chunk / of the next unit
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (as pw_gemm_kernel)
#pragma unroll
    for (int u = 0; u < NUC; ++u) {
      Frag a_nxt;
      Coef cf;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        Frag b_nxt;
        const bool last = u == NUC - 1 && j == NB - 1;
        load_b(Wp, last ? ugn : (j == NB - 1 ? ug0 + u + 1 : ug0 + u), (last || j == NB - 1) ? 0 : j + 1, b_nxt);
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p * NB / 4 == j) {
            if (u < NUC - 1) conv_pair(ra, Cp, ug0 + u + 1, u + 1, p, cf, a_nxt.p[0][p], a_nxt.p[1][p], a_nxt.p[2][p]);
          }
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plane(a_cur, PA[t]), plane(b_cur, PB[t]), acc[j], 0, 0, 0);
        b_cur = b_nxt;
      }
      if (u < NUC - 1) a_cur = a_nxt;
    }
  };

  // ---- epilogue: register q of a block is row (q & 3) + 8 (q >> 2) + 4 h, column r -- one dword store per register writes
  // two 128-byte row segments (rows R and R + 4), full-rate for plain stores (MI355X_MICROARCH.md).  Addresses are a
  // wave-uniform base plus a per-lane offset that never changes.  (A variant that swapped values between neighbouring lanes
  // to store 8 bytes per lane halved the store count but cost two DPP moves and two selects per pair: slower, not kept.)
  const int c_lane = 4 * h * (int)g.ldc + r;                             // floats, per lane, fixed
  const int ad_lane = 4 * h * (int)g.ldadd + r;
  const int x_lane = EPI == 2 ? 4 * h * (int)g.ldx + r : 0;
  auto store_slab = [&](int64_t s, f32x16 (&acc)[NB], auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;              // every row of the slab is < M: no predicates at all
    constexpr int QG = NB >= 4 ? 8 : 16;         // (the split that keeps hipcc inside the register budget without spills)
    const int64_t m0 = s * 32;
    const float* addend = ADD ? g.addend : nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float* cbase = g.C + m0 * g.ldc + n0 + 32 * j;               // wave-uniform
      const float* abase = ADD ? addend + m0 * g.ldadd + n0 + 32 * j : nullptr;
      const float* xbase = EPI == 2 ? g.X + m0 * g.ldx + n0 + 32 * j : nullptr;
#pragma unroll
      for (int hq = 0; hq < 16 / QG; ++hq) {                       // QG registers at a time: their side loads first, then the stores
        float ad[ADD ? QG : 1], xr[EPI == 2 ? QG : 1];
        if constexpr (ADD) {                                             // residual values, in the layout of the stores below
#pragma unroll
          for (int qi = 0; qi < QG; ++qi) {
            const int q = QG * hq + qi, rbase = (q & 3) + 8 * (q >> 2);
            const bool rok = FULL || (m0 + rbase + 4 * h < M);
            ad[qi] = rok ? abase[(int64_t)rbase * g.ldadd + ad_lane] : 0.f;
          }
        }
        if constexpr (EPI == 2) {                                 // the raw tensor whose activation is differentiated
#pragma unroll
          for (int qi = 0; qi < QG; ++qi) {
            const int q = QG * hq + qi, rbase = (q & 3) + 8 * (q >> 2);
            const bool rok = FULL || (m0 + rbase + 4 * h < M);
            xr[qi] = rok ? xbase[(int64_t)rbase * g.ldx + x_lane] : 0.f;
          }
        }
#pragma unroll
        for (int qi = 0; qi < QG; ++qi) {
          const int q = QG * hq + qi, rbase = (q & 3) + 8 * (q >> 2);   // row of register q within the slab, before + 4 h
          const bool rok = FULL || (m0 + rbase + 4 * h < M);
          float v = acc[j][q];
          if constexpr (EPI != 5 && ADD) v += ad[qi];                     // gradient of the residual branch: before the bias and the mask
          v += bias[j];
          if (EPI == 1 && rok) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
          if (EPI == 5) {
            v = kd_act(kd_affine(v, esc[j], esh[j]), g.epi_act);
            if constexpr (ADD) v += ad[qi];
          }
          if constexpr (EPI == 2) {
            const float x = xr[qi];
            v *= kd_act_mask(kd_affine(x, esc[j], esh[j]), g.epi_act);
            if (rok) { s1[j] += v; s2[j] = fmaf(v, (x - emean[j]) * einv[j], s2[j]); }
          }
          if constexpr (stream_tstore(KB, NB, PRO, EPI)) {
            trt[(rbase + 4 * h) * 36 + r] = v;
          } else {
            float* dst = cbase + (int64_t)rbase * g.ldc + c_lane;
            if (FULL) {
              if (g.nt_store) __builtin_nontemporal_store(v, dst); else *dst = v;
            } else if (rok) {
              *dst = v;
            }
          }
        }
      }
      if constexpr (stream_tstore(KB, NB, PRO, EPI)) {
        // the block as rows: lane -> columns 4 (lane & 7) .. + 3 of rows (lane >> 3) + 8 p
        const int tc4 = (lane & 7) * 4, tr0 = lane >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int rr = tr0 + 8 * p;
          const float4 v = kd_ld4(trt + rr * 36 + tc4);
          float* dst = cbase + (int64_t)rr * g.ldc + tc4;
          if (FULL || m0 + rr < M) { if (g.nt_store) kd_st4_nt(dst, v); else kd_st4(dst, v); }
        }
      }
    }
  };

  // ---- the stream.  One register set per streamed tensor: a chunk's k-loop consumes it, the next unit's loads (next
  // chunk of the slab, or chunk 0 of the wave's next slab) are issued into it right after the k-loop and fly during the
  // epilogue; what latency is left is covered by the second wave of the SIMD (two waves per SIMD run the same program
  // out of phase).  All waits are hipcc's own (in-order vmcnt: loads, then the stores). ---------------------------------
  static_assert(!DB || (NCH >= 2 && NCH % 2 == 0), "double buffering alternates two register sets over an even number of chunks");
  f4v rc[DB ? 2 : 1][NT][NA];
  Frag a_cur, b_cur;
  int64_t s = wid;
  if (s < nslab) {
    if constexpr (PRO == 4) trow_nxt = fetch_trow(s);
    load_unit(s, 0, rc[0]);
    load_b(Wh, 0, 0, b_cur);
  }
#ifdef KD_STREAM_DBG
  unsigned long long dbg_acc[4] = {0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
  const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (; s < nslab; s += wtot) {
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    const bool more = s + wtot < nslab;
    if constexpr (PRO == 4) {
      trow_cur = trow_nxt;
      if (more) trow_nxt = fetch_trow(s + wtot);
    }
    int pin0 = 0;                                    // (see compute_chunk: keeps the table reads inside the slab)
    asm volatile("" : "+v"(pin0));
    const float* Cq = Co + pin0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if constexpr (DB) {                            // next unit first: it flies during this chunk's whole k-loop
        if (c < NCH - 1) load_unit(s, c + 1, rc[(c + 1) & 1]);
        else if (more) load_unit(s + wtot, 0, rc[0]);
      }
      conv_all(rc[DB ? (c & 1) : 0], Cq, c * NUC, a_cur);
      KD_SSTAMP(0);
      compute_chunk(rc[DB ? (c & 1) : 0], c, a_cur, b_cur, acc);
      KD_SSTAMP(1);
      if constexpr (!DB) {
        if (c < NCH - 1) load_unit(s, c + 1, rc[0]);
        else if (more) load_unit(s + wtot, 0, rc[0]);
      }
    }
    if (s * 32 + 32 <= M) store_slab(s, acc, std::true_type{}); else store_slab(s, acc, std::false_type{});
    KD_SSTAMP(2);
#ifdef KD_STREAM_DBG
    dbg_acc[3] += 1;
#endif
  }
#ifdef KD_STREAM_DBG
  if (lane == 0) {
    for (int i = 0; i < 4; ++i) atomicAdd(&kd_stream_dbg[i], dbg_acc[i]);
    atomicAdd(&kd_stream_dbg[4], __builtin_amdgcn_s_memrealtime() - dbg_r0);
    atomicAdd(&kd_stream_dbg[5], 1ull);
  }
#endif

  if (EPI == 1 || EPI_BWD) {
    // one statistics row per WORKGROUP (round 4; one per wave before): the eight waves' column sums meet in LDS -- the weight
    // planes are dead once every wave has left its slab loop -- and are added in wave order (fixed: deterministic).  The slab
    // the BatchNorm finalize walks shrinks from 2048 to at most 256 rows: that kernel is latency-bound on the rows per lane.
    kd_lds_barrier();
    float* red = reinterpret_cast<float*>(smem_raw);                       // [SW][2][N]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (h == 0) {
        red[(wave * 2 + 0) * N + 32 * j + r] = t1;
        red[(wave * 2 + 1) * N + 32 * j + r] = t2;
      }
    }
    kd_lds_barrier();
    for (int i = tid; i < 2 * N; i += 64 * SW) {
      const int st = i / N, c = i % N;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < SW; ++w) t += red[(w * 2 + st) * N + c];
      g.partial[((int64_t)blockIdx.x * 2 + st) * g.N + n0 + c] = t;
    }
  }
}

// -> 1 launched, < 0 error (the dynamic-LDS limit of this instance could not be raised on the current device)
template <int KB, int KC, int NB, int PRO, int EPI, bool DB = false, bool ADD = false>
inline int stream_launch_one(const GemmArgs& g, dim3 grid, hipStream_t st) {
  constexpr size_t lds = stream_lds_bytes(32 * KB, 32 * NB, PRO);
  static std::atomic<uint64_t> lds_raised{0};          // one bit per device: the attribute is per (function, device)
  const hipError_t e = kd_raise_dynamic_lds((const void*)pw_stream_kernel<KB, KC, NB, PRO, EPI, DB, ADD>, lds, lds_raised);
  if (e != hipSuccess) {
    kd_set_error("kd_gemm_stream: cannot raise the dynamic LDS limit to %zu B for K=%d N=%d pro=%d epi=%d: %s", lds, g.K, g.N, PRO, EPI,
                 hipGetErrorString(e));
    return -3000;
  }
  hipLaunchKernelGGL((pw_stream_kernel<KB, KC, NB, PRO, EPI, DB, ADD>), grid, dim3(64 * SW), lds, st, g);
  return 1;
}

}  // namespace kd_stream
