// kd_gemm_stream.hip -- weight-resident STREAMING form of the 1x1-convolution GEMM (split bf16x3 arithmetic) for gfx950.
//
// The dense layers of this network are skinny: M = B*H*W or B*N_points rows in the millions, K and N in 32..768
// (reference: camera_encoder.py:24,39, fusion_module.py:12,29,116, lidar_encoder.py:29,32).  A 128x128-tile kernel
// spends such a GEMM in prologues and epilogues: K / 32 = 1..4 K-steps per tile, two barriers each, operands staged
// through LDS, the result staged through LDS again.  Here the roles are turned round:
//
//   * W (at most 96 KB as three bf16 planes) is split ONCE per workgroup and stays in LDS for the whole launch, in the
//     [n][k] order the MFMA B operand wants, 16-byte chunks XOR-swizzled so every ds_read_b128 is conflict-free;
//   * each WAVE owns 32-row slabs of the activation matrix, taken round-robin from one global stream (wave w of the
//     launch handles slabs w, w + W, w + 2W, ...): no barrier after the prologue, no LDS for A at all; a workgroup is
//     8 waves (two per SIMD) on one CU, so one wave's loads and epilogue hide behind its partner's MFMAs;
//   * a lane loads its own MFMA A-operand bytes straight from HBM (lane (r, h) of v_mfma_f32_32x32x16_bf16 holds
//     A[row r][k = 16u + 8h .. +7]: two float4 of row r), applies the deferred BatchNorm + activation (PRO1) -- or
//     recomputes LiDAR layer 0 from the 16-byte point (PRO3) -- and cuts the eight values into three bf16x8 fragments
//     in registers.  That VALU work sits in the shadow of the MFMAs of the previous k-step (6 * NB MFMAs of 32 cycles
//     against ~70 four-cycle VALU issues); the next slab's loads are in flight during the whole current slab;
//   * the accumulator tile goes to HBM directly: register r of a 32x32 block is one 128-byte row segment per half
//     wave (full-rate dword stores), BatchNorm statistics are per-lane column sums kept in registers ACROSS the slabs of
//     the wave and written once per launch: the statistics slab has one row per wave (<= 1024) instead of one per 128
//     matrix rows (160 000 for the LiDAR layers), deterministic because the slab -> wave map is fixed.
//
// Arithmetic: identical to pw_gemm_kernel<.., SPLIT = true> (same pieces, same six products in the same order per
// k-step, k-steps in order), so the raw conv outputs are bit-identical to the tiled kernel's.
#include "kd_gemm_args.h"

#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(4))) float f4v;

// Dev build only (-DKD_STREAM_DBG): per-phase s_memtime totals over all waves, read back by tools/bench_stream through
// kd_stream_dbg_read: [0] load wait + first conversion, [1] k-loop, [2] next loads + epilogue, [3] slabs, [4] s_memrealtime
#ifdef KD_STREAM_DBG
__device__ unsigned long long kd_stream_dbg[8];
#define KD_SSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define KD_SSTAMP(i) do {} while (0)
#endif

constexpr int SW = 8;                 // waves per workgroup: two per SIMD share the resident W planes

// 16-byte chunk swizzle of a W row of CPR chunks (CPR = K / 8): the 16 lanes of a ds_read_b128 group read the same
// logical chunk q of 16 different rows n; their physical chunks must fall into 16 different 16-byte slots of the
// 256-byte bank row (MI355X_MICROARCH.md, LDS).
template <int CPR>
__device__ __forceinline__ int sw_chunk(int n, int q) {
  if constexpr (CPR % 16 == 0) return q ^ (n & 15);
  else if constexpr (CPR % 16 == 8) return q ^ ((n >> 1) & 7);
  else { static_assert(CPR == 4, "K must be 32 or a multiple of 64"); return q ^ ((n >> 2) & 3); }
}

// quad_perm [1, 0, 3, 2]: every lane reads its xor-1 neighbour
__device__ __forceinline__ float kd_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

template <int KB, int NB, int PRO, int EPI>
__global__ __launch_bounds__(64 * SW, 2) void pw_stream_kernel(GemmArgs g) {
  constexpr int K = 32 * KB, N = 32 * NB, CPR = K / 8, NU = 2 * KB;     // NU: 16-wide k-steps
  constexpr int WPL = N * K;                                             // bf16 per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Wh = reinterpret_cast<unsigned short*>(smem_raw);      // [3][N][K] bf16, swizzled
  float* Co = reinterpret_cast<float*>(smem_raw + 3 * WPL * 2);         // [7][K] coefficient tables (PRO1: 2, PRO3: 7)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const bool odd = (lane & 1) != 0;
  const int n0 = blockIdx.y * N;                                         // column tile (N_total > N: several tiles)

  // ---- prologue: W tile -> three bf16 planes in LDS, coefficient tables ------------------------------------------------
  for (int i = tid; i < N * K / 4; i += 64 * SW) {
    const int n = i / (K / 4), k4 = i % (K / 4);
    const float4 w = kd_ld4(g.W + (int64_t)(n0 + n) * g.K + k4 * 4);
    uint2 hi, mid, lo;
    kd_split3(w, hi, mid, lo);
    unsigned short* d = Wh + n * K + sw_chunk<CPR>(n, k4 >> 1) * 8 + (k4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + WPL) = mid;
    *reinterpret_cast<uint2*>(d + 2 * WPL) = lo;
  }
  if (PRO == 1 || PRO == 3) {
    for (int k = tid; k < K; k += 64 * SW) {
      Co[k] = g.p0[k];
      Co[K + k] = g.p1[k];
      if (PRO == 3) {
        const float4 w0 = kd_ld4(g.l0w + k * 4);
        Co[2 * K + k] = w0.x; Co[3 * K + k] = w0.y; Co[4 * K + k] = w0.z; Co[5 * K + k] = w0.w;
        Co[6 * K + k] = g.l0b[k];
      }
    }
  }
  kd_lds_barrier();

  int64_t M = g.M;
  if (g.m_dev) { const int mv = *g.m_dev; M = mv < g.M ? mv : g.M; }
  const int64_t nslab = (M + 31) / 32;
  const int64_t wtot = (int64_t)gridDim.x * SW, wid = (int64_t)blockIdx.x * SW + wave;

  // per-lane column constants: column of block j is n0 + 32 j + r (N_total is a multiple of the tile width: no column tail)
  float bias[NB], esc[EPI == 5 ? NB : 1], esh[EPI == 5 ? NB : 1];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int c = n0 + 32 * j + r;
    bias[j] = g.bias ? g.bias[c] : 0.f;
    if (EPI == 5) { esc[j] = g.esc[c]; esh[j] = g.esh[c]; }
  }
  float s1[EPI == 1 ? NB : 1], s2[EPI == 1 ? NB : 1];
#pragma unroll
  for (int j = 0; j < (EPI == 1 ? NB : 1); ++j) { s1[j] = 0.f; s2[j] = 0.f; }

  constexpr int NA = PRO == 3 ? 1 : 2 * NU;                              // float4 registers of A per slab
  // (plain cache policy on purpose: a lane reads 32 bytes of a 128-byte line per k-step, four instructions touch each line --
  // with the non-temporal hint every one of them went back to HBM: 4x the read traffic, measured)
  // Addresses are (wave-uniform 64-bit base) + (per-lane 32-bit offset that never changes): the base is SALU arithmetic and
  // the loads / stores take it as their scalar operand, so streaming costs no VALU address math.
  const int lda_ = PRO == 3 ? 4 : (int)g.lda;
  const int a_lane = r * lda_ + (PRO == 3 ? 0 : 8 * h);                 // floats
  auto load_slab = [&](int64_t s, f4v (&ra)[NA]) {
    const int64_t m0 = s * 32;
    const float* p = g.A + m0 * lda_ + a_lane;
    if (m0 + 32 > M) {                                                   // tail slab: clamp (rows >= M are never stored / counted)
      const int64_t gm = m0 + r < M ? m0 + r : M - 1;
      p = g.A + gm * lda_ + (PRO == 3 ? 0 : 8 * h);
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const f4v*>(p + 16 * (i >> 1) + 4 * (i & 1));
  };

  // One pair (two consecutive k) of the A fragment of k-step u: transform, cut into three bf16 pieces.
  // Pair p covers k = 16 u + 8 h + 2 p, + 1.  cf: the coefficient float4s of this half k-step (pairs 2e, 2e + 1), read
  // from LDS when the even pair is converted.
  struct Coef { float4 sc, sh, wx, wy, wz, ww, b0; };
  auto conv_pair = [&](const f4v (&ra)[NA], const float* Cp, int u, int p, Coef& cf, uint32_t& ph, uint32_t& pm, uint32_t& pl) {
    const int e = p >> 1, o = (p & 1) * 2;
    if ((PRO == 1 || PRO == 3) && (p & 1) == 0) {
      const int kb = 16 * u + 8 * h + 4 * e;
      cf.sc = kd_ld4(Cp + kb); cf.sh = kd_ld4(Cp + K + kb);
      if (PRO == 3) {
        cf.wx = kd_ld4(Cp + 2 * K + kb); cf.wy = kd_ld4(Cp + 3 * K + kb); cf.wz = kd_ld4(Cp + 4 * K + kb);
        cf.ww = kd_ld4(Cp + 5 * K + kb); cf.b0 = kd_ld4(Cp + 6 * K + kb);
      }
    }
    float v0, v1;
    if (PRO == 3) {
      const float4 pt = make_float4(ra[0][0], ra[0][1], ra[0][2], ra[0][3]);
      const float sc0 = o ? cf.sc.z : cf.sc.x, sc1 = o ? cf.sc.w : cf.sc.y, sh0 = o ? cf.sh.z : cf.sh.x, sh1 = o ? cf.sh.w : cf.sh.y;
      const float4 w0 = o ? make_float4(cf.wx.z, cf.wy.z, cf.wz.z, cf.ww.z) : make_float4(cf.wx.x, cf.wy.x, cf.wz.x, cf.ww.x);
      const float4 w1 = o ? make_float4(cf.wx.w, cf.wy.w, cf.wz.w, cf.ww.w) : make_float4(cf.wx.y, cf.wy.y, cf.wz.y, cf.ww.y);
      v0 = kd_act(kd_affine(kd_l0_raw(pt, w0, o ? cf.b0.z : cf.b0.x), sc0, sh0), g.pro_act);
      v1 = kd_act(kd_affine(kd_l0_raw(pt, w1, o ? cf.b0.w : cf.b0.y), sc1, sh1), g.pro_act);
    } else {
      const f4v x = ra[2 * u + e];
      v0 = x[o]; v1 = x[o + 1];
      if (PRO == 1) {
        v0 = kd_act(kd_affine(v0, o ? cf.sc.z : cf.sc.x, o ? cf.sh.z : cf.sh.x), g.pro_act);
        v1 = kd_act(kd_affine(v1, o ? cf.sc.w : cf.sc.y, o ? cf.sh.w : cf.sh.y), g.pro_act);
      }
    }
    kd_split_pair(v0, v1, ph, pm, pl);
  };
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  struct Frag { uint32_t p[3][4]; };   // the three bf16x8 planes of one operand fragment
  auto plane = [](const Frag& f, int i) { const u32x4 v = {f.p[i][0], f.p[i][1], f.p[i][2], f.p[i][3]}; return __builtin_bit_cast(bf16x8, v); };
  auto conv_all = [&](const f4v (&ra)[NA], const float* Cp, int u, Frag& a) {
    Coef cf;
#pragma unroll
    for (int p = 0; p < 4; ++p) conv_pair(ra, Cp, u, p, cf, a.p[0][p], a.p[1][p], a.p[2][p]);
  };
  const int fsw = h ^ (CPR % 16 == 0 ? (r & 15) : (CPR % 16 == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3)));   // chunk(u) = 2u ^ fsw
  auto load_b = [&](const unsigned short* Wp, int u, int j, Frag& b) {
    const unsigned short* bp = Wp + (32 * j + r) * K + ((2 * u) ^ fsw) * 8;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(bp + p * WPL);
      b.p[p][0] = v[0]; b.p[p][1] = v[1]; b.p[p][2] = v[2]; b.p[p][3] = v[3];
    }
  };

  // ---- one slab: NU k-steps x NB column blocks, software-pipelined: while the six MFMAs of step (u, j) run, the B fragment
  // of the next step is read from LDS and a share of the NEXT k-step's A fragment is converted. ---------------------------------
  auto compute_slab = [&](const f4v (&ra)[NA], Frag& a_cur, Frag& b_cur, f32x16 (&acc)[NB]) {
    // W fragments and coefficient tables do not depend on the slab: left alone, the compiler hoists their LDS reads out
    // of the stream loop and tries to keep the whole of W in registers.  An opaque zero pins them inside the slab.
    int pin = 0;
    asm volatile("" : "+v"(pin));
    const unsigned short* Wp = Wh + pin;
    const float* Cp = Co + pin;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest terms first (as pw_gemm_kernel)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      Frag a_nxt;
      Coef cf;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        Frag b_nxt;
        const bool last = u == NU - 1 && j == NB - 1;
        load_b(Wp, last ? 0 : (j == NB - 1 ? u + 1 : u), last ? 0 : (j == NB - 1 ? 0 : j + 1), b_nxt);
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p * NB / 4 == j) {
            if (u < NU - 1) conv_pair(ra, Cp, u + 1, p, cf, a_nxt.p[0][p], a_nxt.p[1][p], a_nxt.p[2][p]);
          }
#ifdef KD_S_NOMFMA          // ablation build: operands kept alive, matrix pipe idle
        asm volatile("" :: "v"(a_cur.p[0][0]), "v"(a_cur.p[1][1]), "v"(a_cur.p[2][2]), "v"(b_cur.p[0][0]), "v"(b_cur.p[1][1]), "v"(b_cur.p[2][2]));
        acc[j][0] += __builtin_bit_cast(float, a_cur.p[0][3] ^ b_cur.p[0][3]);
#else
#pragma unroll
        for (int t = 0; t < 6; ++t)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(plane(a_cur, PA[t]), plane(b_cur, PB[t]), acc[j], 0, 0, 0);
#endif
        b_cur = b_nxt;
      }
      if (u < NU - 1) a_cur = a_nxt;
    }
  };

  // ---- epilogue: register q of a block is row (q & 3) + 8 (q >> 2) + 4 h, column r.  Neighbouring lanes swap one value of
  // each register pair so that a lane stores 8 contiguous bytes: 128-byte row segments, 32 stores per slab instead of 64
  // (vmcnt counts stores too: with at most 16 loads + 32 stores younger than a slab's first load the waits stay exact). -----
  // Epilogue: register q of a block is row (q & 3) + 8 (q >> 2) + 4 h, column r -- one dword store per register writes two
  // 128-byte row segments (rows R and R + 4), full-rate for plain stores (MI355X_MICROARCH.md).  Addresses are a wave-uniform
  // base plus a per-lane offset that never changes.  (A variant that swapped values between neighbouring lanes to store
  // 8 bytes per lane halved the store count but cost two DPP moves and two selects per pair: slower, not kept.)
  const int c_lane = 4 * h * (int)g.ldc + r;                             // floats, per lane, fixed
  const int ad_lane = 4 * h * (int)g.ldadd + r;
  auto store_slab = [&](int64_t s, f32x16 (&acc)[NB], auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;              // every row of the slab is < M: no predicates at all
    const int64_t m0 = s * 32;
    const float* addend = EPI == 5 ? g.addend : nullptr;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float* cbase = g.C + m0 * g.ldc + n0 + 32 * j;               // wave-uniform
      const float* abase = addend ? addend + m0 * g.ldadd + n0 + 32 * j : nullptr;
      float ad[16];
      if (EPI == 5 && addend) {                                   // residual values, in the layout of the stores below
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int rbase = (q & 3) + 8 * (q >> 2);
          const bool rok = FULL || (m0 + rbase + 4 * h < M);
          ad[q] = rok ? abase[(int64_t)rbase * g.ldadd + ad_lane] : 0.f;
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rbase = (q & 3) + 8 * (q >> 2);                  // row of register q within the slab, before + 4 h
        const bool rok = FULL || (m0 + rbase + 4 * h < M);
        float v = acc[j][q] + bias[j];
        if (EPI == 1 && rok) { s1[j] += v; s2[j] = fmaf(v, v, s2[j]); }
        if (EPI == 5) {
          v = kd_act(kd_affine(v, esc[j], esh[j]), g.epi_act);
          if (addend) v += ad[q];
        }
        float* dst = cbase + (int64_t)rbase * g.ldc + c_lane;
        if (FULL) {
          if (g.nt_store) __builtin_nontemporal_store(v, dst); else *dst = v;
        } else if (rok) {
          *dst = v;
        }
      }
    }
  };

  // ---- the stream.  One register set: the k-loop consumes it, the next slab's loads are issued into it right after the
  // k-loop and fly during the epilogue; what latency is left is covered by the second wave of the SIMD (two waves per
  // SIMD run the same program half a slab apart).  All waits are hipcc's own (in-order vmcnt: loads, then the stores). ------
  f4v rc[NA];
  Frag a_cur, b_cur;
  int64_t s = wid;
  if (s < nslab) {
    load_slab(s, rc);
    load_b(Wh, 0, 0, b_cur);
  }
#ifdef KD_STREAM_DBG
  unsigned long long dbg_acc[4] = {0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
  const unsigned long long dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (; s < nslab; s += wtot) {
    f32x16 acc[NB];
    conv_all(rc, Co, 0, a_cur);
    KD_SSTAMP(0);
    compute_slab(rc, a_cur, b_cur, acc);
    KD_SSTAMP(1);
    if (s + wtot < nslab) load_slab(s + wtot, rc);
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

  if (EPI == 1) {        // one statistics row per wave: [wid][2][N_total]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float t1 = s1[j] + __shfl_xor(s1[j], 32, 64), t2 = s2[j] + __shfl_xor(s2[j], 32, 64);
      if (h == 0) {
        g.partial[(wid * 2 + 0) * g.N + n0 + 32 * j + r] = t1;
        g.partial[(wid * 2 + 1) * g.N + n0 + 32 * j + r] = t2;
      }
    }
  }
}

std::atomic<int> g_stream_on{-1};

int stream_mode() {          // 0 off, 1 only the shapes that win in isolation, 2 every covered shape (default)
  int v = g_stream_on.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("KD_GEMM_STREAM");
    v = (e && e[0] == '0') ? 0 : ((e && e[0] == '1') ? 1 : 2);
    g_stream_on.store(v, std::memory_order_relaxed);
  }
  return v;
}
bool stream_enabled() { return stream_mode() != 0; }
bool stream_all() { return stream_mode() == 2; }

// shapes with an instantiated kernel: (K/32, N-tile/32)
struct StreamCfg { int kb, nb, ntiles; };

bool stream_cfg(int K, int N, int pro, int epi, StreamCfg& c) {
  if (!(pro == 0 || pro == 1 || pro == 3) || !(epi == 0 || epi == 1 || epi == 5)) return false;
  if (pro == 3 && epi == 5) return false;
  if (K % 32 != 0 || N % 32 != 0) return false;
  const int kb = K / 32, nb = N / 32;
  if (!(kb == 1 || kb == 2 || kb == 4) || !(nb == 1 || nb == 2 || nb == 4)) return false;
  if (pro == 3 && kb != 2) return false;
  c = {kb, nb, 1};
  // Measured on an MI355X: per shape at M = 32 frames the streaming form runs 1.0-1.45x the tiled kernel (profiles/
  // r02_stream_vs_tiled.txt; only the 64 -> 128 layers with a BatchNorm prologue at multi-million M lose, 0.8-0.9x), and in
  // the whole KD step at 256 frames "every covered shape" beats "only where the micro-benchmark wins" (95.5 vs 96.6 ms
  // per step; 97.1 ms with the tiled kernels alone) -- so every covered shape is the default; mode 1 keeps the narrow rule.
  if (stream_all()) return true;
  return nb <= 2 || (kb == 4 && nb == 4) || (pro == 3 && epi == 1);
}

int stream_grid(int64_t M, int ntiles) {
  // one workgroup per CU (the W planes take most of the LDS), fewer when there are not enough 32-row slabs for every wave
  int64_t want = (M + 32 * SW - 1) / (32 * SW);
  int cap = 256 / ntiles;
  if (cap < 1) cap = 1;
  return (int)(want < cap ? want : cap);
}

}  // namespace

int kd_gemm_stream_stat_rows(int64_t M, int K, int N, int pro, int epi) {
  StreamCfg c;
  if (!stream_enabled() || !stream_cfg(K, N, pro, epi, c)) return 0;
  return stream_grid(M, c.ntiles) * SW;
}

template <int KB, int NB>
static int stream_dispatch(GemmArgs& g, int pro, int epi, dim3 grid, hipStream_t st) {
  constexpr int K = 32 * KB, N = 32 * NB;
  const size_t lds = (size_t)3 * N * K * 2 + (size_t)7 * K * 4;
#define KD_SCASE(P_, E_)                                                                                          \
  if (pro == P_ && epi == E_) {                                                                                   \
    static bool attr_set = false;                                                                                 \
    if (!attr_set) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)pw_stream_kernel<KB, NB, P_, E_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_set = true;                                                                                            \
    }                                                                                                             \
    hipLaunchKernelGGL((pw_stream_kernel<KB, NB, P_, E_>), grid, dim3(64 * SW), lds, st, g);                      \
    return 1;                                                                                                     \
  }
  KD_SCASE(0, 0) KD_SCASE(0, 1) KD_SCASE(1, 0) KD_SCASE(1, 1) KD_SCASE(0, 5) KD_SCASE(1, 5)
  if constexpr (KB == 2) { KD_SCASE(3, 0) KD_SCASE(3, 1) }
#undef KD_SCASE
  return 0;
}

int kd_gemm_stream_launch(GemmArgs& g, int pro, int epi, hipStream_t st) {
  StreamCfg c;
  if (!stream_enabled() || !stream_cfg(g.K, g.N, pro, epi, c)) return 0;
  const dim3 grid(stream_grid(g.M, c.ntiles), c.ntiles);
  int rc = 0;
#define KD_SHAPE(KB_, NB_) if (c.kb == KB_ && c.nb == NB_) rc = stream_dispatch<KB_, NB_>(g, pro, epi, grid, st);
  KD_SHAPE(1, 1) KD_SHAPE(1, 2) KD_SHAPE(1, 4)
  KD_SHAPE(2, 1) KD_SHAPE(2, 2) KD_SHAPE(2, 4)
  KD_SHAPE(4, 1) KD_SHAPE(4, 2) KD_SHAPE(4, 4)
#undef KD_SHAPE
  if (rc == 1) { const int e = kd_check_launch("kd_gemm_stream"); if (e) return -(e > 0 ? e : -e) - 1000; }   // < 0: error
  return rc;
}

extern "C" int kd_set_gemm_stream(int mode) { return g_stream_on.exchange(mode < 0 ? 0 : (mode > 2 ? 2 : mode)); }   // 0 off, 1 selective, 2 all; returns the previous mode
#ifdef KD_STREAM_DBG
extern "C" int kd_stream_dbg_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(kd_stream_dbg), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(kd_stream_dbg), z, sizeof(z)); }
  return 0;
}
#endif
