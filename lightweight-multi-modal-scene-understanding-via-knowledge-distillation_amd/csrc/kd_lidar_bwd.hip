// kd_lidar_bwd.hip -- ONE backward kernel per LiDAR point-MLP layer (lidar_encoder.py:29-34: Conv1d k=1 + BN1d + ReLU):
// the data gradient AND the weight gradient from one read and one bf16x3 split of the layer's two big tensors.
//
// Round 2 ran two GEMM launches per layer, kd_lidar_l2_dgrad (7.6 ms) + kd_lidar_l2_wgrad (5.7 ms) for the last layer at
// 256 frames x 80 000 points: both stream the same [20.48 M, 128] tensors Y2 (raw layer output; the scatter-max gradient
// is rebuilt from per-cell tables on load) and Y1 (raw layer input) from HBM and both convert them to bf16 planes --
// 52 GB and two conversions where 31 GB and one suffice.  Here a workgroup walks 32-row chunks of the point list and
//
//   dy[32,128]  = al*G + be*Y2 + ga                 G rebuilt from (rows, grid, share), BatchNorm-2 backward folded in
//   a1[32,128]  = act1(Y1*sc1 + sh1)
//   G1[32,128]  = (dy . W2) * act1'(Y1*sc1+sh1)     data gradient + BatchNorm-1 backward sums     (was kd_lidar_l2_dgrad)
//   dW2[128,128] += dy^T . a1                       weight gradient                               (was kd_lidar_l2_wgrad)
//
// with 8 waves in two ROLES, one wave of each per SIMD, so one role's VALU work runs beside the other's MFMAs:
//   waves 0-3 "dy + wgrad": transform the NEXT chunk's Y2 (+ table rows) into dy, cut it into three bf16 planes and store
//     them row-major into the other half of a double-buffered LDS image; then accumulate their 64x64 quadrant of dW2 over
//     the CURRENT chunk with operand fragments fetched by ds_read_b64_tr_b16 (the reduction index of this GEMM is the row);
//   waves 4-7 "a1 + dgrad + epilogue": the same for Y1 -> a1 planes; then one 32x32 block of G1 each: B operand = their 32
//     rows of W2^T as bf16 planes held in REGISTERS for the whole launch (96 VGPRs), A operand = the dy planes of the
//     current chunk (ds_read_b128, one k-step ahead); then the epilogue of the streaming kernels (mask, BatchNorm-backward
//     sums per lane across chunks, dword stores of 128-byte row segments).
// Each role keeps TWO register sets of its HBM stream (Y2 / Y1 of chunks it + 2 and it + 3 are in flight while chunk it is
// multiplied: 64 KB per CU, what one workgroup per CU needs to cover the loaded HBM latency); the cache-resident table rows
// are fetched one chunk ahead.  Loads are issued oldest-needed first, so the in-order vmcnt never waits for the youngest.
// One LDS barrier per chunk.  LDS image: [plane][row][128 bf16], 256-byte rows without padding, 16-byte chunks XOR-swizzled
// by ((row & 3) << 2) | ((row >> 2) & 3): conflict-free for the ds_write_b64 of the converter (a 16-lane group covers half a
// row), the ds_read_b128 of the A fragments (its lane groups hold 16 rows with distinct row % 16) and the transposing reads
// (a 32-lane group reads 4 rows x 64 bytes: four different bank quarters).
//
// Arithmetic: the same pieces, the same six products in the same order per 16-wide k-step as pw_stream_kernel / pw_gemm_kernel
// (SPLIT), so G1 has the bits of kd_lidar_l2_dgrad; dW2 is summed over chunks in a different (fixed) order than
// kd_lidar_l2_wgrad.  Split arithmetic only: in the exact-fp32 mode the two separate kernels run.
#include "kd_gemm_args.h"

#include <atomic>
#include <type_traits>

int kd_gemm_split_mode();     // kd_gemm.hip: 1 = bf16x6 split products (default), 0 = exact-fp32 MFMA

// Dev build only (-DKD_LB_DBG): per-phase s_memtime totals over all waves of a role, read back through kd_lb_dbg_read
// (tools/bench_lidar_bwd.py): [0..3] role B: dy conversion + LDS stores | load issue | wgrad k-loop | barrier wait;
// [4..7] role A: dgrad k-loop | epilogue | a1 conversion + load issue | barrier wait; [8] iterations (role A waves)
#ifdef KD_LB_DBG
__device__ unsigned long long kd_lb_dbg[16];
#define KD_LSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); dbg_acc[i] += t_ - dbg_t; dbg_t = t_; } while (0)
#else
#define KD_LSTAMP(i) do {} while (0)
#endif

// Timing-only probes of dev builds (-DKD_LB_PROBE=bits; results are WRONG by construction): 1 no G1 stores, 2 no dgrad MFMAs,
// 4 no wgrad MFMAs, 8 no dy transform (planes of the raw Y2), 16 no epilogue arithmetic, 32 no table gathers
#ifndef KD_LB_PROBE
#define KD_LB_PROBE 0
#endif

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int LBW = 128;                 // channels of both operand tensors of the last layer (N2 = K1 = 128)
constexpr int LBCH = 32;                 // rows per chunk
constexpr int LBPL = LBCH * LBW;         // bf16 per plane
constexpr int LBBUF = 6 * LBPL;          // bf16 per buffer: 3 dy planes + 3 a planes
constexpr int LBXF = LBCH * LBW;         // floats of the raw Y1 chunk kept beside the planes (epilogue: mask and xhat)
constexpr size_t LB_LDS = (size_t)2 * LBBUF * 2 + (size_t)2 * LBXF * 4;     // bytes, double-buffered: 96 KB of planes + 32 KB of raw Y1

struct LbArgs {
  const float* Y2; int64_t ldy2;                          // raw output of this layer [M,128]
  const int* trows; const float* tmx; const float* tshare;    // scatter-max gradient as per-cell tables (PRO4 of the GEMM kernels)
  const float* al; const float* be; const float* ga;      // BatchNorm-2 backward coefficients (kd_bn_bwd_finalize)
  const float* sc2; const float* sh2; int act2;
  const float* Y1; int64_t ldy1;                          // raw input of this layer [M,128]
  const float* sc1; const float* sh1; const float* mean1; const float* inv1; int act1;
  const float* Wt;                                        // [K1][N2]: W2 transposed (kd_transpose), the dgrad operand
  float* G1; int64_t ldg1;                                // out [M,128]
  float* partial;                                         // out [gridDim.x][2][128]: (sum G1, sum G1*xhat1) per workgroup
  float* wslab;                                           // out [gridDim.x][128][128]: partial dW2 per workgroup
  float* dump;                                            // [128] floats of workspace that absorb the stores of rows beyond M
  int M; int nt_store;
};

__device__ __forceinline__ int lb_key(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// element offset of the 16-byte chunk c (8 bf16) of row `row` inside one [LBCH][128] plane
__device__ __forceinline__ int lb_off(int row, int c) { return row * LBW + ((c ^ lb_key(row)) << 3); }

// fragment of the 32(col) x 16(row) operand block at (row0, col0) of a plane: lane (r = lane & 31, h = lane >> 5) gets rows
// row0 + 8h .. +7 of column col0 + r (same contract as kd_tr_frag in kd_gemm.hip, on the swizzled image)
__device__ __forceinline__ bf16x8 lb_tr_frag(const unsigned short* plane, int row0, int col0, int lane) {
  const int grp = lane >> 4, li = lane & 15;
  const int row = row0 + 8 * (grp >> 1) + (li >> 2);
  const int col = col0 + 16 * (grp & 1) + 4 * (li & 3);
  const unsigned short* p0 = plane + lb_off(row, col >> 3) + (col & 7);
  const unsigned short* p1 = plane + lb_off(row + 4, col >> 3) + (col & 7);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// NT: G1 is large (>= 64 MB) and next read gigabytes later: store it with the non-temporal hint (kd_nt_store).
// All three streamed tensors are dense ([M,128], row stride 128: checked by the host), M < 2^31.
template <bool NT>
__global__ __launch_bounds__(512, 1) void lidar_l2_bwd_kernel(LbArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem_raw);
  float* ldx = reinterpret_cast<float*>(smem_raw + (size_t)2 * LBBUF * 2);      // [2][LBCH][128] raw Y1 of the chunk in each buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = g.M;
  const int nchunk = (M + LBCH - 1) / LBCH;
  const int G = gridDim.x, b = blockIdx.x;
  const int nit = b < nchunk ? (nchunk - b + G - 1) / G : 0;             // this workgroup's chunks: b, b + G, b + 2G, ...
  // chunk of iteration `it`; beyond the end the last one again (its loads are harmless, its planes are never consumed)
  auto chunk_at = [&](int it) { return b + __builtin_amdgcn_readfirstlane(min(it, nit - 1)) * G; };
  // last valid row of iteration `it`'s chunk (>= 31 except in the tail chunk of the whole problem); -1 for the padding
  // iteration that makes the trip count even (the loop body is two steps, one per register set, WITHOUT inner branches:
  // any control flow between memory operations makes the waitcnt pass merge its in-order counts conservatively and drain
  // the queue -- the stores of G1, the two-deep prefetch -- once per chunk)
  auto last_at = [&](int it) {
    const int vm = -(int)(it < nit);                                       // all ones for a real iteration (branch-free on purpose)
    return ((M - 1 - chunk_at(it) * LBCH) & vm) | ~vm;
  };
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest terms first (as pw_gemm_kernel)
  const int tw = tid & 255;                                               // thread index inside the role
  const int c4 = tw & 31, rb = tw >> 5;                                   // converter layout: float4 column group (fixed), first row
  // per-lane element offsets of the converter's four rows rb + 8 i inside a plane (fixed for the launch)
  int cvo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cvo[i] = lb_off(rb + 8 * i, c4 >> 1) + (c4 & 1) * 4;
  auto split_store = [&](float4 v, unsigned short* d) {
    uint2 hi, mid, lo;
    kd_split3(v, hi, mid, lo);
    *reinterpret_cast<uint2*>(d) = hi;
    *reinterpret_cast<uint2*>(d + LBPL) = mid;
    *reinterpret_cast<uint2*>(d + 2 * LBPL) = lo;
  };
  // One chunk of a dense [M,128] tensor in the converter layout: (wave-uniform chunk base) + (per-lane offset); the row is
  // clamped to the last valid one of the chunk (`last` = M - 1 - m0 >= 31 except in the tail chunk of the whole problem)
  auto load_rows4 = [&](const float* T, int chunk, float4 (&dst)[4]) {
    const float* base = T + (size_t)chunk * (LBCH * LBW);
    const int last = M - 1 - chunk * LBCH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rb + 8 * i;
      dst[i] = kd_ld4(base + (row < last ? row : last) * LBW + 4 * c4);
    }
  };

  if (wave < 4) {
    // =============================== role B: dy planes of the next chunk, weight gradient of the current one ===========
    const int wn = wave >> 1, wk = wave & 1;                              // 64x64 quadrant of dW2
    const float4 cal = kd_ld4(g.al + 4 * c4), cbe = kd_ld4(g.be + 4 * c4), cga = kd_ld4(g.ga + 4 * c4);
    const float4 cms = kd_ld4(g.sc2 + 4 * c4), cmh = kd_ld4(g.sh2 + 4 * c4);
    float4 ry[2][4], rx[4], rs[4];
    int trq[2][4];                                                        // table rows, fetched a full step before the table loads need them
    int tvb[2];                                                           // bit i: row i of the chunk lies in a cell (set when its tables are issued)
    auto fetch_rows = [&](int chunk, int (&tr)[4]) {
      const int last = M - 1 - chunk * LBCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb + 8 * i;
        tr[i] = g.trows[chunk * LBCH + (row < last ? row : last)];
      }
    };
    auto load_tables = [&](const int (&trw)[4], int& bits) {              // the table rows of one chunk
      bits = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int t = trw[i];
        if (KD_LB_PROBE & 32) t = i;
        asm volatile("" : "+v"(t));     // first use of the fetched row HERE: instruction selection otherwise floats the sign tests up
                                        // to the load itself (a full step earlier) and the wait for the fetch with them
        const size_t o = (size_t)(t < 0 ? 0 : t) * LBW + 4 * c4;
        rx[i] = kd_ld4(g.tmx + o);
        rs[i] = kd_ld4(g.tshare + o);
        bits |= (t >= 0 ? 1 : 0) << i;
      }
    };
    auto convert_store = [&](int last, const float4 (&src)[4], int bits, unsigned short* buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 x = src[i], mx = rx[i], sv = rs[i];
        const float4 a = kd_affine_act4(x, cms, cmh, g.act2);
        const bool tv = (bits >> i) & 1;
        // dy = al * G + be * Y2 + ga with G = (row in a cell, value positive and the cell maximum) ? share : 0 -- the operand of
        // kd_bwd_operand(G, Y2, al, be, ga, ., ., NONE), whose mask is identically one
        float4 v;
        if (KD_LB_PROBE & 8) { split_store(x, buf + cvo[i]); continue; }
        v.x = fmaf(cal.x, (tv && a.x > 0.f && a.x == mx.x) ? sv.x : 0.f, fmaf(cbe.x, x.x, cga.x));
        v.y = fmaf(cal.y, (tv && a.y > 0.f && a.y == mx.y) ? sv.y : 0.f, fmaf(cbe.y, x.y, cga.y));
        v.z = fmaf(cal.z, (tv && a.z > 0.f && a.z == mx.z) ? sv.z : 0.f, fmaf(cbe.z, x.z, cga.z));
        v.w = fmaf(cal.w, (tv && a.w > 0.f && a.w == mx.w) ? sv.w : 0.f, fmaf(cbe.w, x.w, cga.w));
        const bool ok = rb + 8 * i <= last;                               // rows beyond M contribute nothing to dW2 (a1 may hold anything there)
        v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        split_store(v, buf + cvo[i]);
      }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    // One iteration: planes of chunk it + 1 (Y2 in register set S, tables in rx / rs) -> the other LDS buffer; then, oldest
    // first for the in-order vmcnt: tables of chunk it + 2, rows of chunk it + 3, Y2 of chunk it + 3 (into the set just
    // consumed: two chunks of the HBM stream stay in flight); then the weight-gradient MFMAs of chunk it.
#ifdef KD_LB_DBG
    unsigned long long dbg_acc[4] = {0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#endif
    auto step = [&](int it, auto set_tag) {
      constexpr int S = decltype(set_tag)::value;                         // register set holding Y2 of chunk it + 1
      // Issue order = the order the results are needed in (the in-order vmcnt can then leave every younger load in flight);
      // sched_barriers keep hipcc's scheduler from sinking the table loads below the Y2 prefetch.
      fetch_rows(chunk_at(it + 3), trq[S]);                              // rows of chunk it + 3: their tables are issued by the NEXT step
      __builtin_amdgcn_sched_barrier(0);
      convert_store(last_at(it + 1), ry[S], tvb[S], lds + ((it + 1) & 1) * LBBUF);
      KD_LSTAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      load_tables(trq[S ^ 1], tvb[S ^ 1]);                               // tables of chunk it + 2 (rows fetched one step ago)
      __builtin_amdgcn_sched_barrier(0);
      load_rows4(g.Y2, chunk_at(it + 3), ry[S]);
      __builtin_amdgcn_sched_barrier(0);
      KD_LSTAMP(1);
      const unsigned short* buf = lds + (it & 1) * LBBUF;
#pragma unroll
      for (int ks = 0; ks < LBCH / 16; ++ks) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          bf16x8 d[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) d[p] = lb_tr_frag(buf + p * LBPL, 16 * ks, 64 * wn + 32 * ni, lane);
#pragma unroll
          for (int ki = 0; ki < 2; ++ki) {          // (a fragments re-read per ni: 12 registers instead of 24, LDS has the room)
            bf16x8 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = lb_tr_frag(buf + (3 + p) * LBPL, 16 * ks, 64 * wk + 32 * ki, lane);
#pragma unroll
            for (int t = 0; t < ((KD_LB_PROBE & 4) ? 1 : 6); ++t)
              acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[PA[t]], a[PB[t]], acc[ni][ki], 0, 0, 0);
          }
        }
      }
      KD_LSTAMP(2);
      kd_lds_barrier();
      KD_LSTAMP(3);
    };

    if (nit > 0) {
      // state at the first step (it = 0, S = 1): ry[1] = Y2 of chunk 1, rx / rs / tvb[1] = tables of chunk 1, trq[0] = rows of
      // chunk 2, ry[0] = Y2 of chunk 2; rows of chunk c live in trq[c & 1]
      fetch_rows(chunk_at(0), trq[0]);
      load_tables(trq[0], tvb[0]);
      load_rows4(g.Y2, chunk_at(0), ry[0]);
      fetch_rows(chunk_at(1), trq[1]);
      load_rows4(g.Y2, chunk_at(1), ry[1]);
      convert_store(last_at(0), ry[0], tvb[0], lds);
      load_tables(trq[1], tvb[1]);
      fetch_rows(chunk_at(2), trq[0]);
      load_rows4(g.Y2, chunk_at(2), ry[0]);
    }
    // Enter the loop with an EMPTY memory queue: hipcc may reorder the prologue's independent loads, and whatever is pending
    // on the entry path is merged into the loop's in-order counts conservatively -- a register set that happens to be loaded
    // last here would be waited for with vmcnt(3..0) on every trip, draining the two-deep prefetch.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    kd_lds_barrier();
    for (int it = 0; it < nit; it += 2) {
      step(it, std::integral_constant<int, 1>{});
      step(it + 1, std::integral_constant<int, 0>{});
    }
#ifdef KD_LB_DBG
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&kd_lb_dbg[i], dbg_acc[i]);
#endif
    float* out = g.wslab + (size_t)b * (LBW * LBW);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        const int col = 64 * wk + 32 * ki + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 64 * wn + 32 * ni + (q & 3) + 8 * (q >> 2) + 4 * h;
          out[row * LBW + col] = acc[ni][ki][q];
        }
      }
  } else {
    // =============================== role A: a1 planes of the next chunk, data gradient of the current one + epilogue ====
    const int j = wave - 4;                                              // 32-column block of G1
    const int col = 32 * j + r;
    const float4 cas = kd_ld4(g.sc1 + 4 * c4), cah = kd_ld4(g.sh1 + 4 * c4);
    float4 ra[2][4];
    // a1 planes + the raw rows themselves (the epilogue of the chunk reads them back in the accumulator layout)
    auto convert_store = [&](const float4 (&src)[4], unsigned short* buf, float* xbuf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kd_st4(xbuf + (rb + 8 * i) * LBW + 4 * c4, src[i]);
        split_store(kd_affine_act4(src[i], cas, cah, g.act1), buf + 3 * LBPL + cvo[i]);
      }
    };
    if (nit > 0) {
      load_rows4(g.Y1, chunk_at(0), ra[0]);
      load_rows4(g.Y1, chunk_at(1), ra[1]);
    }
    bf16x8 Wb[8][3];                                                     // W2^T rows 32j + r, all 128 k, three planes: 96 VGPRs
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* wp = g.Wt + col * LBW + 16 * u + 8 * h;
      uint2 h0, m0, l0, h1, m1, l1;
      kd_split3(kd_ld4(wp), h0, m0, l0);
      kd_split3(kd_ld4(wp + 4), h1, m1, l1);
      const u32x4 vh = {h0.x, h0.y, h1.x, h1.y}, vm = {m0.x, m0.y, m1.x, m1.y}, vl = {l0.x, l0.y, l1.x, l1.y};
      Wb[u][0] = __builtin_bit_cast(bf16x8, vh);
      Wb[u][1] = __builtin_bit_cast(bf16x8, vm);
      Wb[u][2] = __builtin_bit_cast(bf16x8, vl);
    }
    const float esc = g.sc1[col], esh = g.sh1[col], emean = g.mean1[col], einv = g.inv1[col];
    float s1 = 0.f, s2 = 0.f;
    float zero = 0.f;
    asm volatile("" : "+v"(zero));      // (+0.0 added like the bias-free GEMM epilogue does: keeps the sign of zero results identical)
    // First use of these loaded constants BEFORE the loop: a value whose first use sits inside the loop is "pending" on the
    // loop-entry path for the waitcnt pass, which then waits for it (vmcnt of a handful: everything but the youngest loads,
    // i.e. all sixteen stores of the previous chunk) on EVERY trip.
    asm volatile("" :: "v"(esc), "v"(esh), "v"(emean), "v"(einv));
    const int o_lane = 4 * h * LBW + col;                                // accumulator layout: register q is row (q & 3) + 8 (q >> 2) + 4 h
    const int a_row = r * LBW, a_kk = (h ^ lb_key(r)) << 3;              // A fragment of k-step u: row r, chunk (2u + h) ^ key(r)
    auto load_a = [&](const unsigned short* buf, int u, bf16x8 (&ap)[3]) {
      const unsigned short* p = buf + a_row + (a_kk ^ (16 * u));
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) ap[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + pl * LBPL));
    };

#ifdef KD_LB_DBG
    unsigned long long dbg_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#endif
    auto step = [&](int it, auto set_tag) {
      constexpr int S = decltype(set_tag)::value;                         // register set holding Y1 of chunk it + 1
      const int m0 = chunk_at(it) * LBCH;
      const int last = last_at(it);
      // (1) data gradient of chunk it FIRST (the partner wave of this SIMD starts its iteration with VALU work -- the dy
      // conversion -- and ends it with MFMAs: the two roles run out of phase); A fragments one k-step ahead of their MFMAs
      const unsigned short* buf = lds + (it & 1) * LBBUF;
      f32x16 dacc;
#pragma unroll
      for (int q = 0; q < 16; ++q) dacc[q] = 0.f;
      bf16x8 ap[2][3];
      load_a(buf, 0, ap[0]);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u < 7) load_a(buf, u + 1, ap[(u + 1) & 1]);
#pragma unroll
        for (int t = 0; t < ((KD_LB_PROBE & 2) ? 1 : 6); ++t)
          dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[u & 1][PA[t]], Wb[u][PB[t]], dacc, 0, 0, 0);
      }
      KD_LSTAMP(4);
      // (2) epilogue: the raw Y1 values (the tensor whose activation is differentiated) come back from LDS in the accumulator
      // layout -- 32 consecutive floats per half wave: conflict-free
      float xr[16];
      const float* xb = ldx + (it & 1) * LBXF + o_lane;
#pragma unroll
      for (int q = 0; q < 16; ++q) xr[q] = xb[((q & 3) + 8 * (q >> 2)) * LBW];
      float* cbase = g.G1 + (size_t)m0 * LBW;
      // sixteen UNCONDITIONAL stores: rows beyond M (tail chunk, padding iteration) go to a dump line of the workspace
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
        const bool ok = row <= last;
        const float x = xr[q];
        float v = dacc[q] + zero;
        if (!(KD_LB_PROBE & 16)) {
          v *= kd_act_mask(kd_affine(x, esc, esh), g.act1);
          const float vs = ok ? v : 0.f;
          s1 += vs; s2 = fmaf(vs, (x - emean) * einv, s2);
        } else s1 += v + x;
        float* dst = ok ? cbase + ((q & 3) + 8 * (q >> 2)) * LBW + o_lane : g.dump + col;
        if (KD_LB_PROBE & 1) dst = g.dump + col;
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
      }
      KD_LSTAMP(5);
      // (3) a1 planes of chunk it + 1 -> the other buffer, then Y1 of chunk it + 3 into the set just consumed
      convert_store(ra[S], lds + ((it + 1) & 1) * LBBUF, ldx + ((it + 1) & 1) * LBXF);
      load_rows4(g.Y1, chunk_at(it + 3), ra[S]);
      KD_LSTAMP(6);
      kd_lds_barrier();
      KD_LSTAMP(7);
#ifdef KD_LB_DBG
      dbg_acc[8] += 1;
#endif
    };

    if (nit > 0) {
      convert_store(ra[0], lds, ldx);
      load_rows4(g.Y1, chunk_at(2), ra[0]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): see role B
    kd_lds_barrier();
    for (int it = 0; it < nit; it += 2) {
      step(it, std::integral_constant<int, 1>{});
      step(it + 1, std::integral_constant<int, 0>{});
    }
#ifdef KD_LB_DBG
    if (lane == 0) for (int i = 4; i < 9; ++i) atomicAdd(&kd_lb_dbg[i], dbg_acc[i]);
#endif
    const float t1 = s1 + __shfl_xor(s1, 32, 64), t2 = s2 + __shfl_xor(s2, 32, 64);
    if (h == 0) {
      g.partial[(b * 2 + 0) * LBW + col] = t1;
      g.partial[(b * 2 + 1) * LBW + col] = t2;
    }
  }
}

int lb_grid(int64_t M) {
  const int64_t nchunk = (M + LBCH - 1) / LBCH;
  return (int)(nchunk < 256 ? nchunk : 256);       // one workgroup per CU (96 KB of LDS, 8 waves)
}

}  // namespace

extern "C" {

// 1 when kd_lidar_l2_bwd has an instance for (N2, K1) in the current GEMM arithmetic
int kd_lidar_l2_bwd_supported(int N2, int K1) { return kd_gemm_split_mode() && N2 == LBW && K1 == LBW; }
// rows of the BatchNorm-backward slab kd_lidar_l2_bwd writes ([rows][2][K1]) and bytes of its weight-gradient workspace
int64_t kd_lidar_l2_bwd_stat_rows(int64_t M) { return lb_grid(M); }
size_t kd_lidar_l2_bwd_ws_bytes(int64_t M, int N2, int K1) { return ((size_t)lb_grid(M) * N2 * K1 + LBW) * sizeof(float); }

// Training backward of the last point-MLP layer in ONE kernel (see the head of this file): G1 and its BatchNorm-1 backward
// sums as kd_lidar_l2_dgrad, dW2 as kd_lidar_l2_wgrad.  partial_rows must equal kd_lidar_l2_bwd_stat_rows(M).
int kd_lidar_l2_bwd(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share, const float* al,
                    const float* be, const float* ga, const float* sc2, const float* sh2, int act2, const float* Wt,
                    float* G1, int64_t ldg1, const float* Y1, int64_t ldy1, const float* sc1, const float* sh1,
                    const float* mean1, const float* invstd1, int act1, float* partial, int64_t partial_rows, float* dW,
                    int64_t M, int N2, int K1, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(Y2 && rows && grid && share && al && be && ga && sc2 && sh2 && Wt && G1 && Y1 && sc1 && sh1 && mean1 && invstd1 && partial && dW &&
             ws && M > 0, KD_ERR_ARG, "kd_lidar_l2_bwd: bad args");
  KD_REQUIRE(kd_lidar_l2_bwd_supported(N2, K1), KD_ERR_SHAPE,
             "kd_lidar_l2_bwd: no instance for N2=%d K1=%d in the %s arithmetic (use kd_lidar_l2_dgrad + kd_lidar_l2_wgrad)", N2, K1,
             kd_gemm_split_mode() ? "split" : "exact-fp32");
  KD_REQUIRE(M < (int64_t)1 << 31 && ldy2 == N2 && ldg1 == K1 && ldy1 == K1, KD_ERR_SHAPE,
             "kd_lidar_l2_bwd: Y2, Y1 and G1 must be dense [M,128] matrices (row strides %lld, %lld, %lld)", (long long)ldy2, (long long)ldy1,
             (long long)ldg1);
  KD_REQUIRE(kd_aligned16(Y2) && kd_aligned16(grid) && kd_aligned16(share) && kd_aligned16(Wt) && kd_aligned16(G1) && kd_aligned16(Y1) &&
             kd_aligned16(al) && kd_aligned16(be) && kd_aligned16(ga) && kd_aligned16(sc2) && kd_aligned16(sh2) && kd_aligned16(sc1) &&
             kd_aligned16(sh1) && kd_aligned16(ws), KD_ERR_ALIGN, "kd_lidar_l2_bwd: 16-byte alignment");
  KD_REQUIRE(act2 == KD_ACT_RELU || act2 == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_l2_bwd: the scatter-max tables need a non-negative activation");
  const int grid_x = lb_grid(M);
  KD_REQUIRE(partial_rows == grid_x, KD_ERR_ARG, "kd_lidar_l2_bwd: statistics slab sized for %lld rows, this launch writes %d "
             "(kd_lidar_l2_bwd_stat_rows)", (long long)partial_rows, grid_x);
  KD_REQUIRE(ws_bytes >= kd_lidar_l2_bwd_ws_bytes(M, N2, K1), KD_ERR_WORKSPACE, "kd_lidar_l2_bwd: workspace too small (%zu B)", ws_bytes);
  static std::atomic<uint64_t> lds_raised[2];
  const int nt = kd_nt_store((size_t)M * K1 * sizeof(float));
  const void* fn = nt ? (const void*)lidar_l2_bwd_kernel<true> : (const void*)lidar_l2_bwd_kernel<false>;
  const hipError_t e = kd_raise_dynamic_lds(fn, LB_LDS, lds_raised[nt]);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_l2_bwd: cannot raise the dynamic LDS limit to %zu B: %s", LB_LDS, hipGetErrorString(e));
  LbArgs g{Y2, ldy2, rows, grid, share, al, be, ga, sc2, sh2, act2, Y1, ldy1, sc1, sh1, mean1, invstd1, act1, Wt, G1, ldg1, partial,
           (float*)ws, (float*)ws + (size_t)grid_x * N2 * K1, (int)M, nt};
  hipStream_t st = (hipStream_t)stream;
  if (nt) hipLaunchKernelGGL(lidar_l2_bwd_kernel<true>, dim3(grid_x), dim3(512), LB_LDS, st, g);
  else hipLaunchKernelGGL(lidar_l2_bwd_kernel<false>, dim3(grid_x), dim3(512), LB_LDS, st, g);
  const int rc = kd_check_launch("kd_lidar_l2_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, grid_x, (int64_t)N2 * K1, dW, st);
}

#ifdef KD_LB_DBG
int kd_lb_dbg_read(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(kd_lb_dbg), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(kd_lb_dbg), z, sizeof(z)); }
  return 0;
}
#endif

}  // extern "C"
