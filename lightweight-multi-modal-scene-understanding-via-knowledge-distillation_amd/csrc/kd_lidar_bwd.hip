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

// Round 4, measured and not kept: the matrix wave that holds a 32 x 32 block of (dy . W2) finishing G1 itself (mask from the vector
// waves' LDS copy of Y1, BatchNorm sums, sixteen dword stores from the accumulators; no stage tile, no vector-wave epilogue): the L2
// kernel went from 8.69-8.75 to 9.37-9.47 ms in the step on the same box (tools/r4_ab_lb.sh) -- the sixteen LDS reads, selects and
// stores sit in the matrix waves' issue stream between two chunks' MFMAs, and those waves are not as idle as the vector waves are busy.
//
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
constexpr size_t LB_LDS = (size_t)2 * LBBUF * 2 + (size_t)4 * LBXF * 4;     // bytes: 96 KB of planes + 32 KB of raw Y1 + 32 KB of stage tiles = all 160 KB

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

// Three-plane split of N float4 values in LOCKSTEP: every step of the cvt / subtract chain is issued for all 2N pairs before the
// next one, so consecutive instructions are independent -- a single wave issues dependent VALU instructions at ~6.6 cycles
// each instead of 4 (MI355X_MICROARCH.md), and the row-after-row form of this code was one long dependent chain.
// Same values as kd_split3 (kd_gemm_args.h) on each element.
template <int N>
__device__ __forceinline__ void lb_split3_lockstep(const float4 (&x)[N], uint2 (&hi)[N], uint2 (&mid)[N], uint2 (&lo)[N]) {
  f32x2 v[2 * N];
  uint32_t p[2 * N];
#pragma unroll
  for (int i = 0; i < N; ++i) { v[2 * i] = f32x2{x[i].x, x[i].y}; v[2 * i + 1] = f32x2{x[i].z, x[i].w}; }
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) p[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[i], bf16x2));
#pragma unroll
  for (int i = 0; i < N; ++i) hi[i] = make_uint2(p[2 * i], p[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) { v[i][0] -= __uint_as_float(p[i] << 16); v[i][1] -= __uint_as_float(p[i] & 0xffff0000u); }
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) p[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[i], bf16x2));
#pragma unroll
  for (int i = 0; i < N; ++i) mid[i] = make_uint2(p[2 * i], p[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) { v[i][0] -= __uint_as_float(p[i] << 16); v[i][1] -= __uint_as_float(p[i] & 0xffff0000u); }
#pragma unroll
  for (int i = 0; i < 2 * N; ++i) p[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[i], bf16x2));
#pragma unroll
  for (int i = 0; i < N; ++i) lo[i] = make_uint2(p[2 * i], p[2 * i + 1]);
}

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
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem_raw);                 // [2][6 planes][32][128] bf16
  float* ldx = reinterpret_cast<float*>(smem_raw + (size_t)2 * LBBUF * 2);           // [2][32][128] raw Y1 of the chunk in each buffer
  float* stage = ldx + 2 * LBXF;                                                     // [2][32][128] (dy . W2) of a chunk, matrix waves -> vector waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = g.M;
  const int nchunk = (M + LBCH - 1) / LBCH;
  const int G = gridDim.x, b = blockIdx.x;
  const int nit = b < nchunk ? (nchunk - b + G - 1) / G : 0;             // this workgroup's chunks: b, b + G, b + 2G, ...
  // chunk of iteration `it`; beyond the end the last one again (its loads are harmless, its planes are never consumed)
  auto chunk_at = [&](int it) __attribute__((always_inline)) { return b + __builtin_amdgcn_readfirstlane(min(max(it, 0), nit - 1)) * G; };
  // last valid row of iteration `it`'s chunk (>= 31 except in the tail chunk of the whole problem); -1 for the padding
  // iteration that makes the trip count even and for it < 0 (the loop body is two steps, one per register set, WITHOUT
  // inner branches: any control flow between memory operations makes the waitcnt pass merge its in-order counts
  // conservatively and drain the queue -- the stores of G1, the two-deep prefetch -- once per chunk)
  auto last_at = [&](int it) __attribute__((always_inline)) {
    const int vm = -(int)(it >= 0 && it < nit);                            // all ones for a real iteration (branch-free on purpose)
    return ((M - 1 - chunk_at(it) * LBCH) & vm) | ~vm;
  };
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest terms first (as pw_gemm_kernel)

  if (wave < 4) {
    // ======== vector waves: all global memory traffic and all VALU work -- operand planes of the NEXT chunk, epilogue of the PREVIOUS
    const int c4 = tid & 31, rb = tid >> 5;                               // float4 column group (fixed), first row; rows rb + 8 i
    int cvo[4];                                                           // element offsets of those rows inside a plane
#pragma unroll
    for (int i = 0; i < 4; ++i) cvo[i] = lb_off(rb + 8 * i, c4 >> 1) + (c4 & 1) * 4;
    const int xo = rb * LBW + 4 * c4;                                     // float offset of row rb in the fp32 tiles (ldx, stage, G1)
    const float4 cal = kd_ld4(g.al + 4 * c4), cbe = kd_ld4(g.be + 4 * c4), cga = kd_ld4(g.ga + 4 * c4);
    const float4 cms = kd_ld4(g.sc2 + 4 * c4), cmh = kd_ld4(g.sh2 + 4 * c4);
    const float4 cas = kd_ld4(g.sc1 + 4 * c4), cah = kd_ld4(g.sh1 + 4 * c4);
    const float4 cmean = kd_ld4(g.mean1 + 4 * c4), cinv = kd_ld4(g.inv1 + 4 * c4);
    float4 ry[2][4], ra[2][4], rx[4], rs[4];
    int trq[2][4];                                                        // table rows of chunk c in trq[c & 1], fetched a full step ahead
    int tvb[2];                                                           // bit i: row i of the chunk lies in a cell (set when its tables are issued)
    float4 s1 = kd_zero4(), s2 = kd_zero4();                              // BatchNorm-1 backward sums of this thread's 4 columns
    auto split_store = [&](float4 v, unsigned short* d) __attribute__((always_inline)) {
      uint2 hi, mid, lo;
      kd_split3(v, hi, mid, lo);
      *reinterpret_cast<uint2*>(d) = hi;
      *reinterpret_cast<uint2*>(d + LBPL) = mid;
      *reinterpret_cast<uint2*>(d + 2 * LBPL) = lo;
    };
    // one chunk of a dense [M,128] tensor: (wave-uniform chunk base) + (per-lane offset); rows clamped to the chunk's last valid one
    auto load_rows4 = [&](const float* T, int chunk, float4 (&dst)[4]) __attribute__((always_inline)) {
      const float* base = T + (size_t)chunk * (LBCH * LBW);
      const int last = M - 1 - chunk * LBCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb + 8 * i;
        dst[i] = kd_ld4(base + (row < last ? row : last) * LBW + 4 * c4);
      }
    };
    auto fetch_rows = [&](int chunk, int (&tr)[4]) __attribute__((always_inline)) {
      const int last = M - 1 - chunk * LBCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb + 8 * i;
        tr[i] = g.trows[chunk * LBCH + (row < last ? row : last)];
      }
    };
    auto load_tables = [&](const int (&trw)[4], int& bits) __attribute__((always_inline)) {              // the table rows of one chunk
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
    // dy and a1 planes (+ the raw Y1 rows for the epilogue two steps later) of one chunk.  Both activations are ReLU (checked by
    // the host).  VALU only: a compare that lands in an SGPR pair, is combined by s_and and comes back as a v_cndmask mask
    // costs the in-order wave a VALU -> SALU -> VALU round trip per element (measured: 9 cycles per instruction on this code).
    //   G = (row in a cell && a > 0 && a == cell max) ? share : 0,  a = relu(Y2*sc2 + sh2)
    //     = (a == max(cell max, denorm_min)) ? share_or_0 : 0       (a >= 0; a positive maximum is >= denorm_min; share_or_0 = 0 off-grid)
    //   dy = al * G + be * Y2 + ga      (kd_bwd_operand(G, Y2, al, be, ga, ., ., NONE), whose mask is identically one)
    auto convert_store = [&](int last, const float4 (&sy)[4], const float4 (&sa)[4], int bits, int bufi) __attribute__((always_inline)) {
      unsigned short* buf = lds + bufi * LBBUF;
      float* xbuf = ldx + bufi * LBXF;
      const float tiny = __uint_as_float(1u);
      {
        float4 vw[4];                                                     // dy of the four rows: all transformed, then all split in lockstep
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 x = sy[i];
          const bool tv = (bits >> i) & 1;
          const float4 mx = make_float4(fmaxf(rx[i].x, tiny), fmaxf(rx[i].y, tiny), fmaxf(rx[i].z, tiny), fmaxf(rx[i].w, tiny));
          const float4 sv = make_float4(tv ? rs[i].x : 0.f, tv ? rs[i].y : 0.f, tv ? rs[i].z : 0.f, tv ? rs[i].w : 0.f);
          const float4 a = make_float4(fmaxf(kd_affine(x.x, cms.x, cmh.x), 0.f), fmaxf(kd_affine(x.y, cms.y, cmh.y), 0.f),
                                       fmaxf(kd_affine(x.z, cms.z, cmh.z), 0.f), fmaxf(kd_affine(x.w, cms.w, cmh.w), 0.f));
          vw[i].x = fmaf(cal.x, a.x == mx.x ? sv.x : 0.f, fmaf(cbe.x, x.x, cga.x));
          vw[i].y = fmaf(cal.y, a.y == mx.y ? sv.y : 0.f, fmaf(cbe.y, x.y, cga.y));
          vw[i].z = fmaf(cal.z, a.z == mx.z ? sv.z : 0.f, fmaf(cbe.z, x.z, cga.z));
          vw[i].w = fmaf(cal.w, a.w == mx.w ? sv.w : 0.f, fmaf(cbe.w, x.w, cga.w));
          if (KD_LB_PROBE & 8) vw[i] = x;
        }
        uint2 hi[4], mid[4], lo[4];
        lb_split3_lockstep<4>(vw, hi, mid, lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned short* d = buf + cvo[i];
          *reinterpret_cast<uint2*>(d) = hi[i];
          *reinterpret_cast<uint2*>(d + LBPL) = mid[i];
          *reinterpret_cast<uint2*>(d + 2 * LBPL) = lo[i];
        }
      }
      {
        float4 vw[4];                                                     // a1 of the four rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          kd_st4(xbuf + xo + 8 * i * LBW, sa[i]);
          vw[i] = make_float4(fmaxf(kd_affine(sa[i].x, cas.x, cah.x), 0.f), fmaxf(kd_affine(sa[i].y, cas.y, cah.y), 0.f),
                              fmaxf(kd_affine(sa[i].z, cas.z, cah.z), 0.f), fmaxf(kd_affine(sa[i].w, cas.w, cah.w), 0.f));
        }
        uint2 hi[4], mid[4], lo[4];
        lb_split3_lockstep<4>(vw, hi, mid, lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned short* d = buf + 3 * LBPL + cvo[i];
          *reinterpret_cast<uint2*>(d) = hi[i];
          *reinterpret_cast<uint2*>(d + LBPL) = mid[i];
          *reinterpret_cast<uint2*>(d + 2 * LBPL) = lo[i];
        }
      }
      if (last < LBCH - 1) {
        // the tail chunk of the whole problem (and the padding iteration): rows beyond M must contribute nothing to dW2 -- zero
        // their dy rows.  A rare, wave-uniform branch around LDS stores only (no vector-memory operation inside: the waitcnt
        // pass keeps its exact counts).
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rb + 8 * i > last) {
            const uint2 z = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(buf + cvo[i]) = z;
            *reinterpret_cast<uint2*>(buf + LBPL + cvo[i]) = z;
            *reinterpret_cast<uint2*>(buf + 2 * LBPL + cvo[i]) = z;
          }
      }
    };
    // G1 rows of one chunk: (dy . W2) from the matrix waves' stage tile, times act1'(Y1*sc1+sh1); BatchNorm-1 backward sums;
    // four UNCONDITIONAL 16-byte stores per thread (rows beyond M and the padding iterations go to a dump line of the workspace)
    float zero = 0.f;
    asm volatile("" : "+v"(zero));      // (+0.0 added like the bias-free GEMM epilogue does: keeps the sign of zero results identical)
    auto epilogue = [&](int it) __attribute__((always_inline)) {
      const int last = last_at(it), bufi = it & 1;
      float* cbase = g.G1 + (size_t)chunk_at(it) * (LBCH * LBW) + xo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 d = kd_ld4(stage + bufi * LBXF + xo + 8 * i * LBW), x = kd_ld4(ldx + bufi * LBXF + xo + 8 * i * LBW);
        const bool ok = rb + 8 * i <= last;
        float4 v = make_float4(d.x + zero, d.y + zero, d.z + zero, d.w + zero);
        if (!(KD_LB_PROBE & 16)) {
          // v *= relu'(Y1*sc1 + sh1): a select on VCC (no SGPR round trip); rows beyond M count as zero (every tile value is finite:
          // the fp32 tiles are zero-filled at the start and only ever hold results of finite loads)
          v.x = kd_affine(x.x, cas.x, cah.x) > 0.f ? v.x : 0.f;
          v.y = kd_affine(x.y, cas.y, cah.y) > 0.f ? v.y : 0.f;
          v.z = kd_affine(x.z, cas.z, cah.z) > 0.f ? v.z : 0.f;
          v.w = kd_affine(x.w, cas.w, cah.w) > 0.f ? v.w : 0.f;
          const float4 vs = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
          s1.x += vs.x; s1.y += vs.y; s1.z += vs.z; s1.w += vs.w;
          s2.x = fmaf(vs.x, (x.x - cmean.x) * cinv.x, s2.x);
          s2.y = fmaf(vs.y, (x.y - cmean.y) * cinv.y, s2.y);
          s2.z = fmaf(vs.z, (x.z - cmean.z) * cinv.z, s2.z);
          s2.w = fmaf(vs.w, (x.w - cmean.w) * cinv.w, s2.w);
        } else { s1.x += v.x + x.x; }
        float* dst = (ok && !(KD_LB_PROBE & 1)) ? cbase + 8 * i * LBW : g.dump + 4 * c4;
        if (NT) kd_st4_nt(dst, v); else kd_st4(dst, v);
      }
    };
#ifdef KD_LB_DBG
    unsigned long long dbg_acc[5] = {0, 0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#endif
    // One step.  Issue order = the order the results are needed in (the in-order vmcnt then leaves every younger operation in
    // flight); sched_barriers keep hipcc's scheduler from sinking the table loads below the prefetch of the two streams.
    auto step = [&](int it, auto set_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;                         // register set holding Y2 / Y1 of chunk it + 1
      fetch_rows(chunk_at(it + 3), trq[S]);                              // rows of chunk it + 3: their tables are issued by the NEXT step
      __builtin_amdgcn_sched_barrier(0);
      epilogue(it - 1);                                                   // the matrix waves left (dy . W2) of chunk it - 1 one barrier ago
      KD_LSTAMP(0);
      __builtin_amdgcn_sched_barrier(0);
      convert_store(last_at(it + 1), ry[S], ra[S], tvb[S], (it + 1) & 1);
      KD_LSTAMP(1);
      __builtin_amdgcn_sched_barrier(0);
      load_tables(trq[S ^ 1], tvb[S ^ 1]);                               // tables of chunk it + 2 (rows fetched one step ago)
      __builtin_amdgcn_sched_barrier(0);
      load_rows4(g.Y2, chunk_at(it + 3), ry[S]);
      load_rows4(g.Y1, chunk_at(it + 3), ra[S]);
      __builtin_amdgcn_sched_barrier(0);
      KD_LSTAMP(2);
      kd_lds_barrier();
      KD_LSTAMP(3);
    };

    for (int i = tid; i < 4 * LBXF / 4; i += 256) kd_st4(ldx + 4 * i, kd_zero4());      // ldx and stage: finite from the first read on
    if (nit > 0) {
      // state at the first step (it = 0, S = 1): ry / ra[1] = chunk 1, rx / rs / tvb[1] = tables of chunk 1, trq[0] = rows of
      // chunk 2, ry / ra[0] = chunk 2; rows of chunk c live in trq[c & 1]
      fetch_rows(chunk_at(0), trq[0]);
      load_tables(trq[0], tvb[0]);
      load_rows4(g.Y2, chunk_at(0), ry[0]);
      load_rows4(g.Y1, chunk_at(0), ra[0]);
      fetch_rows(chunk_at(1), trq[1]);
      load_rows4(g.Y2, chunk_at(1), ry[1]);
      load_rows4(g.Y1, chunk_at(1), ra[1]);
      convert_store(last_at(0), ry[0], ra[0], tvb[0], 0);
      load_tables(trq[1], tvb[1]);
      fetch_rows(chunk_at(2), trq[0]);
      load_rows4(g.Y2, chunk_at(2), ry[0]);
      load_rows4(g.Y1, chunk_at(2), ra[0]);
    }
    // Enter the loop with an EMPTY memory queue: hipcc may reorder the prologue's independent loads, and whatever is pending
    // on the entry path is merged into the loop's in-order counts conservatively.
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0)
    kd_lds_barrier();
    int it = 0;
    for (; it < nit; it += 2) {
      step(it, std::integral_constant<int, 1>{});
      step(it + 1, std::integral_constant<int, 0>{});
    }
    if (nit > 0) epilogue(it - 1);                                        // the last step's chunk (a padding iteration stores nothing)
#ifdef KD_LB_DBG
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&kd_lb_dbg[i], dbg_acc[i]);
#endif
    // column sums of the eight row groups -> one slab row per workgroup (fixed order: deterministic)
    kd_lds_barrier();
    float* red = reinterpret_cast<float*>(smem_raw);                      // [8][2][128] floats over the (dead) planes
    kd_st4(red + (rb * 2 + 0) * LBW + 4 * c4, s1);
    kd_st4(red + (rb * 2 + 1) * LBW + 4 * c4, s2);
    kd_lds_barrier();
    {
      const int st = tid >> 7, c = tid & 127;                             // 256 threads: statistic x column
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += red[(k * 2 + st) * LBW + c];
      g.partial[(b * 2 + st) * LBW + c] = t;
    }
  } else {
    // ======== matrix waves: the two GEMMs of the current chunk, nothing else (their MFMAs pace the SIMD's matrix pipe while the
    // vector wave of the same SIMD converts, loads and stores)
    const int j = wave - 4;                                              // 32-column block of (dy . W2); quadrant (j >> 1, j & 1) of dW2
    const int wn = j >> 1, wk = j & 1;
    const int col = 32 * j + r;
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
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][k][q] = 0.f;
    const int o_lane = 4 * h * LBW + col;                                // accumulator layout: register q is row (q & 3) + 8 (q >> 2) + 4 h
    const int a_row = r * LBW, a_kk = (h ^ lb_key(r)) << 3;              // A fragment of k-step u: row r, chunk (2u + h) ^ key(r)
#ifdef KD_LB_DBG
    unsigned long long dbg_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, dbg_t = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): the W2^T loads (nothing pending at the loop)
    kd_lds_barrier();
    const int nit2 = (nit + 1) & ~1;
    for (int it = 0; it < nit2; ++it) {
      const unsigned short* buf = lds + (it & 1) * LBBUF;
      f32x16 dacc;
#pragma unroll
      for (int q = 0; q < 16; ++q) dacc[q] = 0.f;
      // eight k-steps of the data gradient, each beside one (row half, n block, k block) group of the weight gradient: two
      // independent MFMA streams whose LDS reads hide behind each other's MFMAs
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bf16x8 ap[3], d[3], a[3];
        const unsigned short* p = buf + a_row + (a_kk ^ (16 * u));
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ap[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + pl * LBPL));
        const int ks = u >> 2, ni = (u >> 1) & 1, ki = u & 1;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          d[pl] = lb_tr_frag(buf + pl * LBPL, 16 * ks, 64 * wn + 32 * ni, lane);
          a[pl] = lb_tr_frag(buf + (3 + pl) * LBPL, 16 * ks, 64 * wk + 32 * ki, lane);
        }
#pragma unroll
        for (int t = 0; t < ((KD_LB_PROBE & 2) ? 1 : 6); ++t)
          dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[t]], Wb[u][PB[t]], dacc, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < ((KD_LB_PROBE & 4) ? 1 : 6); ++t)
          acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[PA[t]], a[PB[t]], acc[ni][ki], 0, 0, 0);
      }
      KD_LSTAMP(4);
      // (dy . W2) of this chunk -> stage tile (accumulator layout: 32 consecutive floats per half wave, conflict-free); the
      // vector waves turn it into G1 during the next step
      float* sb = stage + (it & 1) * LBXF + o_lane;
#pragma unroll
      for (int q = 0; q < 16; ++q) sb[((q & 3) + 8 * (q >> 2)) * LBW] = dacc[q];
      KD_LSTAMP(5);
      kd_lds_barrier();
      KD_LSTAMP(7);
#ifdef KD_LB_DBG
      dbg_acc[8] += 1;
#endif
    }
#ifdef KD_LB_DBG
    if (lane == 0) for (int i = 4; i < 9; ++i) atomicAdd(&kd_lb_dbg[i], dbg_acc[i]);
#endif
    kd_lds_barrier();                                                     // (the vector waves' reduction barriers)
    kd_lds_barrier();
    float* out = g.wslab + (size_t)b * (LBW * LBW);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        const int oc = 64 * wk + 32 * ki + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int row = 64 * wn + 32 * ni + (q & 3) + 8 * (q >> 2) + 4 * h;
          out[row * LBW + oc] = acc[ni][ki][q];
        }
      }
  }
}

// =====================================================================================================================
// Layer 1 (Conv1d 64 -> 128, lidar_encoder.py:29) in the same form.  Its input is layer 0's output, which never exists in HBM:
// the vector waves recompute a0 = relu(bn0(l0(point))) from the 16-byte point (kd_l0_raw: the evaluation order every other
// kernel uses).  dy1 = al*G1 + be*Y1 + ga (G1 already carries act1': the operand of kd_lidar_l1_dgrad / _wgrad with mact = none).
//   G0[32,64]   = (dy1 . W1) * relu'(z0)          never stored: BatchNorm-0 backward sums + the moments sum_m G0 * point
//   dW1[128,64] += dy1^T . a0
// Matrix waves 0 / 1: one 32x32 block of (dy1 . W1) each (W1^T rows as bf16 planes in registers); 2 / 3: four 32x32 tiles
// of dW1 each.  48 MFMAs per wave and chunk.  The a0 planes are [row][64 bf16] (128-byte rows), 16-byte chunks swizzled by
// bit 2 ^= (row >> 1) & 1: four consecutive rows of a transposing read fall into four different bank quarters.
constexpr int L1N = 128, L1K = 64;
constexpr int L1PD = LBCH * L1N, L1PA = LBCH * L1K;          // bf16 per dy / a0 plane
constexpr int L1BUF = 3 * L1PD + 3 * L1PA;                   // bf16 per buffer
constexpr int L1SF = LBCH * L1K;                             // floats per stage tile
constexpr size_t L1_LDS = (size_t)2 * L1BUF * 2 + (size_t)2 * L1SF * 4 + (size_t)2 * LBCH * 16;   // 72 KB planes + 16 KB stage + 1 KB points

struct L1Args {
  const float* G; const float* Y1;                         // dense [M,128]: masked gradient from layer 2's backward, raw layer-1 output
  const float* al; const float* be; const float* ga;      // BatchNorm-1 backward coefficients
  const float* pts;                                        // [M,4]
  const float* w0; const float* b0;                        // layer 0: [64][4], [64]
  const float* sc0; const float* sh0; const float* mean0; const float* inv0;
  const float* Wt;                                         // [64][128]: W1 transposed
  float* partial;                                          // out [grid][2][64]
  float* wslab;                                            // out [grid][128][64]
  float* m1slab;                                           // out [grid][4][64]
  int M;
};

__device__ __forceinline__ int la_off(int row, int c) { return row * L1K + ((c ^ (((row >> 1) & 1) << 2)) << 3); }
__device__ __forceinline__ bf16x8 la_tr_frag(const unsigned short* plane, int row0, int col0, int lane) {
  const int grp = lane >> 4, li = lane & 15;
  const int row = row0 + 8 * (grp >> 1) + (li >> 2);
  const int col = col0 + 16 * (grp & 1) + 4 * (li & 3);
  const unsigned short* p0 = plane + la_off(row, col >> 3) + (col & 7);
  const unsigned short* p1 = plane + la_off(row + 4, col >> 3) + (col & 7);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512, 1) void lidar_l1_bwd_kernel(L1Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem_raw);                 // [2][3 dy planes | 3 a0 planes]
  float* stage = reinterpret_cast<float*>(smem_raw + (size_t)2 * L1BUF * 2);         // [2][32][64]
  float4* ldp = reinterpret_cast<float4*>(stage + 2 * L1SF);                         // [2][32] points of the chunk in each buffer
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int M = g.M;
  const int nchunk = (M + LBCH - 1) / LBCH;
  const int G = gridDim.x, b = blockIdx.x;
  const int nit = b < nchunk ? (nchunk - b + G - 1) / G : 0;
  auto chunk_at = [&](int it) __attribute__((always_inline)) { return b + __builtin_amdgcn_readfirstlane(min(max(it, 0), nit - 1)) * G; };
  auto last_at = [&](int it) __attribute__((always_inline)) {
    const int vm = -(int)(it >= 0 && it < nit);
    return ((M - 1 - chunk_at(it) * LBCH) & vm) | ~vm;
  };
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};

  if (wave < 4) {
    // ======== vector waves
    const int c4 = tid & 31, rb = tid >> 5;                               // dy layout: float4 column group, rows rb + 8 i (i < 4)
    const int c4a = tid & 15, ra = tid >> 4;                              // a0 / epilogue layout: 4 of the 64 channels, rows ra + 16 i (i < 2)
    int cvo[4], cao[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) cvo[i] = lb_off(rb + 8 * i, c4 >> 1) + (c4 & 1) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) cao[i] = la_off(ra + 16 * i, c4a >> 1) + (c4a & 1) * 4;
    const float4 cal = kd_ld4(g.al + 4 * c4), cbe = kd_ld4(g.be + 4 * c4), cga = kd_ld4(g.ga + 4 * c4);
    float4 cw[4];                                                         // layer-0 weights of this thread's 4 channels
#pragma unroll
    for (int k = 0; k < 4; ++k) cw[k] = kd_ld4(g.w0 + (4 * c4a + k) * 4);
    const float4 cb = kd_ld4(g.b0 + 4 * c4a), cs = kd_ld4(g.sc0 + 4 * c4a), ch = kd_ld4(g.sh0 + 4 * c4a);
    const float4 cmean = kd_ld4(g.mean0 + 4 * c4a), cinv = kd_ld4(g.inv0 + 4 * c4a);
    float4 rg[2][4], ry[2][4], rp[2][2];
    float4 s1 = kd_zero4(), s2 = kd_zero4(), m1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) m1[k] = kd_zero4();
    auto load_rows4 = [&](const float* T, int chunk, float4 (&dst)[4]) __attribute__((always_inline)) {
      const float* base = T + (size_t)chunk * (LBCH * L1N);
      const int last = M - 1 - chunk * LBCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb + 8 * i;
        dst[i] = kd_ld4(base + (row < last ? row : last) * L1N + 4 * c4);
      }
    };
    auto load_pts = [&](int chunk, float4 (&dst)[2]) __attribute__((always_inline)) {
      const float* base = g.pts + (size_t)chunk * (LBCH * 4);
      const int last = M - 1 - chunk * LBCH;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = ra + 16 * i;
        dst[i] = kd_ld4(base + (row < last ? row : last) * 4);
      }
    };
    auto convert_store = [&](int last, const float4 (&sg)[4], const float4 (&sy)[4], const float4 (&sp)[2], int bufi) __attribute__((always_inline)) {
      unsigned short* buf = lds + bufi * L1BUF;
      {
        float4 vw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vw[i].x = fmaf(cal.x, sg[i].x, fmaf(cbe.x, sy[i].x, cga.x));
          vw[i].y = fmaf(cal.y, sg[i].y, fmaf(cbe.y, sy[i].y, cga.y));
          vw[i].z = fmaf(cal.z, sg[i].z, fmaf(cbe.z, sy[i].z, cga.z));
          vw[i].w = fmaf(cal.w, sg[i].w, fmaf(cbe.w, sy[i].w, cga.w));
        }
        uint2 hi[4], mid[4], lo[4];
        lb_split3_lockstep<4>(vw, hi, mid, lo);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned short* d = buf + cvo[i];
          *reinterpret_cast<uint2*>(d) = hi[i];
          *reinterpret_cast<uint2*>(d + L1PD) = mid[i];
          *reinterpret_cast<uint2*>(d + 2 * L1PD) = lo[i];
        }
      }
      {
        float4 vw[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float4 x = kd_l0_raw4(sp[i], cw, cb);
          vw[i] = make_float4(fmaxf(kd_affine(x.x, cs.x, ch.x), 0.f), fmaxf(kd_affine(x.y, cs.y, ch.y), 0.f),
                              fmaxf(kd_affine(x.z, cs.z, ch.z), 0.f), fmaxf(kd_affine(x.w, cs.w, ch.w), 0.f));
          if (c4a == 0) ldp[bufi * LBCH + ra + 16 * i] = sp[i];
        }
        uint2 hi[2], mid[2], lo[2];
        lb_split3_lockstep<2>(vw, hi, mid, lo);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned short* d = buf + 3 * L1PD + cao[i];
          *reinterpret_cast<uint2*>(d) = hi[i];
          *reinterpret_cast<uint2*>(d + L1PA) = mid[i];
          *reinterpret_cast<uint2*>(d + 2 * L1PA) = lo[i];
        }
      }
      if (last < LBCH - 1) {                    // tail chunk / padding iteration: rows beyond M contribute nothing to dW1
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (rb + 8 * i > last) {
            const uint2 z = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(buf + cvo[i]) = z;
            *reinterpret_cast<uint2*>(buf + L1PD + cvo[i]) = z;
            *reinterpret_cast<uint2*>(buf + 2 * L1PD + cvo[i]) = z;
          }
      }
    };
    // G0 of one chunk from the matrix waves' stage tile: mask, BatchNorm-0 backward sums, moments with the point
    auto epilogue = [&](int it) __attribute__((always_inline)) {
      const int last = last_at(it), bufi = it & 1;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = ra + 16 * i;
        const float4 d = kd_ld4(stage + bufi * L1SF + row * L1K + 4 * c4a);
        const float4 pt = ldp[bufi * LBCH + row];
        const bool ok = row <= last;
        const float4 x = kd_l0_raw4(pt, cw, cb);
        float4 v;
        v.x = (ok && kd_affine(x.x, cs.x, ch.x) > 0.f) ? d.x : 0.f;
        v.y = (ok && kd_affine(x.y, cs.y, ch.y) > 0.f) ? d.y : 0.f;
        v.z = (ok && kd_affine(x.z, cs.z, ch.z) > 0.f) ? d.z : 0.f;
        v.w = (ok && kd_affine(x.w, cs.w, ch.w) > 0.f) ? d.w : 0.f;
        s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
        s2.x = fmaf(v.x, (x.x - cmean.x) * cinv.x, s2.x);
        s2.y = fmaf(v.y, (x.y - cmean.y) * cinv.y, s2.y);
        s2.z = fmaf(v.z, (x.z - cmean.z) * cinv.z, s2.z);
        s2.w = fmaf(v.w, (x.w - cmean.w) * cinv.w, s2.w);
        const float pj[4] = {pt.x, pt.y, pt.z, pt.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m1[k].x = fmaf(v.x, pj[k], m1[k].x); m1[k].y = fmaf(v.y, pj[k], m1[k].y);
          m1[k].z = fmaf(v.z, pj[k], m1[k].z); m1[k].w = fmaf(v.w, pj[k], m1[k].w);
        }
      }
    };
    auto step = [&](int it, auto set_tag) __attribute__((always_inline)) {
      constexpr int S = decltype(set_tag)::value;
      epilogue(it - 1);
      __builtin_amdgcn_sched_barrier(0);
      convert_store(last_at(it + 1), rg[S], ry[S], rp[S], (it + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      const int c3 = chunk_at(it + 3);
      load_rows4(g.G, c3, rg[S]);
      load_rows4(g.Y1, c3, ry[S]);
      load_pts(c3, rp[S]);
      __builtin_amdgcn_sched_barrier(0);
      kd_lds_barrier();
    };
    for (int i = tid; i < (2 * L1SF + 2 * LBCH * 4) / 4; i += 256) kd_st4(stage + 4 * i, kd_zero4());     // stage + points: finite from the first read on
    if (nit > 0) {
      load_rows4(g.G, chunk_at(0), rg[0]); load_rows4(g.Y1, chunk_at(0), ry[0]); load_pts(chunk_at(0), rp[0]);
      load_rows4(g.G, chunk_at(1), rg[1]); load_rows4(g.Y1, chunk_at(1), ry[1]); load_pts(chunk_at(1), rp[1]);
    }
    kd_lds_barrier();                                                     // (the zero fill above, before the first points land in ldp)
    if (nit > 0) {
      convert_store(last_at(0), rg[0], ry[0], rp[0], 0);
      load_rows4(g.G, chunk_at(2), rg[0]); load_rows4(g.Y1, chunk_at(2), ry[0]); load_pts(chunk_at(2), rp[0]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                                   // vmcnt(0): enter the loop with an empty queue (see layer 2)
    kd_lds_barrier();
    int it = 0;
    for (; it < nit; it += 2) {
      step(it, std::integral_constant<int, 1>{});
      step(it + 1, std::integral_constant<int, 0>{});
    }
    if (nit > 0) epilogue(it - 1);
    // column sums of the sixteen row groups -> one slab row per workgroup (fixed order)
    kd_lds_barrier();
    float* red = reinterpret_cast<float*>(smem_raw);                      // [16][6][64] floats over the (dead) planes
    kd_st4(red + (ra * 6 + 0) * L1K + 4 * c4a, s1);
    kd_st4(red + (ra * 6 + 1) * L1K + 4 * c4a, s2);
#pragma unroll
    for (int k = 0; k < 4; ++k) kd_st4(red + (ra * 6 + 2 + k) * L1K + 4 * c4a, m1[k]);
    kd_lds_barrier();
    for (int i = tid; i < 6 * L1K; i += 256) {
      const int st = i / L1K, c = i % L1K;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += red[(k * 6 + st) * L1K + c];
      if (st < 2) g.partial[(b * 2 + st) * L1K + c] = t;
      else g.m1slab[(b * 4 + st - 2) * L1K + c] = t;
    }
  } else {
    // ======== matrix waves
    const int j = wave - 4;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (j < 2) {
      // (dy1 . W1): one 32-column block
      const int col = 32 * j + r;
      bf16x8 Wb[8][3];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* wp = g.Wt + col * L1N + 16 * u + 8 * h;
        uint2 h0, m0, l0, h1, m1, l1;
        kd_split3(kd_ld4(wp), h0, m0, l0);
        kd_split3(kd_ld4(wp + 4), h1, m1, l1);
        const u32x4 vh = {h0.x, h0.y, h1.x, h1.y}, vm = {m0.x, m0.y, m1.x, m1.y}, vl = {l0.x, l0.y, l1.x, l1.y};
        Wb[u][0] = __builtin_bit_cast(bf16x8, vh);
        Wb[u][1] = __builtin_bit_cast(bf16x8, vm);
        Wb[u][2] = __builtin_bit_cast(bf16x8, vl);
      }
      const int o_lane = 4 * h * L1K + col;
      const int a_row = r * L1N, a_kk = (h ^ lb_key(r)) << 3;
      __builtin_amdgcn_s_waitcnt(0x0F70);
      kd_lds_barrier();
      kd_lds_barrier();
      const int nit2 = (nit + 1) & ~1;
      for (int it = 0; it < nit2; ++it) {
        const unsigned short* buf = lds + (it & 1) * L1BUF;
        f32x16 dacc;
#pragma unroll
        for (int q = 0; q < 16; ++q) dacc[q] = 0.f;
        bf16x8 ap[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ap[0][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + a_row + a_kk + pl * L1PD));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u < 7) {
            const unsigned short* p = buf + a_row + (a_kk ^ (16 * (u + 1)));
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) ap[(u + 1) & 1][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + pl * L1PD));
          }
#pragma unroll
          for (int t = 0; t < 6; ++t) dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[u & 1][PA[t]], Wb[u][PB[t]], dacc, 0, 0, 0);
        }
        float* sb = stage + (it & 1) * L1SF + o_lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) sb[((q & 3) + 8 * (q >> 2)) * L1K] = dacc[q];
        kd_lds_barrier();
      }
      kd_lds_barrier();
      kd_lds_barrier();
    } else {
      // dW1: n blocks 2 (j - 2), 2 (j - 2) + 1, both k blocks
      const int nb0 = 2 * (j - 2);
      f32x16 acc[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][k][q] = 0.f;
      kd_lds_barrier();
      kd_lds_barrier();
      const int nit2 = (nit + 1) & ~1;
      for (int it = 0; it < nit2; ++it) {
        const unsigned short* buf = lds + (it & 1) * L1BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            bf16x8 d[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) d[pl] = lb_tr_frag(buf + pl * L1PD, 16 * ks, 32 * (nb0 + ni), lane);
#pragma unroll
            for (int ki = 0; ki < 2; ++ki) {
              bf16x8 a[3];
#pragma unroll
              for (int pl = 0; pl < 3; ++pl) a[pl] = la_tr_frag(buf + 3 * L1PD + pl * L1PA, 16 * ks, 32 * ki, lane);
#pragma unroll
              for (int t = 0; t < 6; ++t)
                acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[PA[t]], a[PB[t]], acc[ni][ki], 0, 0, 0);
            }
          }
        kd_lds_barrier();
      }
      kd_lds_barrier();
      kd_lds_barrier();
      float* out = g.wslab + (size_t)b * (L1N * L1K);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) {
          const int oc = 32 * ki + r;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = 32 * (nb0 + ni) + (q & 3) + 8 * (q >> 2) + 4 * h;
            out[row * L1K + oc] = acc[ni][ki][q];
          }
        }
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
  KD_REQUIRE(act2 == KD_ACT_RELU && act1 == KD_ACT_RELU, KD_ERR_ARG,
             "kd_lidar_l2_bwd: both layers' activations must be ReLU (lidar_encoder.py:28-34); other activations: kd_lidar_l2_dgrad + _wgrad");
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


// ---- layer 1 ------------------------------------------------------------------------------------------------------
int kd_lidar_l1_bwd_supported(int N1, int K0) { return kd_gemm_split_mode() && N1 == L1N && K0 == L1K; }
int64_t kd_lidar_l1_bwd_stat_rows(int64_t M) { return lb_grid(M); }
size_t kd_lidar_l1_bwd_ws_bytes(int64_t M, int N1, int K0) { return (size_t)lb_grid(M) * ((size_t)N1 * K0 + 4 * K0) * sizeof(float); }

// Training backward of point-MLP layer 1 in ONE kernel: dW1, the BatchNorm-0 backward sums of G0 = (dy1 . W1) * relu'(.) and
// the moments m1[j][c] = sum_m G0[m][c] * point[m][j] (kd_lidar_l1_dgrad with m1_out and G0 = NULL, + kd_lidar_l1_wgrad).
// dy1 = al*G + be*Y1 + ga.  G and Y1 dense [M,128]; Wt = W1^T [64][128]; act0 must be ReLU.
int kd_lidar_l1_bwd(const float* G, int64_t ldg, const float* Y1, int64_t ldy, const float* al, const float* be, const float* ga,
                    const float* Wt, const float* pts, const float* w0, const float* b0, const float* sc0, const float* sh0,
                    const float* mean0, const float* invstd0, int act0, float* partial, int64_t partial_rows, float* m1_out,
                    float* dW, int64_t M, int N1, int K0, void* ws, size_t ws_bytes, void* stream) {
  KD_REQUIRE(G && Y1 && al && be && ga && Wt && pts && w0 && b0 && sc0 && sh0 && mean0 && invstd0 && partial && m1_out && dW && ws && M > 0,
             KD_ERR_ARG, "kd_lidar_l1_bwd: bad args");
  KD_REQUIRE(kd_lidar_l1_bwd_supported(N1, K0), KD_ERR_SHAPE,
             "kd_lidar_l1_bwd: no instance for N1=%d K0=%d in the %s arithmetic (use kd_lidar_l1_dgrad + kd_lidar_l1_wgrad)", N1, K0,
             kd_gemm_split_mode() ? "split" : "exact-fp32");
  KD_REQUIRE(M < (int64_t)1 << 31 && ldg == N1 && ldy == N1, KD_ERR_SHAPE, "kd_lidar_l1_bwd: G and Y1 must be dense [M,128] matrices");
  KD_REQUIRE(act0 == KD_ACT_RELU, KD_ERR_ARG, "kd_lidar_l1_bwd: layer 0's activation must be ReLU (lidar_encoder.py:28)");
  KD_REQUIRE(kd_aligned16(G) && kd_aligned16(Y1) && kd_aligned16(al) && kd_aligned16(be) && kd_aligned16(ga) && kd_aligned16(Wt) &&
             kd_aligned16(pts) && kd_aligned16(w0) && kd_aligned16(b0) && kd_aligned16(sc0) && kd_aligned16(sh0) && kd_aligned16(mean0) &&
             kd_aligned16(invstd0) && kd_aligned16(ws), KD_ERR_ALIGN, "kd_lidar_l1_bwd: 16-byte alignment");
  const int grid_x = lb_grid(M);
  KD_REQUIRE(partial_rows == grid_x, KD_ERR_ARG, "kd_lidar_l1_bwd: statistics slab sized for %lld rows, this launch writes %d "
             "(kd_lidar_l1_bwd_stat_rows)", (long long)partial_rows, grid_x);
  KD_REQUIRE(ws_bytes >= kd_lidar_l1_bwd_ws_bytes(M, N1, K0), KD_ERR_WORKSPACE, "kd_lidar_l1_bwd: workspace too small (%zu B)", ws_bytes);
  static std::atomic<uint64_t> lds_raised{0};
  const hipError_t e = kd_raise_dynamic_lds((const void*)lidar_l1_bwd_kernel, L1_LDS, lds_raised);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_l1_bwd: cannot raise the dynamic LDS limit to %zu B: %s", L1_LDS, hipGetErrorString(e));
  float* wslab = (float*)ws;
  float* m1slab = wslab + (size_t)grid_x * N1 * K0;
  L1Args g{G, Y1, al, be, ga, pts, w0, b0, sc0, sh0, mean0, invstd0, Wt, partial, wslab, m1slab, (int)M};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lidar_l1_bwd_kernel, dim3(grid_x), dim3(512), L1_LDS, st, g);
  int rc = kd_check_launch("kd_lidar_l1_bwd");
  if (rc) return rc;
  rc = kd_slab_reduce_launch(wslab, grid_x, (int64_t)N1 * K0, dW, st);
  if (rc) return rc;
  return kd_slab_reduce_launch(m1slab, grid_x, (int64_t)4 * K0, m1_out, st);
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
