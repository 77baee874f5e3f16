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
//   waves 0-3 "convert + wgrad": load the NEXT chunk's Y2 / table rows / Y1 (one chunk ahead in registers), transform, cut
//     into three bf16 planes, store them row-major into the other half of a double-buffered LDS image; then accumulate their
//     64x64 quadrant of dW2 over the CURRENT chunk with operand fragments fetched by ds_read_b64_tr_b16 (the reduction index
//     of this GEMM is the matrix row);
//   waves 4-7 "dgrad + epilogue": one 32x32 block of G1 each; B operand = their 32 rows of W2^T as bf16 planes held in
//     REGISTERS for the whole launch (96 VGPRs), A operand = the dy planes of the current chunk (ds_read_b128); then the
//     epilogue of the streaming kernels (mask, BatchNorm-backward sums per lane across chunks, dword stores of 128-byte
//     row segments).
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

int kd_gemm_split_mode();     // kd_gemm.hip: 1 = bf16x6 split products (default), 0 = exact-fp32 MFMA

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

constexpr int LBW = 128;                 // channels of both operand tensors of the last layer (N2 = K1 = 128)
constexpr int LBCH = 32;                 // rows per chunk
constexpr int LBPL = LBCH * LBW;         // bf16 per plane
constexpr int LBBUF = 6 * LBPL;          // bf16 per buffer: 3 dy planes + 3 a planes
constexpr size_t LB_LDS = (size_t)2 * LBBUF * 2;     // bytes, double-buffered: 96 KB

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

__global__ __launch_bounds__(512, 1) void lidar_l2_bwd_kernel(LbArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* lds = reinterpret_cast<unsigned short*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t M = g.M;
  const int64_t nchunk = (M + LBCH - 1) / LBCH;
  const int64_t G = gridDim.x, b = blockIdx.x;
  const int64_t nit = b < nchunk ? (nchunk - b + G - 1) / G : 0;       // chunks b, b + G, b + 2G, ...
  constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};   // smallest terms first (as pw_gemm_kernel)

  if (wave < 4) {
    // =============================== role B: convert the next chunk, weight gradient of the current one ===============
    const int c4 = tid & 31, rb = tid >> 5;                               // float4 column group (fixed), first row
    const int wn = wave >> 1, wk = wave & 1;                              // 64x64 quadrant of dW2
    const float4 cal = kd_ld4(g.al + 4 * c4), cbe = kd_ld4(g.be + 4 * c4), cga = kd_ld4(g.ga + 4 * c4);
    const float4 cms = kd_ld4(g.sc2 + 4 * c4), cmh = kd_ld4(g.sh2 + 4 * c4);
    const float4 cas = kd_ld4(g.sc1 + 4 * c4), cah = kd_ld4(g.sh1 + 4 * c4);
    float4 ry[4], rx[4], rs[4], ra[4];
    int tr_cur[4], tr_nxt[4];
    auto fetch_rows = [&](int64_t chunk, int (&tr)[4]) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t gm = chunk * LBCH + rb + 8 * i;
        gm = gm < M ? gm : M - 1;
        tr[i] = g.trows[gm];
      }
    };
    auto load_chunk = [&](int64_t chunk, bool more, int64_t next_chunk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) tr_cur[i] = tr_nxt[i];                  // fetched one chunk ago: no dependent load here
      if (more) fetch_rows(next_chunk, tr_nxt);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t gm = chunk * LBCH + rb + 8 * i;
        gm = gm < M ? gm : M - 1;
        ry[i] = kd_ld4(g.Y2 + gm * g.ldy2 + 4 * c4);
        const int64_t o = (int64_t)(tr_cur[i] < 0 ? 0 : tr_cur[i]) * LBW + 4 * c4;
        rx[i] = kd_ld4(g.tmx + o);
        rs[i] = kd_ld4(g.tshare + o);
        ra[i] = kd_ld4(g.Y1 + gm * g.ldy1 + 4 * c4);
      }
    };
    auto convert_store = [&](int64_t chunk, unsigned short* buf) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rb + 8 * i;
        const bool ok = chunk * LBCH + row < M;
        const float4 x = ry[i], mx = rx[i], sv = rs[i];
        const float4 a = kd_affine_act4(x, cms, cmh, g.act2);
        const bool tv = tr_cur[i] >= 0;
        float4 v;
        v.x = kd_bwd_operand((tv && a.x > 0.f && a.x == mx.x) ? sv.x : 0.f, x.x, cal.x, cbe.x, cga.x, 0.f, 0.f, KD_ACT_NONE);
        v.y = kd_bwd_operand((tv && a.y > 0.f && a.y == mx.y) ? sv.y : 0.f, x.y, cal.y, cbe.y, cga.y, 0.f, 0.f, KD_ACT_NONE);
        v.z = kd_bwd_operand((tv && a.z > 0.f && a.z == mx.z) ? sv.z : 0.f, x.z, cal.z, cbe.z, cga.z, 0.f, 0.f, KD_ACT_NONE);
        v.w = kd_bwd_operand((tv && a.w > 0.f && a.w == mx.w) ? sv.w : 0.f, x.w, cal.w, cbe.w, cga.w, 0.f, 0.f, KD_ACT_NONE);
        v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        float4 w = kd_affine_act4(ra[i], cas, cah, g.act1);
        w = make_float4(ok ? w.x : 0.f, ok ? w.y : 0.f, ok ? w.z : 0.f, ok ? w.w : 0.f);
        uint2 hi, mid, lo;
        unsigned short* d = buf + lb_off(row, c4 >> 1) + (c4 & 1) * 4;
        kd_split3(v, hi, mid, lo);
        *reinterpret_cast<uint2*>(d) = hi;
        *reinterpret_cast<uint2*>(d + LBPL) = mid;
        *reinterpret_cast<uint2*>(d + 2 * LBPL) = lo;
        kd_split3(w, hi, mid, lo);
        *reinterpret_cast<uint2*>(d + 3 * LBPL) = hi;
        *reinterpret_cast<uint2*>(d + 4 * LBPL) = mid;
        *reinterpret_cast<uint2*>(d + 5 * LBPL) = lo;
      }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    if (nit > 0) {
      fetch_rows(b, tr_nxt);
      load_chunk(b, nit > 1, b + G);
      convert_store(b, lds);
      if (nit > 1) load_chunk(b + G, nit > 2, b + 2 * G);
    }
    kd_lds_barrier();
    for (int64_t it = 0; it < nit; ++it) {
      const int64_t chunk = b + it * G;
      if (it + 1 < nit) {
        convert_store(chunk + G, lds + ((it + 1) & 1) * LBBUF);
        if (it + 2 < nit) load_chunk(chunk + 2 * G, it + 3 < nit, chunk + 3 * G);
      }
      const unsigned short* buf = lds + (it & 1) * LBBUF;
#pragma unroll
      for (int ks = 0; ks < LBCH / 16; ++ks) {
        bf16x8 a[2][3];
#pragma unroll
        for (int ki = 0; ki < 2; ++ki)
#pragma unroll
          for (int p = 0; p < 3; ++p) a[ki][p] = lb_tr_frag(buf + (3 + p) * LBPL, 16 * ks, 64 * wk + 32 * ki, lane);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          bf16x8 d[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) d[p] = lb_tr_frag(buf + p * LBPL, 16 * ks, 64 * wn + 32 * ni, lane);
#pragma unroll
          for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int ki = 0; ki < 2; ++ki)
              acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d[PA[t]], a[ki][PB[t]], acc[ni][ki], 0, 0, 0);
        }
      }
      kd_lds_barrier();
    }
    float* out = g.wslab + b * (int64_t)(LBW * LBW);
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
    // =============================== role A: data gradient of the current chunk + its epilogue =========================
    const int j = wave - 4;                                              // 32-column block of G1
    const int col = 32 * j + r;
    bf16x8 Wb[8][3];                                                     // W2^T rows 32j + r, all 128 k, three planes: 96 VGPRs
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* wp = g.Wt + (int64_t)col * LBW + 16 * u + 8 * h;
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
    const int x_lane = 4 * h * (int)g.ldy1 + col, c_lane = 4 * h * (int)g.ldg1 + col;
    kd_lds_barrier();
    for (int64_t it = 0; it < nit; ++it) {
      const int64_t chunk = b + it * G, m0 = chunk * LBCH;
      const bool full = m0 + LBCH <= M;
      // the raw tensor whose activation is differentiated, in the accumulator layout (L2 hits: the converter waves read
      // these rows two chunks ago); issued before the k-loop, consumed after it
      float xr[16];
      const float* xbase = g.Y1 + m0 * g.ldy1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rbase = (q & 3) + 8 * (q >> 2);
        const bool rok = full || (m0 + rbase + 4 * h < M);
        xr[q] = rok ? xbase[(int64_t)rbase * g.ldy1 + x_lane] : 0.f;
      }
      const unsigned short* buf = lds + (it & 1) * LBBUF;
      f32x16 dacc;
#pragma unroll
      for (int q = 0; q < 16; ++q) dacc[q] = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        bf16x8 ap[3];
        const unsigned short* p = buf + lb_off(r, 2 * u + h);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) ap[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + pl * LBPL));
#pragma unroll
        for (int t = 0; t < 6; ++t) dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PA[t]], Wb[u][PB[t]], dacc, 0, 0, 0);
      }
      float* cbase = g.G1 + m0 * g.ldg1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rbase = (q & 3) + 8 * (q >> 2);
        const bool rok = full || (m0 + rbase + 4 * h < M);
        const float x = xr[q];
        float v = dacc[q] + zero;
        v *= kd_act_mask(kd_affine(x, esc, esh), g.act1);
        if (rok) { s1 += v; s2 = fmaf(v, (x - emean) * einv, s2); }
        float* dst = cbase + (int64_t)rbase * g.ldg1 + c_lane;
        if (full) {
          if (g.nt_store) __builtin_nontemporal_store(v, dst); else *dst = v;
        } else if (rok) {
          *dst = v;
        }
      }
      kd_lds_barrier();
    }
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
size_t kd_lidar_l2_bwd_ws_bytes(int64_t M, int N2, int K1) { return (size_t)lb_grid(M) * N2 * K1 * sizeof(float); }

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
  KD_REQUIRE(M < (int64_t)1 << 31 && ldy2 % 4 == 0 && ldg1 % 4 == 0 && ldy1 % 4 == 0 && ldy2 >= N2 && ldy1 >= K1 && ldg1 >= K1, KD_ERR_SHAPE,
             "kd_lidar_l2_bwd: row strides must be multiples of 4 and at least the channel count");
  KD_REQUIRE(kd_aligned16(Y2) && kd_aligned16(grid) && kd_aligned16(share) && kd_aligned16(Wt) && kd_aligned16(G1) && kd_aligned16(Y1) &&
             kd_aligned16(al) && kd_aligned16(be) && kd_aligned16(ga) && kd_aligned16(sc2) && kd_aligned16(sh2) && kd_aligned16(sc1) &&
             kd_aligned16(sh1) && kd_aligned16(ws), KD_ERR_ALIGN, "kd_lidar_l2_bwd: 16-byte alignment");
  KD_REQUIRE(act2 == KD_ACT_RELU || act2 == KD_ACT_RELU6, KD_ERR_ARG, "kd_lidar_l2_bwd: the scatter-max tables need a non-negative activation");
  const int grid_x = lb_grid(M);
  KD_REQUIRE(partial_rows == grid_x, KD_ERR_ARG, "kd_lidar_l2_bwd: statistics slab sized for %lld rows, this launch writes %d "
             "(kd_lidar_l2_bwd_stat_rows)", (long long)partial_rows, grid_x);
  KD_REQUIRE(ws_bytes >= kd_lidar_l2_bwd_ws_bytes(M, N2, K1), KD_ERR_WORKSPACE, "kd_lidar_l2_bwd: workspace too small (%zu B)", ws_bytes);
  static std::atomic<uint64_t> lds_raised{0};
  const hipError_t e = kd_raise_dynamic_lds((const void*)lidar_l2_bwd_kernel, LB_LDS, lds_raised);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_lidar_l2_bwd: cannot raise the dynamic LDS limit to %zu B: %s", LB_LDS, hipGetErrorString(e));
  LbArgs g{Y2, ldy2, rows, grid, share, al, be, ga, sc2, sh2, act2, Y1, ldy1, sc1, sh1, mean1, invstd1, act1, Wt, G1, ldg1, partial,
           (float*)ws, (int)M, kd_nt_store((size_t)M * K1 * sizeof(float))};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(lidar_l2_bwd_kernel, dim3(grid_x), dim3(512), LB_LDS, st, g);
  const int rc = kd_check_launch("kd_lidar_l2_bwd");
  if (rc) return rc;
  return kd_slab_reduce_launch((const float*)ws, grid_x, (int64_t)N2 * K1, dW, st);
}

}  // extern "C"
