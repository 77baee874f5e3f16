// kd_gemm_args.h -- argument block and split-arithmetic helpers shared by the 1x1-convolution GEMM kernels
// (kd_gemm.hip: tiled kernels; kd_gemm_stream.hip: weight-resident streaming kernels).
#pragma once
#include "kd_common.h"

struct GemmArgs {
  const float* A; int64_t lda;        // PRO0/1: raw activations; PRO2: D (gradient)
  const float* A2; int64_t lda2;      // PRO2: X raw (the conv output whose BN is differentiated)
  const float* p0; const float* p1; const float* p2;   // PRO1: sc, sh ; PRO2: al, be, ga  (per K)
  const float* p3; const float* p4;   // PRO2 with mask: sc, sh of the masked activation (per K)
  int pro; int pro_act;
  const float* W;                     // [N][K] row-major
  const float* bias;                  // [N] or null
  float* C; int64_t ldc;
  const float* addend; int64_t ldadd; // optional: C += addend (before the EPI2 mask)
  const float* X; int64_t ldx;        // EPI2: raw tensor whose activation is differentiated [M,N]
  const float* esc; const float* esh; const float* emean; const float* einv; int epi_act;
  float* partial;                     // EPI1/2: [rowblocks][2][N]
  int M, K, N;
  const int* m_dev;                   // optional: device-side row count (<= M); rows beyond it are skipped
  const float* l0w; const float* l0b; // PRO3 / EPI3: LiDAR layer-0 weight [C0][4] and bias [C0] (A or X = points [M,4])
  // PRO4: the scatter-max gradient rebuilt on load.  A = Y raw [M,K]; trows[m] = grid row of point m (< 0: none);
  // tmx / tshare = [cells][K] tables (cell maximum, dout / holders): G = (v > 0 && v == mx) ? share : 0 with
  // v = act(Y*p3 + p4), then the PRO2 formula al*G + be*Y + ga.
  const float* tmx; const float* tshare; const int* trows;
  // EPI3 only: m1slab [rowblocks][4][N] receives per-block sums of G0[m][n] * point[m][j] (the part of the layer-0
  // weight gradient that depends on this GEMM's result); with it set, C may be NULL and G0 is never stored.
  float* m1slab;
  int nt_store;                       // C is large (>= 64 MB): store it with the non-temporal hint
};


typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// two fp32 -> packed (hi | mid | lo) bf16 pairs, each piece rounded to nearest even by v_cvt_pk_bf16_f32:
// |mid| <= 2^-9 |x|, |lo| <= 2^-17 |x|, x - (hi + mid + lo) <= 2^-26 |x|, residual signs unbiased.
__device__ __forceinline__ void kd_split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  v[0] -= __uint_as_float(hi << 16);                 // exact (Sterbenz-like: hi shares the leading bits of x)
  v[1] -= __uint_as_float(hi & 0xffff0000u);
  mid = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
  v[0] -= __uint_as_float(mid << 16);
  v[1] -= __uint_as_float(mid & 0xffff0000u);
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void kd_split3(float4 v, uint2& hi, uint2& mid, uint2& lo) {
  kd_split_pair(v.x, v.y, hi.x, mid.x, lo.x);
  kd_split_pair(v.z, v.w, hi.y, mid.y, lo.y);
}


struct WgradArgs {
  const float* D; int64_t ldd;        // dY raw gradient or G (d_mode 2)
  const float* X; int64_t ldx;        // d_mode 2: raw conv output X[M,N]
  const float* al; const float* be; const float* ga; const float* msc; const float* msh; int d_mode; int d_act;
  const float* A; int64_t lda;        // raw input activations [M,K]
  const float* asc; const float* ash; int a_mode; int a_act;
  float* slab;                        // [nsplit][N][K]
  int M, N, K;
  int rows_per_split;                 // multiple of the chunk height
  const float* l0w; const float* l0b; // AMODE 2: A = act(bn(layer0(point))), g.A = points [M,4]
  // DMODE 3: D = Y raw (also the BN-backward X); the scatter-max gradient G is rebuilt from trows / tmx / tshare exactly
  // as in pw_gemm_kernel PRO4 (msc / msh / d_act = Y's BatchNorm + activation)
  const float* tmx; const float* tshare; const int* trows;
  int xcd_order;                      // 1: the output tiles of one row slice go to blocks that share an XCD (see the kernel)
};

//
// SPLIT (same arithmetic as pw_gemm_kernel<.., SPLIT>): the reduction index of this GEMM is the matrix ROW m, so the
// MFMA operands are COLUMNS of the staged tiles.  The bf16 planes stay row-major [m][n] in LDS (8-byte stores, as
// loaded) and the fragments are fetched with ds_read_b64_tr_b16, gfx950's transposing LDS read: a 16-lane group reads
// a 4-row x 16-column block and each lane receives one column of it -- four consecutive m for its n.  Rows are padded
// by 32 bf16 so that the four 64-byte row segments of a block fall in four different bank quarters.
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ bf16x8 kd_tr_frag(const unsigned short* plane, int ld, int row0, int col0, int lane) {
  // fragment of the 32(col) x 16(row) operand block at (row0, col0): lane (r = lane & 31, h = lane >> 5) gets rows
  // row0 + 8h .. +7 of column col0 + r
  const int grp = lane >> 4, li = lane & 15;
  const unsigned short* p0 = plane + (row0 + 8 * (grp >> 1) + (li >> 2)) * ld + col0 + 16 * (grp & 1) + 4 * (li & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p0 + 4 * ld));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}


// kd_wgrad_rs.hip: the role-specialised weight-gradient kernel (round 4).  1: launched (dW written through the slab + reduce),
// 0: shape / mode not covered (the caller uses pw_wgrad_kernel), < 0: error.  kd_wgrad_rs_ws_bytes: slab bytes it needs (0: not covered).
int kd_wgrad_rs_launch(const WgradArgs& g, size_t ws_bytes, float* dW, hipStream_t st);
size_t kd_wgrad_rs_ws_bytes(int64_t M, int N, int K);

// kd_gemm_stream.hip: returns 1 if a streaming kernel took the launch, 0 if the shape / mode is not covered (the caller
// then uses the tiled kernel), < 0 on error.  kd_stream_stat_rows: rows of the BN-statistics slab that launch writes
// (0: not covered).
int kd_gemm_stream_launch(GemmArgs& g, int pro, int epi, hipStream_t st);
int kd_gemm_stream_stat_rows(int64_t M, int K, int N, int pro, int epi, bool add);
