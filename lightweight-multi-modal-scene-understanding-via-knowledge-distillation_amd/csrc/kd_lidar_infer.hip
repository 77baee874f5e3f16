// kd_lidar_infer.hip -- the whole LiDAR point MLP + BEV scatter-max of an eval-mode encoder (frozen KD teacher, validation,
// inference) in ONE kernel: SpatialLiDAREncoder.forward_vectorized (lidar_encoder.py:57-99) with BatchNorm in running-statistics
// mode -- Conv1d 4->64, 64->128, 128->128 (each + BN1d + ReLU, :25-35), then scatter_reduce_(amax) into the grid (:85-96).
//
// Round 2 ran it as two launches (kd_lidar_l1_fwd writes the [points, 128] layer-1 output, kd_lidar_l2_fwd_scatter reads it
// back: 13 GB of HBM traffic per step at 256 frames x 80 000 points).  In eval mode nothing separates the layers -- no batch
// statistics -- so here NO activation ever leaves the CU: a wave takes a 32-point slab from the 16-byte points to the grid.
//
//   layer 1 is computed TRANSPOSED:  Y1^T[128, 32 pts] = W1[128, 64] . a0^T[64, 32 pts]
//     A operand = W1 (bf16 planes resident in LDS), B operand = a0 = relu(bn0(l0(point))) built in registers from the lane's own
//     point (lane = point, 8 consecutive layer-0 channels per k-step: kd_l0_raw, the evaluation order of every other kernel).
//     The accumulator of a 32x32 MFMA block then holds, in lane = point, sixteen layer-1 channels of that point -- which IS an
//     A-operand fragment of the next GEMM (lane = row, 8 values along the reduction) once the reduction index is renumbered:
//     k-step s = 2t + e of layer 2 takes registers 8e .. 8e+7 of block t, i.e. channels 32t + 16e + 4h + (i & 3) + 8 (i >> 2).
//   layer 2:  Y2[32 pts, 128] = a1[32, 128] . W2p^T with W2's columns stored in that renumbered order (W2p, LDS resident):
//     no LDS round trip, no shuffle between the two GEMMs; BN1 + ReLU + the bf16x3 split run on the accumulator registers.
//   epilogue: BN2 + ReLU on the accumulators (lane = channel), running maximum over the lane's consecutive rows of one cell
//     (the points arrive sorted by cell), unsigned atomicMax of the fp32 bit pattern at cell boundaries (values >= 0).
//
// Arithmetic: the six bf16 piece products per fp32 product of the other GEMM kernels, same order per k-step; layer 1's sums
// are those of kd_lidar_l1_fwd bit for bit, layer 2 adds the same sixteen products per k-step in a different (fixed) slot
// order inside the MFMA, so its sums may differ from kd_lidar_l2_fwd_scatter's in the last bit.  Split arithmetic only.
#include "kd_gemm_args.h"

#include <atomic>

int kd_gemm_split_mode();     // kd_gemm.hip

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int IW = 8;                       // waves per workgroup (two per SIMD share the resident weights)
constexpr int K0 = 64, N1 = 128, N2 = 128;  // layer widths (lidar_encoder.py:26-34)
constexpr int W1PL = N1 * K0, W2PL = N2 * N1;                         // bf16 per plane
// NP = 3: the fp32-grade split arithmetic (three bf16 planes per operand, six piece products); NP = 1: the bf16-storage inference
// mode (kdrt/bf16.py) -- operands rounded to bf16 once, one product, fp32 accumulation: what kd_bf16_pwconv computes, minus the
// bf16 [points, 128] layer-1 tensor in HBM
template <int NP> constexpr size_t li_lds() { return (size_t)NP * (W1PL + W2PL) * 2 + (size_t)(7 * K0 + 3 * N1) * 4; }   // 147,456 / 49,152 + 3,328 B

struct LiArgs {
  const float* pts;                         // [P, 4] (compacted / cell-sorted in-range points first)
  const int* cell;                          // [P] flat (frame, cell) grid row of each point (< 0: skip)
  const int* p_dev;                         // optional device-side row count (<= P)
  const float* w0; const float* b0; const float* sc0; const float* sh0;          // layer 0 + eval BN coefficients
  const float* W1; const float* bias1; const float* sc1; const float* sh1;       // [128][64]
  const float* W2; const float* bias2; const float* sc2; const float* sh2;       // [128][128]
  float* grid;                              // [cells][128], zero-initialised by the host wrapper
  int P;
};

// 16-byte chunk swizzle of a weight row of CPR chunks (as kd_gemm_stream_kernel.h): conflict-free ds_read_b128
template <int CPR> __device__ __forceinline__ int li_key(int n) { return CPR % 16 == 0 ? (n & 15) : ((n >> 1) & 7); }

template <int NP>
__global__ __launch_bounds__(64 * IW, 1) void lidar_mlp_scatter_infer_kernel(LiArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* W1h = reinterpret_cast<unsigned short*>(smem_raw);            // [NP][128][64] bf16, swizzled
  unsigned short* W2h = W1h + NP * W1PL;                                         // [NP][128][128] bf16, columns renumbered, swizzled
  float* T0 = reinterpret_cast<float*>(W2h + NP * W2PL);                         // [7][64]: w0.x .y .z .w, b0, sc0, sh0
  float* T1 = T0 + 7 * K0;                                                       // [3][128]: bias1, sc1, sh1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- prologue: weights -> three bf16 planes in LDS (once per workgroup), coefficient tables -------------------------------
  for (int i = tid; i < N1 * K0 / 4; i += 64 * IW) {
    const int n = i / (K0 / 4), k4 = i % (K0 / 4);
    uint2 hi, mid, lo;
    kd_split3(kd_ld4(g.W1 + n * K0 + 4 * k4), hi, mid, lo);
    unsigned short* d = W1h + n * K0 + ((k4 >> 1) ^ li_key<K0 / 8>(n)) * 8 + (k4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = hi;
    if (NP == 3) { *reinterpret_cast<uint2*>(d + W1PL) = mid; *reinterpret_cast<uint2*>(d + 2 * W1PL) = lo; }
  }
  for (int i = tid; i < N2 * N1 / 4; i += 64 * IW) {
    const int n = i / (N1 / 4), q4 = i % (N1 / 4);                    // q4: group of 4 consecutive renumbered positions k' = 4 q4 ..
    const int s = q4 >> 2, hh = (q4 >> 1) & 1, gq = q4 & 1;           // k' = 16 s + 8 hh + 4 gq + (0..3)
    const int ch = 32 * (s >> 1) + 16 * (s & 1) + 4 * hh + 8 * gq;    // ... holds layer-1 channels ch .. ch + 3
    uint2 hi, mid, lo;
    kd_split3(kd_ld4(g.W2 + n * N1 + ch), hi, mid, lo);
    unsigned short* d = W2h + n * N1 + ((q4 >> 1) ^ li_key<N1 / 8>(n)) * 8 + (q4 & 1) * 4;
    *reinterpret_cast<uint2*>(d) = hi;
    if (NP == 3) { *reinterpret_cast<uint2*>(d + W2PL) = mid; *reinterpret_cast<uint2*>(d + 2 * W2PL) = lo; }
  }
  for (int k = tid; k < K0; k += 64 * IW) {
    const float4 w = kd_ld4(g.w0 + 4 * k);
    T0[k] = w.x; T0[K0 + k] = w.y; T0[2 * K0 + k] = w.z; T0[3 * K0 + k] = w.w;
    T0[4 * K0 + k] = g.b0[k]; T0[5 * K0 + k] = g.sc0[k]; T0[6 * K0 + k] = g.sh0[k];
  }
  for (int c = tid; c < N1; c += 64 * IW) { T1[c] = g.bias1 ? g.bias1[c] : 0.f; T1[N1 + c] = g.sc1[c]; T1[2 * N1 + c] = g.sh1[c]; }
  kd_lds_barrier();

  int P = g.P;
  if (g.p_dev) { const int pv = *g.p_dev; P = pv < P ? pv : P; }
  const int nslab = (P + 31) / 32;
  const int wtot = gridDim.x * IW, wid = blockIdx.x * IW + wave;
  // per-lane epilogue constants: channel 32 j + r of layer 2
  float eb[4], es[4], eh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int c = 32 * j + r; eb[j] = g.bias2 ? g.bias2[c] : 0.f; es[j] = g.sc2[c]; eh[j] = g.sh2[c]; }
  const int f1 = h ^ li_key<K0 / 8>(r), f2 = h ^ li_key<N1 / 8>(r);      // physical chunk of k-step u = (2u) ^ f
  constexpr int NX = NP == 3 ? 6 : 1;                                     // piece products per product
  constexpr int PA[6] = {NP == 3 ? 0 : 0, 2, 1, 0, 1, 0}, PB[6] = {NP == 3 ? 2 : 0, 0, 1, 1, 0, 0};  // (activation plane, weight plane), smallest terms first
  unsigned* ugrid = reinterpret_cast<unsigned*>(g.grid);

  for (int s = wid; s < nslab; s += wtot) {
    const int m0 = s * 32;
    // keeps the table / weight reads inside the slab (hipcc would hoist them out of the loop and spill)
    int pin = 0;
    asm volatile("" : "+v"(pin));
    const unsigned short* W1p = W1h + pin;
    const unsigned short* W2p = W2h + pin;
    const float* T0p = T0 + pin;
    const float* T1p = T1 + pin;
    const int prow = m0 + r < P ? m0 + r : P - 1;
    const float4 pt = kd_ld4(g.pts + (size_t)prow * 4);
    // cell index of the sixteen rows this lane holds in the layer-2 accumulators (register q: row (q & 3) + 8 (q >> 2) + 4 h)
    int cell[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const int c = g.cell[row < P ? row : P - 1];
      cell[q] = row < P ? c : -1;
    }

    // ---- layer 1, transposed: acc1[t] = W1[32t .. 32t+31][:] . a0^T ---------------------------------------------------------
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      // B fragment: a0[point r][k = 16u + 8h .. +7]
      uint32_t bh[4], bm[4], bl[4];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kb = 16 * u + 8 * h + 4 * e;
        const float4 wx = kd_ld4(T0p + kb), wy = kd_ld4(T0p + K0 + kb), wz = kd_ld4(T0p + 2 * K0 + kb), ww = kd_ld4(T0p + 3 * K0 + kb);
        const float4 bb = kd_ld4(T0p + 4 * K0 + kb), sc = kd_ld4(T0p + 5 * K0 + kb), sh = kd_ld4(T0p + 6 * K0 + kb);
        const float v0 = fmaxf(kd_affine(kd_l0_raw(pt, make_float4(wx.x, wy.x, wz.x, ww.x), bb.x), sc.x, sh.x), 0.f);
        const float v1 = fmaxf(kd_affine(kd_l0_raw(pt, make_float4(wx.y, wy.y, wz.y, ww.y), bb.y), sc.y, sh.y), 0.f);
        const float v2 = fmaxf(kd_affine(kd_l0_raw(pt, make_float4(wx.z, wy.z, wz.z, ww.z), bb.z), sc.z, sh.z), 0.f);
        const float v3 = fmaxf(kd_affine(kd_l0_raw(pt, make_float4(wx.w, wy.w, wz.w, ww.w), bb.w), sc.w, sh.w), 0.f);
        kd_split_pair(v0, v1, bh[2 * e], bm[2 * e], bl[2 * e]);
        kd_split_pair(v2, v3, bh[2 * e + 1], bm[2 * e + 1], bl[2 * e + 1]);
      }
      const u32x4 ph = {bh[0], bh[1], bh[2], bh[3]}, pm = {bm[0], bm[1], bm[2], bm[3]}, pl = {bl[0], bl[1], bl[2], bl[3]};
      const bf16x8 bp[3] = {__builtin_bit_cast(bf16x8, ph), __builtin_bit_cast(bf16x8, pm), __builtin_bit_cast(bf16x8, pl)};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned short* wp = W1p + (32 * t + r) * K0 + ((2 * u) ^ f1) * 8;
        bf16x8 ap[3];
#pragma unroll
        for (int pl2 = 0; pl2 < NP; ++pl2) ap[pl2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wp + pl2 * W1PL));
#pragma unroll
        for (int x = 0; x < NX; ++x)     // (weight plane PB, activation plane PA): the same six products in the same order as kd_lidar_l1_fwd
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[PB[x]], bp[PA[x]], acc[t], 0, 0, 0);
      }
    }

    // ---- BN1 + ReLU on the accumulators -> the eight A fragments of layer 2 (lane = point) -----------------------------------
    // register q of block t: channel 32t + (q & 3) + 8 (q >> 2) + 4h; fragment s = 2t + e takes q = 8e .. 8e + 7
    bf16x8 a1[8][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        uint32_t fh[4], fm[4], fl[4];
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) {
          const int q0 = 8 * e + 4 * gq, c0 = 32 * t + 16 * e + 8 * gq + 4 * h;       // four consecutive channels c0 .. c0 + 3
          const float4 bi = kd_ld4(T1p + c0), sc = kd_ld4(T1p + N1 + c0), sh = kd_ld4(T1p + 2 * N1 + c0);
          const float v0 = fmaxf(kd_affine(acc[t][q0] + bi.x, sc.x, sh.x), 0.f);
          const float v1 = fmaxf(kd_affine(acc[t][q0 + 1] + bi.y, sc.y, sh.y), 0.f);
          const float v2 = fmaxf(kd_affine(acc[t][q0 + 2] + bi.z, sc.z, sh.z), 0.f);
          const float v3 = fmaxf(kd_affine(acc[t][q0 + 3] + bi.w, sc.w, sh.w), 0.f);
          kd_split_pair(v0, v1, fh[2 * gq], fm[2 * gq], fl[2 * gq]);
          kd_split_pair(v2, v3, fh[2 * gq + 1], fm[2 * gq + 1], fl[2 * gq + 1]);
        }
        const u32x4 ph = {fh[0], fh[1], fh[2], fh[3]}, pm = {fm[0], fm[1], fm[2], fm[3]}, pl = {fl[0], fl[1], fl[2], fl[3]};
        a1[2 * t + e][0] = __builtin_bit_cast(bf16x8, ph);
        a1[2 * t + e][1] = __builtin_bit_cast(bf16x8, pm);
        a1[2 * t + e][2] = __builtin_bit_cast(bf16x8, pl);
      }

    // ---- layer 2: acc[j] = a1 . W2p[32j .. 32j+31][:]^T -------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int sk = 0; sk < 8; ++sk)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned short* wp = W2p + (32 * j + r) * N1 + ((2 * sk) ^ f2) * 8;
        bf16x8 bp[3];
#pragma unroll
        for (int pl2 = 0; pl2 < NP; ++pl2) bp[pl2] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wp + pl2 * W2PL));
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[sk][PA[x]], bp[PB[x]], acc[j], 0, 0, 0);
      }

    // ---- epilogue: BN2 + ReLU, running maximum over the lane's rows of one cell, atomicMax at cell boundaries ------------------
    // (cell[] is uniform over the 32 lanes of a half wave: the boundary tests do not diverge inside it)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float run = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float v = fmaxf(kd_affine(acc[j][q] + eb[j], es[j], eh[j]), 0.f);
        const bool first = q == 0 || cell[q] != cell[q > 0 ? q - 1 : 0];
        const bool last = q == 15 || cell[q] != cell[q < 15 ? q + 1 : q];
        run = (first || v > run) ? v : run;
        if (last && cell[q] >= 0 && run > 0.f) atomicMax(ugrid + (size_t)cell[q] * N2 + 32 * j + r, __float_as_uint(run));
      }
    }
  }
}

}  // namespace

extern "C" {

// 1 when kd_lidar_mlp_scatter_infer has an instance for this encoder (widths 4 -> 64 -> 128 -> 128, split arithmetic)
int kd_lidar_mlp_scatter_infer_supported(int C0, int C1, int C2) { return kd_gemm_split_mode() && C0 == K0 && C1 == N1 && C2 == N2; }

// Eval-mode point MLP + BEV scatter-max in one kernel (see the head of this file).  All three activations are ReLU; sc* / sh*
// are the eval BatchNorm coefficients (kd_bn_eval_coeffs); pts are the compacted in-range points with their grid rows in
// `cell` (< 0: skip), p_dev an optional device-side count; grid [ncells][128] is zeroed here.
static int li_launch(int np, const char* who, const float* pts, const int* cell, const int* p_dev, const float* w0, const float* b0,
                     const float* sc0, const float* sh0, const float* W1, const float* bias1, const float* sc1, const float* sh1,
                     const float* W2, const float* bias2, const float* sc2, const float* sh2, float* grid, int64_t ncells, int64_t P,
                     void* stream) {
  KD_REQUIRE(pts && cell && w0 && b0 && sc0 && sh0 && W1 && sc1 && sh1 && W2 && sc2 && sh2 && grid && ncells > 0 && P > 0, KD_ERR_ARG,
             "%s: bad args", who);
  KD_REQUIRE(P < (int64_t)1 << 31, KD_ERR_SHAPE, "%s: too many points", who);
  KD_REQUIRE(kd_aligned16(pts) && kd_aligned16(w0) && kd_aligned16(W1) && kd_aligned16(W2), KD_ERR_ALIGN, "%s: 16-byte alignment", who);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(grid, 0, (size_t)ncells * N2 * sizeof(float), st);
  KD_REQUIRE(e == hipSuccess, (int)e, "%s: memset failed: %s", who, hipGetErrorString(e));
  const int64_t nslab = (P + 31) / 32, want = (nslab + IW - 1) / IW;
  LiArgs g{pts, cell, p_dev, w0, b0, sc0, sh0, W1, bias1, sc1, sh1, W2, bias2, sc2, sh2, grid, (int)P};
  if (np == 3) {
    static std::atomic<uint64_t> raised3{0};
    e = kd_raise_dynamic_lds((const void*)lidar_mlp_scatter_infer_kernel<3>, li_lds<3>(), raised3);
    KD_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise the dynamic LDS limit to %zu B: %s", who, li_lds<3>(), hipGetErrorString(e));
    hipLaunchKernelGGL(lidar_mlp_scatter_infer_kernel<3>, dim3((int)(want < 256 ? want : 256)), dim3(64 * IW), li_lds<3>(), st, g);
  } else {
    static std::atomic<uint64_t> raised1{0};
    e = kd_raise_dynamic_lds((const void*)lidar_mlp_scatter_infer_kernel<1>, li_lds<1>(), raised1);
    KD_REQUIRE(e == hipSuccess, (int)e, "%s: cannot raise the dynamic LDS limit to %zu B: %s", who, li_lds<1>(), hipGetErrorString(e));
    hipLaunchKernelGGL(lidar_mlp_scatter_infer_kernel<1>, dim3((int)(want < 256 ? want : 256)), dim3(64 * IW), li_lds<1>(), st, g);
  }
  return kd_check_launch(who);
}

int kd_lidar_mlp_scatter_infer(const float* pts, const int* cell, const int* p_dev, const float* w0, const float* b0,
                               const float* sc0, const float* sh0, const float* W1, const float* bias1, const float* sc1,
                               const float* sh1, const float* W2, const float* bias2, const float* sc2, const float* sh2,
                               float* grid, int64_t ncells, int64_t P, int C0, int C1, int C2, void* stream) {
  KD_REQUIRE(kd_lidar_mlp_scatter_infer_supported(C0, C1, C2), KD_ERR_SHAPE,
             "kd_lidar_mlp_scatter_infer: no instance for widths %d / %d / %d in the %s arithmetic", C0, C1, C2,
             kd_gemm_split_mode() ? "split" : "exact-fp32");
  return li_launch(3, "kd_lidar_mlp_scatter_infer", pts, cell, p_dev, w0, b0, sc0, sh0, W1, bias1, sc1, sh1, W2, bias2, sc2, sh2, grid, ncells, P,
                   stream);
}

// The same encoder in the bf16-storage inference mode (kdrt/bf16.py): operands rounded to bf16, one MFMA product per element,
// fp32 accumulation, fp32 grid -- kd_bf16_pwconv(a_kind 3) + kd_bf16_pwconv(epi 4) without the bf16 [points, 128] tensor between them.
int kd_bf16_lidar_mlp_scatter_supported(int C0, int C1, int C2) { return C0 == K0 && C1 == N1 && C2 == N2; }
int kd_bf16_lidar_mlp_scatter(const float* pts, const int* cell, const int* p_dev, const float* w0, const float* b0,
                              const float* sc0, const float* sh0, const float* W1, const float* bias1, const float* sc1,
                              const float* sh1, const float* W2, const float* bias2, const float* sc2, const float* sh2,
                              float* grid, int64_t ncells, int64_t P, int C0, int C1, int C2, void* stream) {
  KD_REQUIRE(kd_bf16_lidar_mlp_scatter_supported(C0, C1, C2), KD_ERR_SHAPE, "kd_bf16_lidar_mlp_scatter: no instance for widths %d / %d / %d", C0, C1, C2);
  return li_launch(1, "kd_bf16_lidar_mlp_scatter", pts, cell, p_dev, w0, b0, sc0, sh0, W1, bias1, sc1, sh1, W2, bias2, sc2, sh2, grid, ncells, P,
                   stream);
}

}  // extern "C"
