// kd_block.hip -- inference-mode fusion of a depthwise-separable tail:
//     depthwise 3x3 (input BatchNorm + activation applied on load) -> BatchNorm -> activation -> 1x1 conv -> BatchNorm
//     -> activation (+ residual)
// in ONE kernel, for the eval-mode / no-grad forward (frozen KD teacher, validation): the last two units of an
// InvertedResidual (camera_encoder.py:30-42: dw 3x3 + BN + ReLU6, project 1x1 + BN, + x) and every DWSeparableConv
// (fusion_module.py:25-34: dw 3x3 + BN + ReLU, 1x1 + BN + ReLU).  In eval mode all BatchNorm coefficients are known before
// the kernel starts, so the depthwise output -- the widest tensor of the block (6x the block's channels) -- never goes to
// HBM: the separate kernels write it once and read it once.
//
// A workgroup owns a tile of TH x TW output pixels of one image (8 x 16 at stride 1, 8 x 8 at stride 2 = the M-tile of the
// GEMM: 128 / 64 rows) and ALL output channels (32, 64 or 128), and walks the hidden channels in chunks of 32:
//   1. the chunk's input halo tile ((TH-1)*S+3) x ((TW-1)*S+3) pixels x 32 channels is loaded (registers, one chunk ahead),
//      activated (deferred BatchNorm + activation of the producer, zero outside the image) and staged in LDS as fp32;
//      the chunk of the 1x1 weight is cut into its three bf16 planes and staged too;
//   2. thread (pixel, channel quad) computes the depthwise output from LDS (same fma order as dw_fwd_sw_kernel), applies
//      BatchNorm + activation, cuts the four values into three bf16 pieces and writes them where the MFMA A operand wants them;
//   3. the four waves accumulate the 32-wide K-chunk on v_mfma_f32_32x32x16_bf16 (six piece products per product, fp32
//      accumulate) -- the same arithmetic in the same order as pw_gemm_kernel<.., SPLIT> / pw_stream_kernel.
// Epilogue: BatchNorm + activation (+ residual) straight from the accumulators, dword stores (EPI5 of the GEMM kernels).
// The result is bit-identical to kd_dwconv3x3_fwd followed by kd_pwconv_gemm(pro 1, epi 5) in the split arithmetic
// (tests/test_gpu_units.py::test_dw_pw_inference_fusion_same_bits).
#include "kd_gemm_args.h"

namespace {

struct DwPwArgs {
  const float* x; const float* isc; const float* ish; int iact;     // depthwise input [B,H,W,Ch]: raw + deferred BN (isc null: as is)
  const float* wd;                                                  // depthwise taps [Ch][9]
  const float* dsc; const float* dsh; int dact;                     // BatchNorm (eval affine) + activation behind the depthwise conv
  const float* wp; const float* pbias;                              // 1x1 weight [Cout][Ch], bias [Cout] or null
  const float* psc; const float* psh; int pact;                     // BatchNorm + activation behind the 1x1 conv
  const float* res; int64_t ldres;                                  // optional residual [B*Ho*Wo][ldres], added last
  float* out; int64_t ldo;                                          // [B*Ho*Wo][ldo]
  int B, H, W, Ch, Ho, Wo;
  int tiles_x, tiles_y;
};

// 256 threads and 32 hidden channels per chunk: two workgroups per CU, so one's barriers and load waits overlap the other's
// phases.  (512 threads with 64-channel chunks -- half the barriers, one workgroup per CU -- measured 0-35 % SLOWER.)
constexpr int NTHR = 256, KC = 32, CQN = KC / 4;                    // threads per workgroup; hidden channels per chunk; channel quads per chunk
constexpr int XLD = KC + 4;                                         // floats per staged pixel (pad: rows stay 16-byte aligned)
// bf16 offset of 16-byte chunk `chunk` in a 64-byte row, XOR-swizzled by (row >> 2) & 3 (the LDS image of pw_gemm_kernel<.., SPLIT>:
// conflict-free for the ds_read_b128 lane groups and for the 8-byte stores)
__device__ __forceinline__ int sw_off(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 8; }

template <int STRIDE, int NCB>                                       // NCB = Cout / 32
__global__ __launch_bounds__(NTHR, 2) void dw_pw_infer_kernel(DwPwArgs a) {
  constexpr int TH = 8, TW = STRIDE == 1 ? 16 : 8, MT = TH * TW, NRB = MT / 32;
  constexpr int HT = (TH - 1) * STRIDE + 3, WT = (TW - 1) * STRIDE + 3, HP = HT * WT;
  constexpr int COUT = 32 * NCB;
  constexpr int NH = (HP * CQN + NTHR - 1) / NTHR;                   // halo float4 per thread per chunk
  constexpr int NJG = (NTHR / 64) / NRB;                             // groups of column blocks the waves split into
  constexpr int NBW = NCB >= NJG ? NCB / NJG : 1;                    // 32x32 accumulator blocks per (working) wave
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Xs = reinterpret_cast<float*>(smem_raw);                                          // [HP][XLD] activated input, fp32
  unsigned short* Ap = reinterpret_cast<unsigned short*>(smem_raw + (size_t)HP * XLD * 4); // [3][MT][KC] bf16 planes of the A tile
  unsigned short* Wp = Ap + 3 * MT * KC;                                                   // [3][COUT][KC] bf16 planes of the W chunk
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  int t = blockIdx.x;
  const int tx0 = (t % a.tiles_x) * TW; t /= a.tiles_x;
  const int ty0 = (t % a.tiles_y) * TH;
  const int b = t / a.tiles_y;
  const int iy0 = ty0 * STRIDE - 1, ix0 = tx0 * STRIDE - 1;          // input coordinates of halo pixel (0, 0)
  const int cq = tid & (CQN - 1);                                     // this thread's channel quad of the chunk, in every phase
  const int pl = tid / CQN;                                           // ... and its pixel / weight-row lane (0..31)
  const bool deferred = a.isc != nullptr;

  // accumulator blocks of this wave: row block rb, column blocks jb0 .. jb0 + NBW - 1 (waves beyond the block count only stage)
  const int rb = wave % NRB;
  const int jb0 = (wave / NRB) * NBW;
  const bool mma = jb0 < NCB;
  f32x16 acc[NBW];
#pragma unroll
  for (int j = 0; j < NBW; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  // ---- per-chunk global loads, one chunk ahead -------------------------------------------------------------------------------
  float4 xh[NH], wr[NCB], wt[9], isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = kd_zero4(), dsc, dsh;
  auto issue = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      int e = tid + NTHR * i;
      e = e < HP * CQN ? e : HP * CQN - 1;
      const int hp = e / CQN;
      int iy = iy0 + hp / WT, ix = ix0 + hp % WT;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      xh[i] = kd_ld4(a.x + (((int64_t)b * a.H + iy) * a.W + ix) * a.Ch + k0 + 4 * cq);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) wr[i] = kd_ld4(a.wp + (int64_t)(pl + 32 * i) * a.Ch + k0 + 4 * cq);
#pragma unroll
    for (int i = 0; i < 9; ++i) wt[i] = kd_ld4(a.wd + (int64_t)(k0 + 4 * cq) * 9 + 4 * i);     // 4 channels x 9 taps, contiguous
    if (deferred) { isc = kd_ld4(a.isc + k0 + 4 * cq); ish = kd_ld4(a.ish + k0 + 4 * cq); }
    dsc = kd_ld4(a.dsc + k0 + 4 * cq); dsh = kd_ld4(a.dsh + k0 + 4 * cq);
  };

  issue(0);
  for (int k0 = 0; k0 < a.Ch; k0 += KC) {
    // ---- 1. stage the activated halo tile and the split W chunk ------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      const int e = tid + NTHR * i;
      const int hp = (e < HP * CQN ? e : HP * CQN - 1) / CQN;
      const int iy = iy0 + hp / WT, ix = ix0 + hp % WT;
      const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      float4 v = xh[i];
      if (deferred) v = kd_affine_act4(v, isc, ish, a.iact);
      v = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
      if (e < HP * CQN) kd_st4(Xs + hp * XLD + 4 * cq, v);
    }
#pragma unroll
    for (int i = 0; i < NCB; ++i) {
      const int n = pl + 32 * i;
      uint2 hi, mid, lo;
      kd_split3(wr[i], hi, mid, lo);
      unsigned short* d = Wp + n * KC + sw_off(n, cq >> 1) + (cq & 1) * 4;
      *reinterpret_cast<uint2*>(d) = hi;
      *reinterpret_cast<uint2*>(d + COUT * KC) = mid;
      *reinterpret_cast<uint2*>(d + 2 * COUT * KC) = lo;
    }
    // the depthwise taps / coefficients of THIS chunk move to their own registers: the prefetch below overwrites wt, dsc, dsh
    float w9[4][9];
    {
      const float flat[36] = {wt[0].x, wt[0].y, wt[0].z, wt[0].w, wt[1].x, wt[1].y, wt[1].z, wt[1].w, wt[2].x, wt[2].y, wt[2].z, wt[2].w,
                              wt[3].x, wt[3].y, wt[3].z, wt[3].w, wt[4].x, wt[4].y, wt[4].z, wt[4].w, wt[5].x, wt[5].y, wt[5].z, wt[5].w,
                              wt[6].x, wt[6].y, wt[6].z, wt[6].w, wt[7].x, wt[7].y, wt[7].z, wt[7].w, wt[8].x, wt[8].y, wt[8].z, wt[8].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) w9[j][tp] = flat[j * 9 + tp];
    }
    const float4 csc = dsc, csh = dsh;
    kd_lds_barrier();
    if (k0 + KC < a.Ch) issue(k0 + KC);                               // flies during the depthwise phase and the MFMAs

    // ---- 2. depthwise 3x3 on the staged tile -> BatchNorm + activation -> three bf16 planes of the A tile -------------------
#pragma unroll
    for (int p = 0; p < MT / 32; ++p) {
      const int px = p * 32 + pl;
      const int ty = px / TW, tx = px % TW;
      const float* xp = Xs + ((ty * STRIDE) * WT + tx * STRIDE) * XLD + 4 * cq;
      float4 v = kd_zero4();
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float4 l = kd_ld4(xp + (kh * WT + 0) * XLD), c = kd_ld4(xp + (kh * WT + 1) * XLD), rr = kd_ld4(xp + (kh * WT + 2) * XLD);
        v.x = fmaf(l.x, w9[0][kh * 3], fmaf(c.x, w9[0][kh * 3 + 1], fmaf(rr.x, w9[0][kh * 3 + 2], v.x)));
        v.y = fmaf(l.y, w9[1][kh * 3], fmaf(c.y, w9[1][kh * 3 + 1], fmaf(rr.y, w9[1][kh * 3 + 2], v.y)));
        v.z = fmaf(l.z, w9[2][kh * 3], fmaf(c.z, w9[2][kh * 3 + 1], fmaf(rr.z, w9[2][kh * 3 + 2], v.z)));
        v.w = fmaf(l.w, w9[3][kh * 3], fmaf(c.w, w9[3][kh * 3 + 1], fmaf(rr.w, w9[3][kh * 3 + 2], v.w)));
      }
      v = kd_affine_act4(v, csc, csh, a.dact);
      uint2 hi, mid, lo;
      kd_split3(v, hi, mid, lo);
      unsigned short* d = Ap + px * KC + sw_off(px, cq >> 1) + (cq & 1) * 4;
      *reinterpret_cast<uint2*>(d) = hi;
      *reinterpret_cast<uint2*>(d + MT * KC) = mid;
      *reinterpret_cast<uint2*>(d + 2 * MT * KC) = lo;
    }
    kd_lds_barrier();

    // ---- 3. one K-chunk of the GEMM in 16-wide steps -------------------------------------------------------------------------
    if (mma) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        const int ko = sw_off(r, ks * 2 + h);                         // swizzle bits of a block row == those of its lane: block offsets are multiples of 32
        bf16x8 fa[3], fb[NBW][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[p] = *reinterpret_cast<const bf16x8*>(Ap + (p * MT + rb * 32 + r) * KC + ko);
#pragma unroll
          for (int j = 0; j < NBW; ++j) fb[j][p] = *reinterpret_cast<const bf16x8*>(Wp + (p * COUT + (jb0 + j) * 32 + r) * KC + ko);
        }
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};    // smallest terms first (as pw_gemm_kernel)
#pragma unroll
        for (int tt = 0; tt < 6; ++tt)
#pragma unroll
          for (int j = 0; j < NBW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[tt]], fb[j][PB[tt]], acc[j], 0, 0, 0);
      }
    }
    kd_lds_barrier();                                                 // the next chunk overwrites Xs / Wp / Ap
  }

  // ---- epilogue: BatchNorm + activation (+ residual), register q of a block = tile row (q & 3) + 8 (q >> 2) + 4 h ------------
  if (!mma) return;
#pragma unroll
  for (int j = 0; j < NBW; ++j) {
    const int col = (jb0 + j) * 32 + r;
    const float bias = a.pbias ? a.pbias[col] : 0.f, esc = a.psc[col], esh = a.psh[col];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int px = rb * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const int oy = ty0 + px / TW, ox = tx0 + px % TW;
      if (oy < a.Ho && ox < a.Wo) {
        const int64_t m = ((int64_t)b * a.Ho + oy) * a.Wo + ox;
        float v = kd_act(kd_affine(acc[j][q] + bias, esc, esh), a.pact);
        if (a.res) v += a.res[m * a.ldres + col];
        a.out[m * a.ldo + col] = v;
      }
    }
  }
}

template <int STRIDE, int NCB>
int launch(const DwPwArgs& a, hipStream_t st) {
  constexpr int TH = 8, TW = STRIDE == 1 ? 16 : 8, MT = TH * TW;
  constexpr int HP = ((TH - 1) * STRIDE + 3) * ((TW - 1) * STRIDE + 3);
  constexpr size_t lds = (size_t)HP * XLD * 4 + (size_t)3 * MT * KC * 2 + (size_t)3 * NCB * 32 * KC * 2;
  static_assert(lds <= 160 * 1024, "the staged tiles must fit the LDS of one CU");
  static std::atomic<uint64_t> lds_raised{0};
  const hipError_t e = kd_raise_dynamic_lds((const void*)dw_pw_infer_kernel<STRIDE, NCB>, lds, lds_raised);
  KD_REQUIRE(e == hipSuccess, (int)e, "kd_dw_pw_infer: cannot raise the dynamic LDS limit to %zu B: %s", lds, hipGetErrorString(e));
  const int64_t grid = (int64_t)a.B * a.tiles_y * a.tiles_x;
  hipLaunchKernelGGL((dw_pw_infer_kernel<STRIDE, NCB>), dim3((unsigned)grid), dim3(NTHR), lds, st, a);
  return kd_check_launch("kd_dw_pw_infer");
}

}  // namespace

extern "C" {

// 1 if kd_dw_pw_infer has an instance for this shape
int kd_dw_pw_infer_supported(int Ch, int Cout, int stride) {
  if (Ch < 32 || Ch % 32 != 0) return 0;               // hidden channels are walked in chunks of 32
  // (the 128-output-column instances needed 48-52 bytes of scratch per lane and lost to the two separate kernels on every shape they
  // covered -- profiles/r02_dw_pw_fusion.txt: not built since round 3)
  if (stride == 1) return Cout == 32 || Cout == 64;
  if (stride == 2) return Cout == 64;
  return 0;
}

// out[B*Ho*Wo, Cout] = pact(bn_p(conv1x1(dact(bn_d(dwconv3x3_stride(iact(bn_i(x)))))))) (+ res), all BatchNorms as eval-mode
// affine (scale, shift) pairs; split bf16x3 GEMM arithmetic.  isc == NULL: x is used as it is.  See the file header.
int kd_dw_pw_infer(const float* x, const float* isc, const float* ish, int iact, const float* wd, const float* dsc, const float* dsh,
                   int dact, const float* wp, const float* pbias, const float* psc, const float* psh, int pact, const float* res,
                   int64_t ldres, float* out, int64_t ldo, int B, int H, int W, int Ch, int stride, int Cout, void* stream) {
  KD_REQUIRE(x && wd && dsc && dsh && wp && psc && psh && out && B > 0 && H > 0 && W > 0, KD_ERR_ARG, "kd_dw_pw_infer: bad args");
  KD_REQUIRE(!isc || ish, KD_ERR_ARG, "kd_dw_pw_infer: isc needs ish");
  KD_REQUIRE(kd_dw_pw_infer_supported(Ch, Cout, stride), KD_ERR_SHAPE, "kd_dw_pw_infer: no instance for Ch=%d Cout=%d stride=%d", Ch, Cout, stride);
  KD_REQUIRE(ldo >= Cout && (!res || ldres >= Cout), KD_ERR_SHAPE, "kd_dw_pw_infer: row strides smaller than Cout");
  KD_REQUIRE(kd_aligned16(x) && kd_aligned16(wd) && kd_aligned16(wp) && kd_aligned16(dsc) && kd_aligned16(dsh) && kd_aligned16(isc) &&
             kd_aligned16(ish), KD_ERR_ALIGN, "kd_dw_pw_infer: 16-byte alignment");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  KD_REQUIRE((int64_t)B * Ho * Wo < ((int64_t)1 << 31), KD_ERR_SHAPE, "kd_dw_pw_infer: too many output pixels");
  const int TH = 8, TW = stride == 1 ? 16 : 8;
  DwPwArgs a{x, isc, ish, iact, wd, dsc, dsh, dact, wp, pbias, psc, psh, pact, res, ldres, out, ldo, B, H, W, Ch, Ho, Wo,
             (Wo + TW - 1) / TW, (Ho + TH - 1) / TH};
  hipStream_t st = (hipStream_t)stream;
  if (stride == 1) {
    if (Cout == 32) return launch<1, 1>(a, st);
    return launch<1, 2>(a, st);
  }
  return launch<2, 2>(a, st);
}

}  // extern "C"
