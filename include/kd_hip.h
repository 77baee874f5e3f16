/* kd_hip.h -- C ABI of libkd_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * camera+LiDAR knowledge-distillation training step.
 *
 * The reference (KELVIN-ASU/Lightweight-Multi-Modal-Scene-Understanding-via-Knowledge-Distillation)
 * has no native layer: its hot path is Python nn.Modules dispatching to ATen.  Each entry point
 * below therefore replaces the ATen op sequence issued by the cited reference lines (paths are
 * relative to the reference root).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless stated; the caller (PyTorch's caching allocator)
 *     owns all memory including workspaces; the library never allocates and keeps no pointer;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no host synchronisation,
 *     so every call is legal inside hipStreamBeginCapture / hipGraph replay;
 *   - return 0 on success, < 0 for argument / shape / alignment / workspace errors, > 0 for a
 *     hipError_t; kd_last_error_string() describes the last failure on the calling thread;
 *   - activations are fp32 NHWC: a row-major matrix [M = B*H*W, C] with a row stride `ld` (floats);
 *   - a "deferred" operand (x, sc, sh, act) means value = act(x*sc[c] + sh[c]); sc == NULL means
 *     the tensor is already materialised; act: 0 none, 1 ReLU, 2 ReLU6;
 *   - BatchNorm batch statistics travel as a slab `partial[rows][2][C]` of per-workgroup sums whose
 *     row count is given by the matching kd_*_stat_rows() query (host function, no GPU work);
 *     `pstride` is the slab's channel stride (== C unless the BN covers a slice of a wider slab).
 */
#ifndef KD_HIP_H
#define KD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KD_ACT_NONE 0
#define KD_ACT_RELU 1
#define KD_ACT_RELU6 2

int kd_version(void);
const char* kd_arch(void);
const char* kd_last_error_string(void);

/* ---- 1x1 convolution = fp32-MFMA GEMM ------------------------------------------------------
 * Replaces nn.Conv2d(k=1) / nn.Conv1d(k=1) forward and its data gradient:
 * camera_encoder.py:24,39  fusion_module.py:12,29,116  lidar_encoder.py:29,32.
 *   C[M,N] = Aeff[M,K] . W[N,K]^T (+ bias[N]) (+ addend[M,N])
 *   pro 0: Aeff = A                      pro 1: Aeff = act(A*p0 + p1)         (p0/p1 = sc/sh [K])
 *   pro 2: Aeff = p0*(A*mask(A2*p3+p4)) + p1*A2 + p2   (BN backward folded in: A = G, A2 = X raw,
 *          p0/p1/p2 = al/be/ga from kd_bn_bwd_finalize, p3/p4 = sc/sh of the mask or NULL)
 *   epi 0: store     epi 1: store + partial (sum, sum^2)      [forward, feeds kd_bn_finalize_train]
 *   epi 2: C *= act'(X*esc+esh); partial (sum C, sum C*xhat)  [dgrad, feeds kd_bn_bwd_finalize]
 * The dgrad call passes W = transposed weight [K_out=Cin][N_red=Cout] (kd_transpose).
 * m_dev (optional device int): data-dependent row count <= M read by the kernel itself (no host sync). */
/* Arithmetic of the GEMM family: 0 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = every fp32 operand split
 * exactly into three bf16 pieces and the six leading piece products accumulated in fp32 on the bf16 matrix pipe
 * (error <= 2^-23 |x||y| per product, i.e. fp32-grade).  Process-wide; returns the previous setting. */
int kd_set_gemm_split(int on);
int64_t kd_pwconv_stat_rows(int64_t M);
int64_t kd_pwconv_stat_rows_for(int64_t M, int K, int N, int pro, int epi, int with_addend);
/* Rows of the statistics slab the launch for (K, N, pro, epi, addend given or not) will write in the current arithmetic (one per workgroup for the
 * weight-resident streaming kernels of kd_gemm_stream.hip, one per 128 matrix rows for the tiled kernels): size the slab and
 * drive kd_bn_finalize_train / kd_bn_bwd_finalize with it.  kd_set_gemm_stream: 0 = tiled kernels only, 1 = streaming
 * kernels only for the shapes that win in isolation, 2 = every covered shape (default; env KD_GEMM_STREAM=0|1|all), 3 = every covered forward shape, tiled data gradients;
 * returns the previous mode. */
int kd_set_gemm_stream(int mode);
int kd_pwconv_gemm(const float* A, int64_t lda, const float* A2, int64_t lda2, int pro, int pro_act,
                   const float* p0, const float* p1, const float* p2, const float* p3, const float* p4,
                   const float* W, const float* bias, float* C, int64_t ldc, const float* addend,
                   int64_t ldadd, int epi, const float* X, int64_t ldx, const float* esc, const float* esh,
                   const float* emean, const float* einv, int epi_act, float* partial, int64_t partial_rows, int64_t M,
                   int K, int N, const int* m_dev, void* stream);
/* partial_rows (here and in kd_lidar_l1_fwd / _l1_dgrad / _l2_dgrad): the row count the caller sized `partial` for --
 * the value kd_pwconv_stat_rows_for() / kd_lidar_l*_dgrad_stat_rows() returned -- and will pass to kd_bn_finalize_train /
 * kd_bn_bwd_finalize.  The launch is REFUSED (status < 0) when it differs from what the kernel form selected now would
 * write (the streaming and tiled forms write different row counts and the choice depends on process-wide switches):
 * a slab sized for the other form is an error, never a silently wrong reduction.  Ignored when partial == NULL. */
/* weight gradient dW[N,K] = Deff[M,N]^T . Aeff[M,K] (split-M partial tiles in `ws`, fixed-order sum).
 * d_mode 0: Deff = D; d_mode 2: Deff = al*(D*mask(X*msc+msh)) + be*X + ga.  a_mode 0/1 like pro 0/1. */
size_t kd_pwconv_wgrad_ws_bytes(int64_t M, int N, int K);
/* Weight-gradient kernel form (split arithmetic only).  1 (default): layers where the role-specialised kernel (csrc/kd_wgrad_rs.hip: one
 * workgroup per CU owns a whole block of dW, vector waves load / convert, matrix waves multiply) measured faster use it; 2 (env
 * KD_WGRAD_RS=all): every layer that has an instance; 0 (KD_WGRAD_RS=0): the tiled kernel everywhere.  Returns the previous mode.
 * kd_pwconv_wgrad_ws_bytes answers for the larger of the two forms whatever the mode. */
int kd_set_wgrad_rs(int mode);
int kd_pwconv_wgrad(const float* D, int64_t ldd, const float* X, int64_t ldx, int d_mode, int d_act,
                    const float* al, const float* be, const float* ga, const float* msc, const float* msh,
                    const float* A, int64_t lda, int a_mode, int a_act, const float* asc, const float* ash,
                    float* dW, int64_t M, int N, int K, void* ws, size_t ws_bytes, void* stream);
int kd_transpose(const float* in, float* out, int R, int C, void* stream);
/* the same for n weights in one launch: table = device int64 [n][5] {in pointer, out pointer, R, C, first 256-element block of
 * this matrix}, first blocks ascending from 0, nblocks = sum of ceil(R*C/256) */
int kd_transpose_batch(const int64_t* table, int n, int nblocks, void* stream);
/* up to four small device-to-device float copies in one launch (unused segments: n = 0): packed parameter gradients -> their
 * slots in a flat gradient buffer */
int kd_copy_segments(const float* s0, float* d0, int64_t n0, const float* s1, float* d1, int64_t n1, const float* s2, float* d2,
                     int64_t n2, const float* s3, float* d3, int64_t n3, void* stream);

/* ---- stem 3x3/s2 conv (camera_encoder.py:63-67), NCHW image in, NHWC raw out, Cout == 32 ---- */
int64_t kd_stem_stat_rows(int64_t npix_out);
int kd_stem_conv_fwd(const float* x_nchw, const float* w, float* y_nhwc, float* partial, int B, int Cin, int H,
                     int W, int Cout, void* stream);
/* inference (eval mode, no autograd): y = act(conv(x) * sc + sh) in one kernel, Cin == 3; the bits of kd_stem_conv_fwd + kd_bn_act_apply */
int kd_stem_conv_fwd_infer(const float* x_nchw, const float* w, const float* sc, const float* sh, int act, float* y_nhwc,
                           int B, int Cin, int H, int W, int Cout, void* stream);
/* im2col (K padded to Kp, zeros) so that the stem weight gradient is kd_pwconv_wgrad with K = Kp. */
int kd_stem_im2col(const float* x_nchw, float* col, int B, int Cin, int H, int W, int Kp, void* stream);

/* ---- depthwise 3x3, pad 1, stride 1|2 (camera_encoder.py:30-35, fusion_module.py:25-27,78) ---- */
int64_t kd_dwconv_stat_rows(int64_t npix_out, int C);
int kd_dwconv3x3_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, float* y,
                     float* partial, int B, int H, int W, int C, int stride, void* stream);
int64_t kd_dwconv_bwd_stat_rows(int64_t npix_in, int C);
/* form of the stride-1 backward when both gradients are requested: 0 separate data / weight kernels, 1 one fused
 * column-walk kernel, 2 one fused tile kernel (operands staged once through LDS), 3 (default) chosen by shape */
int kd_set_dw_bwd_mode(int mode);
size_t kd_dwconv_bwd_ws_bytes(int64_t npix_out, int C);
int kd_dwconv3x3_bwd(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                     const float* dsc, const float* dsh, int d_act, const float* x, const float* sc,
                     const float* sh, int act, const float* mean, const float* invstd, const float* w, float* gx,
                     float* partial, float* dw, int B, int H, int W, int C, int stride, void* ws, size_t ws_bytes,
                     void* stream);
/* the same with a second gradient path into the conv's input (the residual of an inverted-residual block whose first
 * convolution is this one, camera_encoder.py:46-51 at expansion_ratio 1): gx = (conv^T dy + addend) * act'(.), sums included.
 * Needs both gradients and a shape kd_dwconv3x3_bwd_add_supported() accepts (the stride-1 column-walk form). */
int kd_dwconv3x3_bwd_add_supported(int C, int W, int stride);
int kd_dwconv3x3_bwd_add(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                         const float* dsc, const float* dsh, int d_act, const float* x, const float* sc,
                         const float* sh, int act, const float* mean, const float* invstd, const float* w,
                         const float* addend, float* gx, float* partial, float* dw, int B, int H, int W, int C, int stride,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- BatchNorm coefficient kernels (nn.BatchNorm1d/2d: eps 1e-5, momentum 0.1) ---------------- */
int kd_bn_finalize_train(float* partial, int rows, int C, int pstride, int64_t count, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                         void* stream);
int kd_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                      float eps, int C, float* mean, float* invstd, float* scale, float* shift, void* stream);
int64_t kd_rowwise_stat_rows(int64_t M, int C);
int kd_bn_act_apply(const float* x, int64_t ldx, const float* sc, const float* sh, int act, const float* res,
                    int64_t ldr, float* out, int64_t ldo, int64_t M, int C, void* stream);
/* the same with a residual that is itself deferred: out = act(x*sc+sh) + ract(res*rsc+rsh) */
int kd_bn_act_apply_res(const float* x, int64_t ldx, const float* sc, const float* sh, int act, const float* res, int64_t ldr,
                        const float* rsc, const float* rsh, int ract, float* out, int64_t ldo, int64_t M, int C, void* stream);
int kd_bn_bwd_reduce(const float* D, int64_t ldd, const float* X, int64_t ldx, const float* sc, const float* sh,
                     int act, const float* mean, const float* invstd, float* partial, int64_t M, int C,
                     void* stream);
int kd_bn_bwd_finalize(float* partial, int rows, int C, int pstride, int64_t count, const float* gamma, const float* mean,
                       const float* invstd, int training, float* dgamma, float* dbeta, float* al, float* be,
                       float* ga, float* dbias, void* stream);

/* ---- LiDAR branch (lidar_encoder.py:25-35,42-99) ----------------------------------------------- */
/* layer 0 (Conv1d 4 -> C, bias): y == NULL runs the BatchNorm-statistics pass only -- the consumers below recompute
 * the layer from the 16-byte point, so its [P, C] output never exists in HBM */
int kd_lidar_l0_fwd(const float* pts, const float* w, const float* b, float* y, float* partial, int64_t P, int C,
                    const int* p_dev, void* stream);
/* layer 1 (Conv1d C0 -> N, lidar_encoder.py:29) over act0(bn0(layer0(pts))) recomputed on load: forward (epi 0 store,
 * 1 store + BN statistics), data gradient (writes G0 = d/d(layer-0 output) * act0' with the BN0-backward sums; Wt = W1
 * transposed to [K0][N1]) and weight gradient.  Same kernels and coefficient conventions as kd_pwconv_gemm / _wgrad. */
int kd_lidar_l1_fwd(const float* pts, const float* w0, const float* b0, const float* sc0, const float* sh0, int act0,
                    const float* W1, const float* bias1, float* C, int64_t ldc, int epi, float* partial,
                    int64_t partial_rows, int64_t M, int K, int N, const int* m_dev, void* stream);
int kd_lidar_l1_dgrad(const float* G, int64_t ldg, const float* Y1, int64_t ldy, const float* al, const float* be,
                      const float* ga, const float* msc, const float* msh, int mact, const float* Wt, float* G0,
                      int64_t ldg0, const float* pts, const float* w0, const float* b0, const float* sc0,
                      const float* sh0, const float* mean0, const float* invstd0, int act0, float* partial,
                      int64_t partial_rows, float* m1_out, void* m1_ws, size_t m1_ws_bytes, int64_t M, int N1, int K0,
                      void* stream);
/* m1_out (optional, [4][K0]) = sum_m G0[m][c] * pts[m][j]: the only way the layer-0 weight gradient depends on G0
 * (dW0 = al0 * m1 + sum_m (be0*y0 + ga0) * pts, the second term from kd_lidar_l0_bwd with D = NULL).  With m1_out set,
 * G0 may be NULL and the [points, K0] gradient is never written.  m1_ws: kd_lidar_l1_dgrad_ws_bytes. */
size_t kd_lidar_l1_dgrad_ws_bytes(int64_t M, int K0);
/* rows of the BatchNorm-backward slab (`partial`) kd_lidar_l1_dgrad / kd_lidar_l2_dgrad write for a shape in the current
 * arithmetic (one per workgroup when a streaming kernel takes the launch, one per 128 matrix rows otherwise) */
int64_t kd_lidar_l1_dgrad_stat_rows(int64_t M, int N1, int K0);
int64_t kd_lidar_l2_dgrad_stat_rows(int64_t M, int N2, int K1);
int kd_lidar_l1_wgrad(const float* D, int64_t ldd, const float* X, int64_t ldx, int d_act, const float* al,
                      const float* be, const float* ga, const float* msc, const float* msh, const float* pts,
                      const float* w0, const float* b0, const float* sc0, const float* sh0, int act0, float* dW,
                      int64_t M, int N, int K, void* ws, size_t ws_bytes, void* stream);
size_t kd_lidar_l0_bwd_ws_bytes(int64_t P, int C);
/* Y == NULL: the layer output is recomputed from (w, b) and the point (it was never written, see kd_lidar_l1_*) */
int kd_lidar_l0_bwd(const float* D, const float* Y, const float* w, const float* b, const float* al, const float* be, const float* ga,
                    const float* pts, float* dwb, int64_t P, int C, void* ws, size_t ws_bytes, void* stream);
/* eval mode: last point-MLP layer + BatchNorm + ReLU + BEV scatter-max in one kernel (zeroes `grid` first); rows are the
 * compacted in-range points, cell_idx[p] their flat (frame, cell) grid row; the layer output is never written */
int kd_lidar_l2_fwd_scatter(const float* A, int64_t lda, const float* sc1, const float* sh1, int act1, const float* W2,
                            const float* bias2, const float* sc2, const float* sh2, int act2, const int* cell_idx,
                            float* grid, int64_t ncells, int64_t M, int K, int N, const int* m_dev, void* stream);
/* The WHOLE eval-mode encoder in one kernel (csrc/kd_lidar_infer.hip): point MLP 4 -> 64 -> 128 -> 128 (Conv1d + eval BatchNorm +
 * ReLU each, lidar_encoder.py:25-35) + scatter-max (:85-96); no activation leaves the CU (layer 1 is computed transposed so
 * that its accumulators are layer 2's operand fragments).  Split arithmetic; zeroes `grid`; sc / sh = kd_bn_eval_coeffs. */
int kd_lidar_mlp_scatter_infer_supported(int C0, int C1, int C2);
int kd_lidar_mlp_scatter_infer(const float* pts, const int* cell, const int* p_dev, const float* w0, const float* b0,
                               const float* sc0, const float* sh0, const float* W1, const float* bias1, const float* sc1,
                               const float* sh1, const float* W2, const float* bias2, const float* sc2, const float* sh2,
                               float* grid, int64_t ncells, int64_t P, int C0, int C1, int C2, void* stream);
/* the same encoder in the bf16-storage inference mode (operands rounded to bf16 once, one MFMA product, fp32 accumulate, fp32
 * grid): replaces kd_bf16_pwconv(a_kind 3) + kd_bf16_pwconv(epi 4) and the bf16 [P, 128] tensor between them */
int kd_bf16_lidar_mlp_scatter_supported(int C0, int C1, int C2);
int kd_bf16_lidar_mlp_scatter(const float* pts, const int* cell, const int* p_dev, const float* w0, const float* b0,
                               const float* sc0, const float* sh0, const float* W1, const float* bias1, const float* sc1,
                               const float* sh1, const float* W2, const float* bias2, const float* sc2, const float* sh2,
                               float* grid, int64_t ncells, int64_t P, int C0, int C1, int C2, void* stream);
int kd_lidar_scatter_max_fwd(const float* pts, const float* y, const float* sc, const float* sh, int act,
                             float* grid, int B, int64_t N, int C, int H, int W, float x0, float x1, float y0,
                             float y1, void* stream);
int64_t kd_lidar_scatter_stat_rows(int64_t P, int C);
size_t kd_lidar_scatter_bwd_ws_bytes(int B, int H, int W, int C);
int kd_lidar_scatter_max_bwd(const float* pts, const float* y, const float* sc, const float* sh, int act,
                             const float* grid, const float* dout, const float* mean, const float* invstd,
                             float* G, float* partial, int B, int64_t N, int C, int H, int W, float x0, float x1,
                             float y0, float y1, void* ws, size_t ws_bytes, void* stream);
int kd_lidar_bev_index(const float* pts, int* cell, int64_t P, int H, int W, float x0, float x1, float y0,
                       float y1, void* stream);
/* training path of the same scatter-max (lidar_encoder.py:57-99), atomic-free: bin the point ids by grid row
 * once (counting sort), then one wave per row takes the max / counts the ties / splits the gradient.
 * row_of_point[B*N] = b*H*W + cell or -1; seg_start[B*H*W + 1]; perm[B*N] (first seg_start[B*H*W] entries used).
 * Same results as kd_lidar_scatter_max_fwd / _bwd (bit-identical grid and G).  C must be 64, 128 or 256. */
size_t kd_lidar_cell_sort_ws_bytes(int B, int64_t N, int H, int W);
int kd_lidar_cell_sort(const float* pts, int B, int64_t N, int H, int W, float x0, float x1, float y0, float y1,
                       int* row_of_point, int* seg_start, int* perm, void* ws, size_t ws_bytes, void* stream);
/* the same bins with a deterministic in-cell order (ascending point id), applied to the POINTS: the point MLP then runs
 * on pts_sorted, every grid row owns a contiguous row range (pass perm = NULL to kd_lidar_seg_max_*), the out-of-range
 * points are the tail, and the eval path's compacted list is the first seg_start[B*H*W] rows.  perm may be NULL.
 * KD_ERR_SHAPE (-4) when H*W + 1 > 36865 (144 KB LDS histogram): fall back to kd_lidar_cell_sort. */
size_t kd_lidar_sort_points_ws_bytes(int B, int64_t N, int H, int W);
int kd_lidar_sort_points(const float* pts, int B, int64_t N, int H, int W, float x0, float x1, float y0, float y1,
                         float* pts_sorted, int* row_sorted, int* seg_start, int* perm, void* ws, size_t ws_bytes,
                         void* stream);
/* Training backward of the last point-MLP layer WITHOUT the [points, C] scatter-max gradient (rows sorted by
 * kd_lidar_sort_points).  kd_lidar_seg_share_bwd leaves share[cells, C] = dout / holders and the BatchNorm-backward sums;
 * the two GEMMs rebuild G[m][c] = (rows[m] >= 0 && v > 0 && v == grid[rows[m]][c]) ? share[rows[m]][c] : 0 on load
 * (v = act2(Y2*sc2 + sh2)) -- bit-identical to kd_lidar_seg_max_bwd + kd_pwconv_gemm(pro 2, epi 2) / kd_pwconv_wgrad
 * (d_mode 2), minus 2.5 passes over a [points, C] tensor. */
int64_t kd_lidar_seg_share_stat_rows(int64_t ncells, int64_t P);
int kd_lidar_seg_share_bwd(const float* y, const float* sc, const float* sh, int act, const float* grid,
                           const float* dout, const float* mean, const float* invstd, const int* seg_start,
                           const int* row_sorted, float* share, float* cnt_ws, float* partial, int64_t P,
                           int64_t ncells, int C, void* stream);
int kd_lidar_l2_dgrad(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share,
                      const float* al, const float* be, const float* ga, const float* sc2, const float* sh2, int act2,
                      const float* Wt, float* G1, int64_t ldg1, const float* Y1, int64_t ldy1, const float* sc1,
                      const float* sh1, const float* mean1, const float* invstd1, int act1, float* partial,
                      int64_t partial_rows, int64_t M, int N2, int K1, void* stream);
int kd_lidar_l2_wgrad(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share,
                      const float* al, const float* be, const float* ga, const float* sc2, const float* sh2, int act2,
                      const float* Y1, int64_t ldy1, const float* sc1, const float* sh1, int act1, float* dW,
                      int64_t M, int N2, int K1, void* ws, size_t ws_bytes, void* stream);
/* The same backward in ONE kernel (csrc/kd_lidar_bwd.hip): G1 + BatchNorm-1 backward sums (== kd_lidar_l2_dgrad, bit for bit)
 * and dW2 (== kd_lidar_l2_wgrad up to summation order) from one read and one bf16x3 split of Y2 and Y1 -- 31 GB instead of
 * 52 GB of HBM traffic at 256 frames x 80 000 points.  Split arithmetic, N2 = K1 = 128 (kd_lidar_l2_bwd_supported); `partial`
 * has kd_lidar_l2_bwd_stat_rows(M) rows (passed back as partial_rows and checked), ws >= kd_lidar_l2_bwd_ws_bytes. */
int kd_lidar_l2_bwd_supported(int N2, int K1);
int64_t kd_lidar_l2_bwd_stat_rows(int64_t M);
size_t kd_lidar_l2_bwd_ws_bytes(int64_t M, int N2, int K1);
int kd_lidar_l2_bwd(const float* Y2, int64_t ldy2, const int* rows, const float* grid, const float* share, const float* al,
                    const float* be, const float* ga, const float* sc2, const float* sh2, int act2, const float* Wt,
                    float* G1, int64_t ldg1, const float* Y1, int64_t ldy1, const float* sc1, const float* sh1,
                    const float* mean1, const float* invstd1, int act1, float* partial, int64_t partial_rows, float* dW,
                    int64_t M, int N2, int K1, void* ws, size_t ws_bytes, void* stream);
/* Layer 1 in the same one-kernel form: dW1, the BatchNorm-0 backward sums of G0 = (dy1 . W1) * relu'(bn0(l0(point))) and the
 * moments m1_out[4][K0] = sum_m G0 * point (== kd_lidar_l1_dgrad(G0 = NULL, m1_out) + kd_lidar_l1_wgrad with an unmasked G:
 * dy1 = al*G + be*Y1 + ga).  Split arithmetic, N1 = 128, K0 = 64, ReLU; G and Y1 dense. */
int kd_lidar_l1_bwd_supported(int N1, int K0);
int64_t kd_lidar_l1_bwd_stat_rows(int64_t M);
size_t kd_lidar_l1_bwd_ws_bytes(int64_t M, int N1, int K0);
int kd_lidar_l1_bwd(const float* G, int64_t ldg, const float* Y1, int64_t ldy, const float* al, const float* be, const float* ga,
                    const float* Wt, const float* pts, const float* w0, const float* b0, const float* sc0, const float* sh0,
                    const float* mean0, const float* invstd0, int act0, float* partial, int64_t partial_rows, float* m1_out,
                    float* dW, int64_t M, int N1, int K0, void* ws, size_t ws_bytes, void* stream);
int kd_lidar_gather_sorted(const float* pts, const int* perm, const int* row_of_point, const int* nvalid_dev,
                           float* out_pts, int* out_row, int64_t P, void* stream);
/* row_sorted (optional, rows sorted by kd_lidar_sort_points, perm == NULL): grid rows holding more than 256 points are
 * then processed 64 points per wave, so a scene concentrated in a few cells costs what a uniform one costs. */
int kd_lidar_seg_max_fwd(const float* y, const float* sc, const float* sh, int act, const int* seg_start,
                         const int* perm, const int* row_sorted, float* grid, int64_t P, int64_t ncells, int C,
                         void* stream);
int64_t kd_lidar_seg_stat_rows(int64_t ncells);
int kd_lidar_seg_max_bwd(const float* y, const float* sc, const float* sh, int act, const float* grid,
                         const float* dout, const float* mean, const float* invstd, const int* seg_start,
                         const int* perm, const int* row_of_point, float* G, float* partial, int64_t P,
                         int64_t ncells, int C, void* stream);
/* inference-only: compact the in-range points (arbitrary order) with their flat (batch, cell) row; `counter`
 * (one device int, zeroed here) receives the count.  Then scatter-max over the pre-binned rows. */
int kd_lidar_compact(const float* pts, float* out_pts, int* out_cell, int* counter, int B, int64_t N, int H, int W,
                     float x0, float x1, float y0, float y1, void* stream);
int kd_lidar_scatter_max_idx_fwd(const float* y, const float* sc, const float* sh, int act, const int* cell_idx,
                                 float* grid, int64_t P, int C, int64_t ncells, const int* p_dev, void* stream);

/* ---- FPN resize, weighted-fusion tail, classifier (fusion_module.py:58-63,115-120,170-173) ----- */
int kd_bilinear_accum_fwd(const float* in, const float* sc, const float* sh, int act, float* out, int accumulate,
                          int B, int Hi, int Wi, int Ho, int Wo, int C, void* stream);
/* the whole FPN sum in one pass: up to three deferred laterals (in_i == NULL: absent, no gaps), bit-identical to
 * accumulating them one by one with kd_bilinear_accum_fwd */
int kd_bilinear_sum_fwd(const float* in0, const float* sc0, const float* sh0, int act0, int H0, int W0, const float* in1,
                        const float* sc1, const float* sh1, int act1, int H1, int W1, const float* in2, const float* sc2,
                        const float* sh2, int act2, int H2, int W2, float* out, int B, int Ho, int Wo, int C, void* stream);
int kd_bilinear_bwd(const float* dout, const float* in, const float* sc, const float* sh, int act,
                    const float* mean, const float* invstd, float* gin, float* partial, int B, int Hi, int Wi,
                    int Ho, int Wo, int C, void* stream);
int kd_weighted_fuse_fwd(const float* cat, const float* sc, const float* sh, const float* hraw, const float* w2,
                         const float* b2, float* out, float* wts, int64_t M, int C, void* stream);
size_t kd_weighted_fuse_bwd_ws_bytes(int64_t M, int C);
int kd_weighted_fuse_bwd(const float* dout, const float* cat, const float* sc, const float* sh, const float* hraw,
                         const float* w2, const float* wts, float* dcat, float* gh, float* dparams, int64_t M,
                         int C, void* ws, size_t ws_bytes, void* stream);
int kd_cls_conv_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, const float* b,
                    float* logits_nchw, int64_t M, int HW, int Cin, int NC, void* stream);
size_t kd_cls_conv_bwd_ws_bytes(int64_t M, int Cin, int NC);
int64_t kd_cls_conv_bwd_stat_rows(int64_t M, int Cin);
int kd_cls_conv_bwd(const float* dlog_nchw, const float* x, const float* sc, const float* sh, int act,
                    const float* mean, const float* invstd, const float* w, float* gx, float* partial, float* dwb,
                    int64_t M, int HW, int Cin, int NC, void* ws, size_t ws_bytes, void* stream);

/* ---- "x4" decoder head: LightweightSegmentationHead (fusion_module.py:142-159) ------------------
 * ConvTranspose2d(k=4, s=2, p=1, bias=False) = kd_pwconv_gemm with W.view(Cin, Cout*16)^T (columns ordered
 * co*16 + kh*4 + kw, the weight's own layout) followed by col2im; backward = im2col of the folded dy
 * (al/be/ga of kd_bn_bwd_finalize, mask of the stage's own BN+ReLU) followed by the wgrad / dgrad GEMMs.
 * cls3x3 = Conv2d(Cin<=32, NC<=4, 3, padding=1) over a deferred NHWC input, NCHW logits. */
int64_t kd_deconv_stat_rows(int64_t npix_out, int Cout);
int kd_deconv4x4s2_col2im_fwd(const float* col, float* out, float* partial, int B, int H, int W, int Cout,
                             void* stream);
int kd_deconv4x4s2_im2col_bwd(const float* D, const float* Y, const float* al, const float* be, const float* ga,
                             const float* msc, const float* msh, int act, float* dcol, int B, int H, int W,
                             int Cout, void* stream);
int kd_cls3x3_fwd(const float* x, const float* sc, const float* sh, int act, const float* w, const float* b,
                  float* logits_nchw, int B, int H, int W, int Cin, int NC, void* stream);
int64_t kd_cls3x3_bwd_stat_rows(int64_t npix, int Cin);
size_t kd_cls3x3_bwd_ws_bytes(int64_t npix, int Cin, int NC);
int kd_cls3x3_bwd(const float* dlog_nchw, const float* x, const float* sc, const float* sh, int act,
                  const float* mean, const float* invstd, const float* w, float* gx, float* partial, float* dwb,
                  int B, int H, int W, int Cin, int NC, void* ws, size_t ws_bytes, void* stream);

/* ---- per-frame input preparation (src/data_loading/pandaset_dataset.py:13-45,108-127) ------------------
 * kd_bev_rasterize = remap_semantic (remap != 0: label = bit `id` of remap_bits, ids outside 0..63 -> 0) +
 * rasterize_bev over B ragged frames (offsets: device int64 [B+1], points of frame b are [offsets[b], offsets[b+1])):
 * mask[b,r,c] = label of the first point in input order of that cell with a non-zero label, else 0.  Range test is
 * inclusive (x_min <= x <= x_max); cell = trunc((v - min) / span * (n-1)) in float32, span = max - min as the
 * reference's float32 arithmetic sees it.  ws: kd_bev_rasterize_ws_bytes(B, H, W). */
size_t kd_bev_rasterize_ws_bytes(int B, int H, int W);
int kd_bev_rasterize(const float* x, const float* y, const int64_t* cls, const int64_t* offsets, int B, int64_t n_total,
                     int remap, uint64_t remap_bits, int H, int W, float x_min, float x_span, float x_max, float y_min,
                     float y_span, float y_max, void* ws, size_t ws_bytes, int64_t* mask, void* stream);
int kd_semantic_remap(const int64_t* raw, int64_t n, uint64_t remap_bits, int64_t* out, void* stream);
/* out[max_points,4] = stack(x,y,z,i) zero-padded, or (choice != NULL, n >= max_points) the rows choice[0..max_points) */
int kd_points_prepare(const float* x, const float* y, const float* z, const float* intensity, const int64_t* choice,
                      int64_t n, int64_t max_points, float* out, void* stream);
int kd_image_u8hwc_to_f32chw(const uint8_t* in, float* out, int H, int W, void* stream);

/* ---- losses, metric, optimiser (trainer.py:18-37,55-56,86-90; KD terms are build-defined) ------ */
size_t kd_seg_loss_ws_bytes(int64_t npix);
int kd_seg_loss_fwd_bwd(const float* zs, const float* zt, const int64_t* target, const float* class_w,
                        int ignore_index, float T, float alpha, float gscale, const float* gscale_dev, float* losses,
                        float* dzs, int B, int NC, int HW, void* ws, size_t ws_bytes, void* stream);
size_t kd_mse_ws_bytes(int64_t n);
int kd_mse_fwd_bwd(const float* a, const float* b, int64_t n, float gcoef, const float* gscale_dev, float* loss,
                   float* da, void* ws, size_t ws_bytes, void* stream);
/* value of the KD objective from its parts: total = ce_kl[0] + ckl * ce_kl[1] + beta * (mse_c[0] + mse_l[0]) (either MSE may be
 * NULL = 0), every operation rounded to fp32 in that order */
int kd_kd_total(const float* ce_kl, const float* mse_c, const float* mse_l, float ckl, float beta, float* total, void* stream);
/* the same objective with fewer launches: kd_mse_partial leaves the per-block sums of (a-b)^2 in a slab of kd_mse_slab_blocks(n)
 * floats (and writes da = gcoef * (a-b) when da != NULL); kd_kd_objective_final reduces two such slabs to out[0], out[1] = the two
 * feature MSEs and out[2] = the total (rounding order of kd_kd_total) */
int64_t kd_mse_slab_blocks(int64_t n);
int kd_mse_partial(const float* a, const float* b, int64_t n, float gcoef, float* da, float* slab, void* stream);
int kd_kd_objective_final(const float* ce_kl, const float* slab_c, int64_t n_c, const float* slab_l, int64_t n_l, float ckl, float beta,
                          float* out, void* stream);
int kd_argmax_confusion(const float* logits, const int64_t* target, int ignore_index, uint64_t* conf,
                        int64_t* pred, int B, int NC, int HW, void* stream);
int kd_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int step, float ginv, void* stream);
/* hipGraph-replay-safe form: state = device float[4] {lr, step, 1-beta1^step, sqrt(1-beta2^step)}; the call
 * advances state[1] on the device and refreshes the bias corrections before the update. */
int kd_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float* state, float beta1,
                      float beta2, float eps, float weight_decay, float ginv, void* stream);

/* ---- inference-mode block fusion (csrc/kd_block.hip) ------------------------------------------------------------------
 * The tail of an InvertedResidual (reference camera_encoder.py:30-42: depthwise 3x3 + BN + ReLU6, project 1x1 + BN, + x) or a
 * whole DWSeparableConv (fusion_module.py:25-34) in ONE kernel for the eval-mode / no-grad forward: every BatchNorm is an
 * (scale, shift) pair known before the launch, so the depthwise output -- the widest tensor of the block -- never goes to HBM.
 *   out[B*Ho*Wo, Cout] = pact(bn_p(conv1x1(dact(bn_d(dwconv3x3_stride(iact(bn_i(x)))))))) (+ res)
 * x: [B,H,W,Ch] NHWC (isc == NULL: used as it is), wd: [Ch][9], wp: [Cout][Ch]; split bf16x3 GEMM arithmetic; bit-identical to
 * kd_dwconv3x3_fwd followed by kd_pwconv_gemm(pro 1, epi 5).  Shapes: Ch a multiple of 32; stride 1 with Cout 32 / 64 / 128,
 * stride 2 with Cout 64 / 128 (kd_dw_pw_infer_supported); anything else returns KD_ERR_SHAPE. */
int kd_dw_pw_infer_supported(int Ch, int Cout, int stride);
int kd_dw_pw_infer(const float* x, const float* isc, const float* ish, int iact, const float* wd, const float* dsc, const float* dsh,
                   int dact, const float* wp, const float* pbias, const float* psc, const float* psh, int pact, const float* res,
                   int64_t ldres, float* out, int64_t ldo, int B, int H, int W, int Ch, int stride, int Cout, void* stream);

/* ---- bf16-storage INFERENCE path (csrc/kd_bf16.hip; BASELINE.json configs[1]) -----------------------------------------
 * A second mode beside the fp32 contract: eval forward only.  Activations are bf16 NHWC matrices [M][C], already
 * normalised + activated; every entry point is one whole unit conv -> fma(raw, sc, sh) -> act (+ residual) -> bf16, with
 * fp32 accumulation; weights / coefficient vectors are fp32.  Replaces, in eval mode: the stem (camera_encoder.py:63-67),
 * depthwise 3x3 (:30-35, fusion_module.py:25-27,78), every 1x1 conv / Conv1d(k=1) (camera_encoder.py:24,39,
 * fusion_module.py:12,29, lidar_encoder.py:26-34 incl. the scatter-max of :85-96), the FPN resize + sum (:58-63) and the
 * classifier (:170).  kd_bf16_pwconv: a_kind 0 = A bf16, 1 = A fp32 (rounded on load), 3 = A are LiDAR points [M][4] and
 * layer 0 (l0w [K][4], l0b, sc0, sh0, act0) is recomputed; epi 0 = C bf16 (+ res bf16), 4 = scatter-max of the
 * non-negative result into the zero-filled fp32 grid [cells][ldgrid] by cell[m] (< 0: skipped). */
int kd_bf16_stem(const float* x_nchw, const float* w, const float* sc, const float* sh, int act, void* y, int B, int Cin, int H,
                 int W, int Cout, void* stream);
int kd_bf16_dwconv3x3(const void* x, const float* w, const float* sc, const float* sh, int act, void* y, int B, int H, int W,
                      int C, int stride, void* stream);
int kd_bf16_pwconv(const void* A, int64_t lda, int a_kind, const float* W, const float* bias, const float* esc, const float* esh,
                   int act, void* C, int64_t ldc, const void* res, int64_t ldres, int epi, int64_t M, int K, int N,
                   const int* m_dev, const float* l0w, const float* l0b, const float* sc0, const float* sh0, int act0,
                   const int* cell, float* grid, int64_t ldgrid, void* stream);
int kd_bf16_bilinear_sum(const void* in0, int H0, int W0, const void* in1, int H1, int W1, const void* in2, int H2, int W2,
                         void* out, int B, int Ho, int Wo, int C, void* stream);
int kd_bf16_cls_conv(const void* x, const float* w, const float* b, float* logits_nchw, int64_t M, int HW, int Cin, int NC,
                     void* stream);
/* weighted-fusion tail (fusion_module.py:115-120): h [M][C] bf16 = relu(attention.0(cat)); out [M][C] bf16 =
 * w_0 * cat[:, :C] + w_1 * cat[:, C:], (w_0, w_1) = softmax(h . w2^T + b2); w2 [2][C], b2 [2] fp32. */
int kd_bf16_weighted_tail(const void* h, const void* cat, const float* w2, const float* b2, void* out, int64_t M, int C,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KD_HIP_H */
