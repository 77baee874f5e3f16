// kd_gemm_stream_bwd2.hip -- streaming-GEMM instances for the data gradient THROUGH an activation + BatchNorm (EPI2:
// times act'(X), BatchNorm-backward sums), with the PRO2 operand or the scatter-max gradient rebuilt from the per-cell
// tables (PRO4): kd_gemm_stream_kernel.h.
#include "kd_gemm_stream_kernel.h"

using namespace kd_stream;

int kd_stream_bwd2_dispatch(const GemmArgs& g, int kb, int nb, int pro, int epi, dim3 grid, hipStream_t st) {
  // LiDAR layer 2 (three streamed tensors): one register set -- two spill and run 8.1 instead of 7.3 ms
  if (pro == 4 && epi == 2 && kb == 4 && nb == 4) return stream_launch_one<4, 1, 4, 4, 2, false, false>(g, grid, st);
  if (pro != 2 || epi != 2) return 0;
  const bool add = g.addend != nullptr;
  // Only the (K, column tile, residual) combinations that fit 256 registers WITHOUT scratch are built; stream_cfg
  // (kd_gemm_stream.hip) never promises the others -- the tiled kernel takes those launches.
#define KD_B(KB_, KC_, NB_, DB_, ADD_)                                                     \
  if (kb == KB_ && nb == NB_ && add == ADD_) return stream_launch_one<KB_, KC_, NB_, 2, 2, DB_, ADD_>(g, grid, st);
  KD_B(1, 1, 1, false, false) KD_B(1, 1, 1, false, true) KD_B(1, 1, 2, false, true) KD_B(1, 1, 4, false, false)
  KD_B(2, 1, 1, true, false) KD_B(2, 1, 2, true, true) KD_B(2, 1, 4, true, false)
  KD_B(4, 1, 1, true, false) KD_B(4, 1, 2, true, true) KD_B(4, 1, 4, true, false)
#undef KD_B
  return 0;
}
