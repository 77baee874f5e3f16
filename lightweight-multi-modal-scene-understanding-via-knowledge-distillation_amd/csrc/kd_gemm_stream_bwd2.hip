// kd_gemm_stream_bwd2.hip -- streaming-GEMM instances for the data gradient THROUGH an activation + BatchNorm (EPI2:
// times act'(X), BatchNorm-backward sums; EPI3: X = LiDAR layer 0 recomputed from the point, plus the moments of the
// layer-0 weight gradient), with the PRO2 operand or the scatter-max gradient rebuilt from the per-cell tables (PRO4):
// kd_gemm_stream_kernel.h.
#include "kd_gemm_stream_kernel.h"

using namespace kd_stream;

int kd_stream_bwd2_dispatch(const GemmArgs& g, int kb, int nb, int pro, int epi, dim3 grid, hipStream_t st) {
  if (pro == 4 && epi == 2 && kb == 4 && nb == 4) { stream_launch_one<4, 1, 4, 4, 2>(g, grid, st); return 1; }
  if (pro == 2 && epi == 3 && kb == 4 && nb == 2) { stream_launch_one<4, 1, 2, 2, 3>(g, grid, st); return 1; }
  if (pro != 2 || epi != 2) return 0;
#define KD_B(KB_, KC_, NB_) if (kb == KB_ && nb == NB_) { stream_launch_one<KB_, KC_, NB_, 2, 2>(g, grid, st); return 1; }
  KD_B(1, 1, 1) KD_B(1, 1, 2) KD_B(1, 1, 4)
  KD_B(2, 1, 1) KD_B(2, 2, 2) KD_B(2, 1, 4)
  KD_B(4, 1, 1) KD_B(4, 2, 2) KD_B(4, 1, 4)
  KD_B(6, 1, 1) KD_B(6, 2, 2) KD_B(6, 1, 4)
  KD_B(8, 1, 1) KD_B(8, 2, 2)
  KD_B(12, 1, 1) KD_B(12, 2, 2)
  KD_B(24, 1, 1)
#undef KD_B
  return 0;
}
