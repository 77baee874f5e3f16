// kd_gemm_stream_bwd0.hip -- streaming-GEMM instances for the data gradient without an activation to differentiate
// behind it (PRO2 operand al*(D*mask) + be*X + ga, plain store (+ residual gradient)): kd_gemm_stream_kernel.h.
#include "kd_gemm_stream_kernel.h"

using namespace kd_stream;

int kd_stream_bwd0_dispatch(const GemmArgs& g, int kb, int nb, dim3 grid, hipStream_t st) {
#define KD_B(KB_, KC_, NB_) if (kb == KB_ && nb == NB_) { stream_launch_one<KB_, KC_, NB_, 2, 0>(g, grid, st); return 1; }
  KD_B(1, 1, 1) KD_B(1, 1, 2) KD_B(1, 1, 4)
  KD_B(2, 2, 1) KD_B(2, 2, 2) KD_B(2, 2, 4)
  KD_B(4, 2, 1) KD_B(4, 2, 2) KD_B(4, 2, 4)
  KD_B(6, 2, 1) KD_B(6, 2, 2) KD_B(6, 2, 4)
  KD_B(8, 2, 1) KD_B(8, 2, 2)
  KD_B(12, 2, 1) KD_B(12, 2, 2)
  KD_B(24, 2, 1)
#undef KD_B
  return 0;
}
