// kd_gemm_stream_bwd0.hip -- streaming-GEMM instances for the data gradient without an activation to differentiate
// behind it (PRO2 operand al*(D*mask) + be*X + ga, plain store (+ residual gradient)): kd_gemm_stream_kernel.h.
#include "kd_gemm_stream_kernel.h"

using namespace kd_stream;

int kd_stream_bwd0_dispatch(const GemmArgs& g, int kb, int nb, dim3 grid, hipStream_t st) {
  const bool add = g.addend != nullptr;
#define KD_B(KB_, KC_, NB_, DB_)                                                           \
  if (kb == KB_ && nb == NB_) {                                                            \
    return add ? stream_launch_one<KB_, KC_, NB_, 2, 0, DB_, true>(g, grid, st)            \
               : stream_launch_one<KB_, KC_, NB_, 2, 0, DB_, false>(g, grid, st);          \
  }
  KD_B(1, 1, 1, false) KD_B(1, 1, 2, false) KD_B(1, 1, 4, false)
  KD_B(2, 1, 1, true) KD_B(2, 1, 2, true) KD_B(2, 1, 4, true)
  KD_B(4, 1, 1, true) KD_B(4, 1, 2, true) KD_B(4, 1, 4, true)
#undef KD_B
  return 0;
}
