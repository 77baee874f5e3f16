// kd_gemm_stream_wide.hip -- the one K >= 192 reduction that the streaming form wins (round 4): the stage-2 expand convolution's
// data gradient, K = 192 -> N = 32 over 4.2 M rows (camera_encoder.py:22-27 backward), in the two-register-set form.
// Measured at 256 frames (profiles/r04_stream_wide_k.txt, tools/bench_stream): 1444 -> 1314 us (epi 0), 1606 -> 1441 us (epi 2).  The
// other K = 192 / 256 / 384 shapes whose weight planes fit the LDS were built and measured in the same run and LOSE to the tiled
// kernel -- forward 384 -> 64: 426 -> 466 us, 384 -> 128 (two column tiles): 151 -> 172, 256 -> 64: 280 -> 305, 256 -> 128: 469 -> 595;
// data gradient 384 -> 64: 694 -> 751 -- and are not instantiated.
#include "kd_gemm_stream_kernel.h"

using namespace kd_stream;

// -> 1 launched, 0 no such instance, < 0 error
int kd_stream_wide_dispatch(const GemmArgs& g, int kb, int nb, int pro, int epi, dim3 grid, hipStream_t st) {
  const bool add = g.addend != nullptr;
  if (kb == 6 && nb == 1 && pro == 2) {
    if (epi == 0) return add ? stream_launch_one<6, 1, 1, 2, 0, true, true>(g, grid, st) : stream_launch_one<6, 1, 1, 2, 0, true, false>(g, grid, st);
    if (epi == 2 && !add) return stream_launch_one<6, 1, 1, 2, 2, true, false>(g, grid, st);     // (with the residual addend: 41 spilled registers)
  }
  return 0;
}
