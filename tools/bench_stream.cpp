// Dev tool: A/B of the weight-resident streaming GEMM kernels (kd_gemm_stream.hip) against the tiled kernels
// (kd_gemm.hip) through the C ABI: bitwise comparison of the raw outputs, comparison of the BatchNorm-statistics
// sums, HIP-event timings.  Build:
//   hipcc --offload-arch=gfx950 -O2 tools/bench_stream.cpp -I include -L<csrc> -lkd_hip -o tools/bench_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include <dlfcn.h>
#include "kd_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("rc=%d %s (line %d)\n", rc_, kd_last_error_string(), __LINE__); exit(1); } } while (0)

struct Shape { const char* name; long M; int K, N; int kind; };   // kind 0: camera-style (pro1), 1: LiDAR L1 (pro3: A = points)

int main(int argc, char** argv) {
  const long scale = argc > 1 ? atol(argv[1]) : 32;    // frames
  std::vector<Shape> shapes = {
      {"lidar L1 64->128 (pro3)", scale * 80000, 64, 128, 1}, {"lidar L2 128->128", scale * 80000, 128, 128, 0},
      {"lidar L1 64->128 (pro1)", scale * 80000, 64, 128, 0},
      {"stage1 pw 32->32", scale * 16384, 32, 32, 0}, {"stage5-ish 128->64", scale * 4096, 128, 64, 0},
      {"fpn 64->128", scale * 4096, 64, 128, 0}, {"fpn/fusion 128->128", scale * 4096, 128, 128, 0},
      {"head 64->32", scale * 4096, 64, 32, 0}, {"stage5 expand 128->768", scale * 1024, 128, 768, 0},
      {"stage2 project 192->64", scale * 4096, 192, 64, 0}, {"stage3 project 384->64", scale * 4096, 384, 64, 0},
      {"stage4 project 384->128", scale * 1024, 384, 128, 0}, {"head 256->64", scale * 4096, 256, 64, 0},
      {"attention 256->128", scale * 4096, 256, 128, 0}, {"odd M 384->64", 100003, 384, 64, 0},
      {"odd M 128->128", 100003, 128, 128, 0}, {"tiny M 64->128", 77, 64, 128, 0}};
  size_t big = 0;
  for (auto& s : shapes) big = std::max(big, (size_t)s.M * std::max(s.K, s.N));
  float *A, *C, *C2, *W, *vec, *partial, *partial2, *pts, *add;
  CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&C, big * 4)); CK(hipMalloc(&C2, big * 4)); CK(hipMalloc(&add, big * 4));
  CK(hipMalloc(&W, 768 * 768 * 4)); CK(hipMalloc(&vec, 16 * 1024 * 4));
  const size_t prow = (size_t)((scale * 80000 + 127) / 128 + 8);
  CK(hipMalloc(&partial, prow * 2 * 768 * 4)); CK(hipMalloc(&partial2, prow * 2 * 768 * 4));
  CK(hipMalloc(&pts, (size_t)scale * 80000 * 16));
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(add, h.data(), big * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data(), 768 * 768 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vec, h.data() + 12345, 16 * 1024 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(pts, h.data() + 999, (size_t)scale * 80000 * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kd_set_gemm_split(1);
  float *sc = vec, *sh = vec + 1024, *w0 = vec + 2048, *b0 = vec + 4096, *bias = vec + 5120, *esc = vec + 6144, *esh = vec + 7168;
  auto timeit = [&](auto fn) {
    fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
  int bad = 0;
  for (auto& s : shapes) {
    const double fl = 2.0 * s.M * s.K * s.N;
    const double by = 4.0 * ((double)s.M * (s.kind ? 4 : s.K) + (double)s.M * s.N);
    for (int variant = 0; variant < 3; ++variant) {          // 0: epi1 (stats), 1: epi0, 2: epi5 (+ residual)
      if (s.kind == 1 && variant == 2) continue;
      const int epi = variant == 0 ? 1 : (variant == 1 ? 0 : 5);
      auto run = [&](float* out, float* part) {
        if (s.kind == 1)
          RC(kd_lidar_l1_fwd(pts, w0, b0, sc, sh, 1, W, bias, out, s.N, epi, part, kd_pwconv_stat_rows_for(s.M, s.K, s.N, 3, epi, 0), s.M, s.K, s.N, nullptr, nullptr));
        else
          RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, bias, out, s.N, epi == 5 ? add : nullptr, s.N, epi,
                            nullptr, 0, esc, esh, nullptr, nullptr, 1, part, kd_pwconv_stat_rows_for(s.M, s.K, s.N, 1, epi, 0), s.M, s.K, s.N, nullptr, nullptr));
      };
      kd_set_gemm_stream(0);
      const long rows0 = kd_pwconv_stat_rows_for(s.M, s.K, s.N, s.kind ? 3 : 1, epi, 0);
      CK(hipMemset(C2, 0xff, (size_t)s.M * s.N * 4));
      float t_old = timeit([&] { run(C2, partial2); });
      kd_set_gemm_stream(2);
      const long rows1 = kd_pwconv_stat_rows_for(s.M, s.K, s.N, s.kind ? 3 : 1, epi, 0);
      CK(hipMemset(C, 0xee, (size_t)s.M * s.N * 4));
      float t_new = timeit([&] { run(C, partial); });
      if (auto rd = (int (*)(unsigned long long*, int))dlsym(RTLD_DEFAULT, "kd_stream_dbg_read")) {   // dev build with phase stamps
        unsigned long long c[8];
        rd(c, 1); run(C, partial); rd(c, 1);
        const double sl = (double)c[3], nw = (double)c[5];
        if (sl > 0) printf("    phases per slab (cycles): load wait + conv0 %.0f | k-loop %.0f | next loads + epilogue %.0f | slabs/wave %.1f | wave lifetime %.1f us\n",
                           c[0] / sl, c[1] / sl, c[2] / sl, sl / nw, c[4] / nw / 100.0);
      }
      // compare
      const size_t n = (size_t)s.M * s.N;
      std::vector<float> r0(std::min(n, (size_t)1 << 24)), r1(r0.size());
      const size_t off = n > r0.size() ? n - r0.size() : 0;            // the tail (odd M, last slabs)
      CK(hipMemcpy(r0.data(), C2 + off, r0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), C + off, r1.size() * 4, hipMemcpyDeviceToHost));
      size_t ndiff = 0; double md = 0;
      for (size_t i = 0; i < r0.size(); ++i) if (!(r0[i] == r1[i])) { ++ndiff; md = std::max(md, (double)fabsf(r0[i] - r1[i])); }
      double sd = 0;
      if (epi == 1) {
        std::vector<float> p0((size_t)rows0 * 2 * s.N), p1((size_t)rows1 * 2 * s.N);
        CK(hipMemcpy(p0.data(), partial2, p0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), partial, p1.size() * 4, hipMemcpyDeviceToHost));
        for (int st = 0; st < 2; ++st)
          for (int c = 0; c < s.N; ++c) {
            double a = 0, b = 0;
            for (long r = 0; r < rows0; ++r) a += p0[((size_t)r * 2 + st) * s.N + c];
            for (long r = 0; r < rows1; ++r) b += p1[((size_t)r * 2 + st) * s.N + c];
            sd = std::max(sd, fabs(a - b) / std::max(1.0, fabs(a)));
          }
      }
      const bool ok = ndiff == 0 && sd < 1e-5;
      bad += !ok;
      printf("%-26s M=%9ld epi%d | tiled %8.1fus %6.1fTF %5.2fTB/s | stream %8.1fus %6.1fTF %5.2fTB/s | x%.2f | rows %ld -> %ld | diff %zu (max %.2e) stats %.1e %s\n",
             s.name, s.M, epi, t_old * 1e3, fl / t_old / 1e9, by / t_old / 1e9, t_new * 1e3, fl / t_new / 1e9, by / t_new / 1e9, t_old / t_new,
             rows0, rows1, ndiff, md, sd, ok ? "OK" : "MISMATCH");
    }
  }
  // ---- data gradient: dX = (al*(G*mask) + be*Y + ga) . W^T [* act'(X), BatchNorm-backward sums] -- the K <= 128 reductions --------
  struct DShape { const char* name; long M; int K, N; };     // K = reduction width (the layer's output channels), N = its input channels
  std::vector<DShape> dshapes = {{"stage1 project 32<-32", scale * 16384, 32, 32}, {"stage3 project 384<-64", scale * 4096, 64, 384},
                                 {"stage5 project 768<-128", scale * 1024, 128, 768}, {"fpn/fusion 128<-128", scale * 4096, 128, 128},
                                 {"head 128<-64", scale * 4096, 64, 128}, {"odd M 128<-128", 100003, 128, 128},
                                 {"stage2 expand 32<-192", scale * 16384, 192, 32}, {"stage3 expand 64<-384", scale * 4096, 384, 64},
                                 {"odd M 64<-384", 100003, 384, 64}};
  float *al = vec + 8192, *be = vec + 9216, *ga = vec + 10240, *mean = vec + 11264, *inv = vec + 12288;
  for (auto& s : dshapes) {
    const double fl = 2.0 * s.M * s.K * s.N;
    for (int epi = 0; epi <= 2; epi += 2) {
      const double by = 4.0 * ((double)s.M * s.K * 2 + (double)s.M * s.N * (epi == 2 ? 2 : 1));
      auto run = [&](float* out, float* part) {
        RC(kd_pwconv_gemm(A, s.K, add, s.K, 2, 2, al, be, ga, sc, sh, W, nullptr, out, s.N, nullptr, 0, epi, epi == 2 ? add : nullptr, s.N, esc, esh,
                          mean, inv, 2, part, kd_pwconv_stat_rows_for(s.M, s.K, s.N, 2, epi, 0), s.M, s.K, s.N, nullptr, nullptr));
      };
      kd_set_gemm_stream(0);
      const long rows0 = kd_pwconv_stat_rows_for(s.M, s.K, s.N, 2, epi, 0);
      CK(hipMemset(C2, 0xff, (size_t)s.M * s.N * 4));
      float t_old = timeit([&] { run(C2, partial2); });
      kd_set_gemm_stream(2);
      const long rows1 = kd_pwconv_stat_rows_for(s.M, s.K, s.N, 2, epi, 0);
      CK(hipMemset(C, 0xee, (size_t)s.M * s.N * 4));
      float t_new = timeit([&] { run(C, partial); });
      const size_t n = (size_t)s.M * s.N;
      std::vector<float> r0(std::min(n, (size_t)1 << 24)), r1(r0.size());
      const size_t off = n > r0.size() ? n - r0.size() : 0;
      CK(hipMemcpy(r0.data(), C2 + off, r0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), C + off, r1.size() * 4, hipMemcpyDeviceToHost));
      size_t ndiff = 0;
      for (size_t i = 0; i < r0.size(); ++i) ndiff += !(r0[i] == r1[i]);
      bad += ndiff != 0;
      printf("dgrad %-24s M=%9ld epi%d | tiled %8.1fus %6.1fTF %5.2fTB/s | stream %8.1fus %6.1fTF %5.2fTB/s | x%.2f | rows %ld -> %ld | diff %zu %s\n",
             s.name, s.M, epi, t_old * 1e3, fl / t_old / 1e9, by / t_old / 1e9, t_new * 1e3, fl / t_new / 1e9, by / t_new / 1e9, t_old / t_new,
             rows0, rows1, ndiff, ndiff ? "MISMATCH" : (rows0 == rows1 && epi == 2 ? "(tiled in both modes)" : "OK"));
    }
  }
  printf(bad ? "FAILED: %d mismatching cases\n" : "all cases bit-identical\n", bad);
  return bad != 0;
}
