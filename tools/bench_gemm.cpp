// Dev tool (not part of the product): times kd_pwconv_gemm / kd_pwconv_wgrad on the hot shapes of the
// KD step with HIP events and prints TFLOP/s and algorithmic TB/s next to a plain copy of the same
// bytes.  Build:
//   hipcc --offload-arch=gfx950 -O2 tools/bench_gemm.cpp -I include -L<csrc> -lkd_hip -o tools/bench_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <dlfcn.h>
#include <cmath>
#include <vector>
#include "kd_hip.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("rc=%d %s (line %d)\n", rc_, kd_last_error_string(), __LINE__); exit(1); } } while (0)

__global__ void copy_kernel(const float4* a, float4* c, size_t na, size_t nc) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t k = i; k < na; k += st) { float4 v = a[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  for (size_t k = i; k < nc; k += st) c[k] = acc;
}

struct Shape { const char* name; long M; int K, N; };

int main(int argc, char** argv) {
  std::vector<Shape> shapes = {
      {"lidar L1 64->128", 2560000, 64, 128}, {"lidar L2 128->128", 2560000, 128, 128},
      {"stage2 expand 32->192", 524288, 32, 192}, {"stage2 project 192->64", 131072, 192, 64},
      {"stage3 expand 64->384", 131072, 64, 384}, {"stage3 project 384->64", 131072, 384, 64},
      {"stage5 expand 128->768", 32768, 128, 768}, {"stage5 project 768->128", 32768, 768, 128},
      {"fpn/fusion 128->128", 131072, 128, 128}, {"concat fuse 256->256", 131072, 256, 256}};
  size_t big = 0;
  for (auto& s : shapes) big = std::max(big, (size_t)s.M * std::max(s.K, s.N));
  float *A, *A2, *C, *X, *W, *vec, *partial, *ws;
  CK(hipMalloc(&A, big * 4)); CK(hipMalloc(&A2, big * 4)); CK(hipMalloc(&C, big * 4)); CK(hipMalloc(&X, big * 4));
  CK(hipMalloc(&W, 768 * 768 * 4)); CK(hipMalloc(&vec, 16 * 1024 * 4)); CK(hipMalloc(&partial, (size_t)20000 * 2 * 768 * 4));
  size_t wsb = 512ull << 20; CK(hipMalloc(&ws, wsb));
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(A2, h.data(), big * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(X, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), 768 * 768 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vec, h.data(), 16 * 1024 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = argc > 1 ? atoi(argv[1]) : 5;
  const int split = argc > 2 ? atoi(argv[2]) : 0;       // 1: bf16x6 split-MFMA arithmetic (also prints its deviation from the fp32-MFMA result)
  kd_set_gemm_split(split);
  float* C2 = nullptr;
  if (split) CK(hipMalloc(&C2, big * 4));
  printf("gemm arithmetic: %s\n", split ? "bf16x6 split products (v_mfma_f32_32x32x16_bf16)" : "fp32 (v_mfma_f32_32x32x2_f32)");
  printf("%-25s %8s | %-27s | %-17s | %-27s | %-27s | %-18s\n", "shape", "M", "fwd pro1 epi1", "fwd pro1 epi0", "dgrad pro2 epi2", "wgrad d2 a1", "copy same bytes");
  for (auto& s : shapes) {
    const double fl = 2.0 * s.M * s.K * s.N;
    const double by_f = 4.0 * ((double)s.M * s.K + (double)s.M * s.N);
    const double by_d = 4.0 * ((double)s.M * s.N * 2 + (double)s.M * s.K * 2);   // dgrad: G, Y in; Gout out, X(epi) in
    const double by_w = 4.0 * ((double)s.M * s.N * 2 + (double)s.M * s.K);
    float *sc = vec, *sh = vec + 1024, *al = vec + 2048, *be = vec + 3072, *ga = vec + 4096, *mean = vec + 5120, *inv = vec + 6144;
    auto timeit = [&](auto fn) {
      fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) fn();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps; };
    float t1 = timeit([&] { RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, nullptr, C, s.N, nullptr, 0, 1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, partial, kd_pwconv_stat_rows_for(s.M, s.K, s.N, 1, 1, 0), s.M, s.K, s.N, nullptr, nullptr)); });
    float t0 = timeit([&] { RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, nullptr, C, s.N, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s.M, s.K, s.N, nullptr, nullptr)); });
    // dgrad: A = G [M,N], A2 = Y [M,N], W^T stored [K][N], out [M,K], X(epi) [M,K]
    float t2 = timeit([&] { RC(kd_pwconv_gemm(A, s.N, A2, s.N, 2, 0, al, be, ga, nullptr, nullptr, W, nullptr, C, s.K, nullptr, 0, 2, X, s.K, sc, sh, mean, inv, 2, partial, kd_pwconv_stat_rows_for(s.M, s.N, s.K, 2, 2, 0), s.M, s.N, s.K, nullptr, nullptr)); });
    float t3 = timeit([&] { RC(kd_pwconv_wgrad(A, s.N, A2, s.N, 2, 0, al, be, ga, nullptr, nullptr, X, s.K, 1, 2, sc, sh, C, s.M, s.N, s.K, ws, wsb, nullptr)); });
    if (split) {   // same forward with both arithmetics: max |diff| relative to max |value|
      const size_t n = std::min((size_t)s.M, (size_t)65536) * s.N;
      std::vector<float> r0(n), r1(n);
      RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, nullptr, C, s.N, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s.M, s.K, s.N, nullptr, nullptr));
      kd_set_gemm_split(0);
      RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, nullptr, C2, s.N, nullptr, 0, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, s.M, s.K, s.N, nullptr, nullptr));
      kd_set_gemm_split(1);
      CK(hipMemcpy(r1.data(), C, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r0.data(), C2, n * 4, hipMemcpyDeviceToHost));
      double md = 0, mv = 0;
      for (size_t i = 0; i < n; ++i) { md = std::max(md, (double)fabsf(r1[i] - r0[i])); mv = std::max(mv, (double)fabsf(r0[i])); }
      printf("    split vs fp32 forward: max|diff| %.3e, max|value| %.3e, ratio %.2e\n", md, mv, md / mv);
    }
    if (auto rd = (int (*)(unsigned long long*, int))dlsym(RTLD_DEFAULT, "kd_dbg_read")) {   // dev build with phase counters
      unsigned long long c[8];
      rd(c, 1);
      RC(kd_pwconv_gemm(A, s.K, nullptr, 0, 1, 2, sc, sh, nullptr, nullptr, nullptr, W, nullptr, C, s.N, nullptr, 0, 1, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, partial, kd_pwconv_stat_rows_for(s.M, s.K, s.N, 1, 1, 0), s.M, s.K, s.N, nullptr, nullptr));
      rd(c, 1);
      const double t = (double)c[6];
      printf("    phases (cycles/tile, wave 0): load-wait+transform %.0f | barrier %.0f | split+LDS %.0f | barrier %.0f | MFMA %.0f | epilogue %.0f | tiles %.0f\n",
             c[0] / t, c[1] / t, c[2] / t, c[3] / t, c[4] / t, c[5] / t, t);
    }
    float tc = timeit([&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const float4*)A, (float4*)C, (size_t)s.M * s.K / 4, (size_t)s.M * s.N / 4); });
    printf("%-25s %8ld | %7.1fus %5.1fTF %4.2fTB/s | %7.1fus %5.1fTF | %7.1fus %5.1fTF %4.2fTB/s | %7.1fus %5.1fTF %4.2fTB/s | %7.1fus %4.2fTB/s\n", s.name, s.M,
           t1 * 1e3, fl / t1 / 1e9, by_f / t1 / 1e9, t0 * 1e3, fl / t0 / 1e9, t2 * 1e3, fl / t2 / 1e9, by_d / t2 / 1e9,
           t3 * 1e3, fl / t3 / 1e9, by_w / t3 / 1e9, tc * 1e3, by_f / tc / 1e9);
  }
  return 0;
}
