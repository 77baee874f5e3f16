// Dev tool: times the depthwise 3x3 kernels (kd_dwconv3x3_fwd / _bwd) on the shapes of the KD step (B = 32 frames)
// and prints algorithmic GB/s (every tensor read / written once).
//   hipcc --offload-arch=gfx950 -O2 tools/bench_dw.cpp -I include -L<csrc> -lkd_hip -o tools/bench_dw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "kd_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) { printf("rc=%d %s (line %d)\n", rc_, kd_last_error_string(), __LINE__); exit(1); } } while (0)
struct S { const char* name; int H, W, C, s; };
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32;
  std::vector<S> shapes = {{"stage2 dw 192 s2 128^2", 128, 128, 192, 2}, {"stage3 dw 384 s1 64^2", 64, 64, 384, 1},
                           {"stage4 dw 384 s2 64^2", 64, 64, 384, 2}, {"stage5 dw 768 s1 32^2", 32, 32, 768, 1},
                           {"fpn/fuse dw 128 s1 64^2", 64, 64, 128, 1}, {"concat dw 256 s1 64^2", 64, 64, 256, 1},
                           {"stage1 dw 32 s1 128^2", 128, 128, 32, 1}, {"head dw 64 s1 64^2", 64, 64, 64, 1}};
  size_t big = (size_t)B * 128 * 128 * 192;
  float *x, *y, *d, *gx, *vec, *partial, *ws, *dw;
  CK(hipMalloc(&x, big * 4)); CK(hipMalloc(&y, big * 4)); CK(hipMalloc(&d, big * 4)); CK(hipMalloc(&gx, big * 4));
  CK(hipMalloc(&vec, 16 * 1024 * 4)); CK(hipMalloc(&partial, (size_t)32768 * 2 * 1024 * 4)); CK(hipMalloc(&dw, 1024 * 9 * 4));
  size_t wsb = 1024ull << 20; CK(hipMalloc(&ws, wsb));
  std::vector<float> h(big);
  for (size_t i = 0; i < big; ++i) h[i] = (float)(((i * 2654435761u) >> 8) & 0xffff) / 65536.f - 0.5f;
  CK(hipMemcpy(x, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(y, h.data(), big * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d, h.data(), big * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vec, h.data(), 16 * 1024 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) fn();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
  float *sc = vec, *sh = vec + 1024, *al = vec + 2048, *be = vec + 3072, *ga = vec + 4096, *mean = vec + 5120, *inv = vec + 6144, *w = vec + 8192;
  printf("%-26s | %-22s | %-22s | %-22s | %-30s\n", "shape", "fwd (+stats)", "bwd data (+stats)", "bwd weight", "bwd data + weight in one call");
  for (auto& s : shapes) {
    const int Ho = (s.H - 1) / s.s + 1, Wo = (s.W - 1) / s.s + 1;
    const double in = 4.0 * B * s.H * s.W * s.C, out = 4.0 * B * Ho * Wo * s.C;
    float tf = timeit([&] { RC(kd_dwconv3x3_fwd(x, sc, sh, 2, w, y, partial, B, s.H, s.W, s.C, s.s, nullptr)); });
    float td = timeit([&] { RC(kd_dwconv3x3_bwd(d, y, al, be, ga, nullptr, nullptr, 0, x, sc, sh, 2, mean, inv, w, gx, partial, nullptr, B, s.H, s.W, s.C, s.s, ws, wsb, nullptr)); });
    float tw = timeit([&] { RC(kd_dwconv3x3_bwd(d, y, al, be, ga, nullptr, nullptr, 0, x, sc, sh, 2, mean, inv, w, nullptr, nullptr, dw, B, s.H, s.W, s.C, s.s, ws, wsb, nullptr)); });
    float tb = timeit([&] { RC(kd_dwconv3x3_bwd(d, y, al, be, ga, nullptr, nullptr, 0, x, sc, sh, 2, mean, inv, w, gx, partial, dw, B, s.H, s.W, s.C, s.s, ws, wsb, nullptr)); });
    float t1 = 0, t2 = 0;
    if (s.s == 1) {                        // the two one-pass stride-1 forms, forced
      int prev = kd_set_dw_bwd_mode(1);
      t1 = timeit([&] { RC(kd_dwconv3x3_bwd(d, y, al, be, ga, nullptr, nullptr, 0, x, sc, sh, 2, mean, inv, w, gx, partial, dw, B, s.H, s.W, s.C, s.s, ws, wsb, nullptr)); });
      kd_set_dw_bwd_mode(2);
      t2 = timeit([&] { RC(kd_dwconv3x3_bwd(d, y, al, be, ga, nullptr, nullptr, 0, x, sc, sh, 2, mean, inv, w, gx, partial, dw, B, s.H, s.W, s.C, s.s, ws, wsb, nullptr)); });
      kd_set_dw_bwd_mode(prev);
    }
    printf("%-26s | %7.1fus %5.2f TB/s    | %7.1fus %5.2f TB/s    | %7.1fus %5.2f TB/s    | %7.1fus %5.2f TB/s (vs %.1fus separately; column walk %.1fus, tile form %.1fus)\n", s.name, tf * 1e3, (in + out) / tf / 1e9,
           td * 1e3, (2 * out + 2 * in) / td / 1e9, tw * 1e3, (2 * out + in) / tw / 1e9, tb * 1e3, (2 * out + 2 * in) / tb / 1e9, (td + tw) * 1e3, t1 * 1e3, t2 * 1e3);
  }
  return 0;
}
