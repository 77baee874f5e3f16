#!/usr/bin/env python3
"""Dev tool: the camera weight-gradient GEMMs of the KD step (kd_pwconv_wgrad, BN-backward on D, BN + act on A) at the benchmarked size
(256 frames), HIP-event timed; KD_HIP_LIB selects a probe build (-DKD_WG_PROBE).  usage: bench_wgrad.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import torch
from kdrt import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
shapes = [("stage2 expand 32->192 @128^2", 128 * 128, 32, 192), ("stage2 project 192->64 @64^2", 64 * 64, 192, 64), ("stage3 expand 64->384", 64 * 64, 64, 384),
          ("stage3 project 384->64", 64 * 64, 384, 64), ("stage4 expand 64->384", 64 * 64, 64, 384), ("stage4 project 384->128 @32^2", 32 * 32, 384, 128),
          ("stage5 expand 128->768", 32 * 32, 128, 768), ("stage5 project 768->128", 32 * 32, 768, 128), ("fpn / proj 128->128", 64 * 64, 128, 128),
          ("head pw 128->64", 64 * 64, 128, 64), ("attention 256->128", 64 * 64, 256, 128), ("concat fuse 256->256", 64 * 64, 256, 256),
          ("fpn lateral 64->128", 64 * 64, 64, 128), ("head (concat) 256->64", 64 * 64, 256, 64)]
g = torch.Generator(device="cuda").manual_seed(1)
v = lambda n: torch.rand(n, device="cuda", generator=g) + 0.5
print(os.environ.get("KD_HIP_LIB", "default lib"))
tot = [0.0, 0.0]
for name, hw, K, N in shapes:
    M = B * hw
    D, X, A = torch.randn(M, N, device="cuda", generator=g), torch.randn(M, N, device="cuda", generator=g), torch.randn(M, K, device="cuda", generator=g)
    dW = torch.empty(N, K, device="cuda")
    al, be, ga, msc, msh, asc, ash = v(N), v(N) * 0.1, v(N) * 0.1, v(N), v(N) - 1.0, v(K), v(K) - 1.0
    def run():
        ops.pw_wgrad(D, A, dW, M=M, N=N, K=K, X=X, d_mode=2, d_act=2, al=al, be=be, ga=ga, msc=msc, msh=msh, a_mode=1, a_act=2, asc=asc, ash=ash)
    res = []
    for form in (0, 1):                      # 0: tiled pw_wgrad_kernel, 1: role-specialised (kd_wgrad_rs.hip) where the layer has one
        ops.lib.kd_set_wgrad_rs(2 * form)
        run(); run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record(); torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / 5 * 1e3, dW.clone()))
    ops.lib.kd_set_wgrad_rs(1)
    by = 4.0 * (2 * M * N + M * K)
    floor = max(by / 8e12, 2.0 * M * N * K * 6 / 2.5e15) * 1e6
    (us0, w0), (us1, w1) = res
    tot[0] += us0; tot[1] += us1
    dev = float((w0 - w1).abs().max() / w0.abs().max())
    print(f"  {name:34s} M={M:9d}  tiled {us0:8.1f} us {by / us0 / 1e6:5.2f} TB/s | rs {us1:8.1f} us {by / us1 / 1e6:5.2f} TB/s {2.0 * M * N * K / us1 / 1e6:6.1f} TFLOP/s"
          f" | x{us0 / us1:4.2f} | floor {floor:7.1f} us -> {floor / us1:4.2f} | rel dev {dev:.1e}")
    del D, X, A
print(f"  sum tiled {tot[0]:.0f} us, role-specialised {tot[1]:.0f} us")
