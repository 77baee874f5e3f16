#!/bin/bash
O=gpurun_out/r3s; mkdir -p $O
python3 - > $O/acc.log 2>&1 <<'PY'
import sys, os
sys.path[:0] = ["lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd", "oracle", "tests"]
import torch, kd_oracle as O
from kdrt import units
from src.models.lidar_encoder import LiDAREncoder
for (B, N, G) in ((2, 700, 16), (4, 20000, 64)):
    _, pts, _ = O.make_inputs(B, 64, N, G, 9, pad_tail=N // 10)
    enc = LiDAREncoder(encoder_type="spatial", grid_size=(G, G)).cuda()
    st = O.randomize_state({k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}, 33)
    enc.load_state_dict(st); enc.eval()
    st64 = {k: (v.double() if v.is_floating_point() else v) for k, v in st.items()}
    want = O.spatial_lidar_encoder(pts.double(), st64, "encoder.", (G, G), False)
    with torch.no_grad():
        units._LIDAR_FUSED_INFER = True; a = enc(pts.cuda()).double().cpu()
        units._LIDAR_FUSED_INFER = False; b = enc(pts.cuda()).double().cpu()
    s = want.abs().max().item()
    print(B, N, G, "max", s, "fused err", (a - want).abs().max().item() / s, "layered err", (b - want).abs().max().item() / s, "fused-layered", (a - b).abs().max().item() / s)
PY
cat $O/acc.log | tail -4
bash tools/r3_prof.sh r3s > $O/prof.log 2>&1
grep -E "lidar_mlp|pw_gemm_kernel<1, 4|stream_kernel<2, 2, 4, 3, 0" $O/summary.txt | head
head -3 $O/summary.txt
