#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
python3 - > $O/acc.log 2>&1 <<'PY'
import sys
sys.path[:0] = ["lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd", "oracle", "tests"]
import torch, kd_oracle as O
from kdrt import units
from _gpu_util import build_product, load_random_state, oracle_run, max_err, ftol
B, HW, N, G = 2, 64, 512, 16
for fusion in ("concat", "minimal", "weighted"):
    model = build_product(fusion, G); st = load_random_state(model, fusion, 0); model.eval()
    images, pts, _ = O.make_inputs(B, HW, N, G, 0, pad_tail=40)
    ref = oracle_run(st, fusion, images, pts, G, training=False)
    for flag in (True, False):
        units._LIDAR_FUSED_INFER = flag
        with torch.no_grad():
            logits, mids = model(images.cuda(), pts.cuda(), return_intermediates=True)
        print(fusion, "fused" if flag else "layered", {k: "%.2e / tol %.2e" % (max_err(mids[k], ref[k])[0], ftol(ref[k])) for k in ("lidar_feat", "pre_fusion", "post_fusion")}, "logits %.2e / %.2e" % (max_err(logits, ref["logits"])[0], ftol(ref["logits"])))
PY
grep -v Warn $O/acc.log | tail -6
for m in 0 1 0 1; do
  echo "KD_LIDAR_FUSED_INFER=$m: $(KD_LIDAR_FUSED_INFER=$m timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-selfcheck --no-roofline 2>&1 | grep '^{' | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')" | tee -a $O/ab.log
done
