#!/bin/bash
# Register / scratch / LDS use of EVERY kernel in csrc/ as hipcc reports it (-Rpass-analysis=kernel-resource-usage) -> stdout.
# usage: tools/kres_all.sh > profiles/r03_kernel_resources.txt
cd "$(dirname "$0")/../lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd/csrc" || exit 1
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=on -fno-fast-math -fno-slp-vectorize"
echo "hipcc $FLAGS -Rpass-analysis=kernel-resource-usage, every .hip of csrc/ (tools/kres_all.sh); columns: kernel, VGPRs, AGPRs, spilled VGPRs, scratch bytes/lane, waves/SIMD, static LDS"
tot=0; bad=0
for f in *.hip; do
  echo "== $f"
  out=$(/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/kres_$$.o 2>&1 | python3 ../../tools/kres.py)
  echo "$out"
  tot=$((tot + $(echo "$out" | grep -c vgpr)))
  bad=$((bad + $(echo "$out" | awk '$9 != "0"' | grep -c vgpr)))
done
rm -f /tmp/kres_$$.o
echo "== $tot kernels, $bad with scratch"
