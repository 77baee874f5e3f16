#!/bin/bash
set -u
O=gpurun_out/r3z; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step diag 300 python tools/diag_resize_determinism.py
step dw256 300 ./tools/bench_dw 256
step tests 1100 python -m pytest tests -m gpu -q
bash tools/r3_prof.sh r3z > $O/prof.log 2>&1
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
cat $O/diag.log | tail -24; tail -n 6 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200; head -1 $O/summary.txt; grep "bn_\|slab_" $O/summary.txt | cut -c1-140; cat $O/dw256.log | cut -c1-250
