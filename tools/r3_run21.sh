#!/bin/bash
set -u
O=gpurun_out/r4b; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step dw256 300 ./tools/bench_dw 256
step tests 1100 python -m pytest tests -m gpu -q -k "not fp64"
step seeds 1100 python tools/diag_fp64_seeds.py $O/fp64_clean_seeds.json 1 14 2
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
bash tools/r3_prof.sh r4b > $O/prof.log 2>&1
tail -n 6 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200; head -1 $O/summary.txt; grep "bn_\|slab_\|dw_" $O/summary.txt | cut -c1-140; cat $O/dw256.log | cut -c1-250; grep -c CLEAN $O/seeds.log
