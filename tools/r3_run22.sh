#!/bin/bash
set -u
O=gpurun_out/r4c; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step seeds 1100 python tools/diag_fp64_seeds.py $O/fp64_clean_seeds.json 1 20 3
python tools/fp64_seed_table.py $O/fp64_clean_seeds.json $O/seeds.log
step tests 1100 python -m pytest tests -m gpu -q
step bench 900 python bench.py --steps 10 --warmup 3
tail -n 6 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200
