#!/bin/bash
set -u
O=gpurun_out/last3; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests 900 python -m pytest tests/test_gpu_gemm_shapes.py tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_kd_objective.py -q
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-roofline"
step pmcF 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o p -- $B
step pmcW 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o p -- $B
step bench 900 python bench.py
python3 tools/pmc_bench_traffic.py $O/pmcF $O/pmcW | head -3
tail -n 3 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200
