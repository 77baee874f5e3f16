#!/bin/bash
set -u
O=gpurun_out/r3w; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests 600 python -m pytest tests/test_gpu_bf16.py -q -s
step bench 900 python bench.py --steps 10 --warmup 3 --dump-launches $O/launches.txt
step avail 120 rocprofv3 -L
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-roofline"
step pmc1 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -o p -- $B
step pmc2 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc2 -o p -- $B
step pmc3 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc3 -o p -- $B
python3 tools/pmc_gemm_util.py $O/pmc1 $O/pmc2 $O/pmc3 > $O/pmc_util.txt 2>&1
tail -n 5 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-300; head -20 $O/pmc_util.txt
