#!/bin/bash
# round-4 evidence run: kernel trace + stats, PMC traffic (2 passes), PMC utilisation (4 passes), depthwise table, launch table
set -u
O=gpurun_out/ev4; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
bash tools/r4_prof.sh ev4 > $O/prof.log 2>&1
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-roofline --no-side-benches"
step pmcF 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o p -- $B
step pmcW 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o p -- $B
step pmc1 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -o p -- $B
step pmc2 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $O/pmc2 -o p -- $B
step pmc3 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d $O/pmc3 -o p -- $B
step pmc4 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc4 -o p -- $B
step dw256 300 ./tools/bench_dw 256
step bench 900 python bench.py --steps 10 --warmup 3 --dump-launches $O/launches.txt
BF="python3 tools/bf16_forward_only.py"
step bf16stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16stats -o p -- $BF
step bf16F 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bf16F -o p -- $BF
step bf16W 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bf16W -o p -- $BF
cat $O/steps.log | tr '\n' ' '; tail -n 1 $O/bench.log | cut -c1-200
