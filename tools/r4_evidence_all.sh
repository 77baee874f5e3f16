#!/bin/bash
# round-4 evidence in one GPU call: fp64 seed scan on the final kernels, then tools/r4_evidence.sh (kernel trace + stats, PMC traffic, PMC
# utilisation, depthwise table, bench line + launch table, bf16 passes)
mkdir -p gpurun_out/ev4
timeout -k 10 700 python tools/diag_fp64_seeds.py gpurun_out/ev4/fp64_seeds.json 1 20 3 > gpurun_out/ev4/fp64_scan.log 2>&1; echo "scan rc=$?"; tail -2 gpurun_out/ev4/fp64_scan.log | cut -c1-300
bash tools/r4_evidence.sh
