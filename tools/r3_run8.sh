#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
echo "base: $(timeout -k 10 120 python3 tools/bench_lidar_bwd.py 256 5 fused 2>&1 | grep 'one kernel')" | tee -a $O/variants.log
for p in 1 2 3 4 5; do
  echo "variant $p: $(KD_HIP_LIB=$PWD/tools/dbg/v$p/libkd_hip.so timeout -k 10 120 python3 tools/bench_lidar_bwd.py 256 5 fused 2>&1 | grep 'one kernel')" | tee -a $O/variants.log
done
echo "base: $(timeout -k 10 120 python3 tools/bench_lidar_bwd.py 256 5 fused 2>&1 | grep 'one kernel')" | tee -a $O/variants.log
