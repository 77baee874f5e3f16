#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
for p in 0 1 2 4 8 16 32 6 63; do
  echo "probe $p: $(KD_HIP_LIB=$PWD/tools/dbg/p$p/libkd_hip.so timeout -k 10 120 python3 tools/bench_lidar_bwd.py 256 5 fused 2>&1 | grep 'one kernel')" | tee -a $O/probes.log
done
