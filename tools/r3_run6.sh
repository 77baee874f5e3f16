#!/bin/bash
set -u
O=gpurun_out/r3g; mkdir -p $O
KD_HIP_LIB=$PWD/tools/dbg/lb/libkd_hip.so timeout -k 10 300 python3 tools/bench_lidar_bwd.py 256 3 fused > $O/dbg.log 2>&1
cat $O/dbg.log
