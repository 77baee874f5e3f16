#!/bin/bash
set -u
O=gpurun_out/r3o; mkdir -p $O
timeout -k 10 180 python -m pytest tests/test_gpu_gemm_shapes.py -q -x -k "fused_backward" > $O/unit.log 2>&1; echo "unit rc=$?"; tail -3 $O/unit.log
grep -q "passed" $O/unit.log || exit 1
timeout -k 10 300 python3 tools/bench_lidar_bwd.py 256 5 both > $O/time.log 2>&1 || exit 1
cat $O/time.log
KD_HIP_LIB=$PWD/tools/dbg/lb/libkd_hip.so timeout -k 10 300 python3 tools/bench_lidar_bwd.py 256 3 fused > $O/dbg.log 2>&1
cat $O/dbg.log
