# per-launch timing of the bf16 forward under the timing probes (tools/dbg/bf_pN: 1 no stores, 2 no A loads): rocprofv3 kernel trace per build
mkdir -p gpurun_out/q4t; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in base p1 p2; do
  if [ $v = base ]; then unset KD_HIP_LIB; else export KD_HIP_LIB=$R/tools/dbg/bf_$v/libkd_hip.so; fi
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/q4t/$v -o p -- python3 $R/tools/bf16_forward_only.py > $R/gpurun_out/q4t/$v.log 2>&1 || exit 1
done
echo done
