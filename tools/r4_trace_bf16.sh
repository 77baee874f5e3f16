# kernel trace of the bf16 forward alone (per-launch durations)
mkdir -p gpurun_out/q4u; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/q4u/v2 -o p -- python3 $R/tools/bf16_forward_only.py > $R/gpurun_out/q4u/v2.log 2>&1 && echo done
