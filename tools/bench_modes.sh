#!/bin/bash
# Dev tool: bench.py under the three streaming-GEMM modes (selective default / off / all), one line each.
out=gpurun_out/${1:-modes}; mkdir -p $out
for m in all 1 0; do
  KD_GEMM_STREAM=$m python bench.py --steps ${2:-10} --warmup 3 --no-cpu-baseline $([ $m != all ] && echo --no-selfcheck) > $out/bench_$m.json 2> $out/bench_$m.err
  python - <<PY
import json
d = json.load(open("$out/bench_$m.json"))
print("KD_GEMM_STREAM=$m", d["value"], "frames/s", d["ms_per_step"], "ms/step  roofline.frac", d["roofline"]["frac"], json.dumps({k: (v["ms_per_step"], v["frac_of_roofline"]) for k, v in d["roofline"]["by_group"].items()}))
PY
done
