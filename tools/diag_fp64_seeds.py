"""Seed scan for tests/test_gpu_parity.py::test_gradients_against_fp64_oracle: which input seeds keep EVERY evaluation
(CPU fp32 oracle; GPU in both GEMM arithmetics x streaming mode 0 / 1 / 2) of a (objective, student fusion) step free of
activation flips against the float64 oracle.  A flip (a pre-activation within fp32 rounding of a ReLU / ReLU6 / max kink)
shows as 1e-3-class relative error in one evaluation while the others sit at 1e-6; it is a property of the batch, not of
the kernels, and the no-tolerance-games check needs batches without one.

usage: python tools/diag_fp64_seeds.py OUT.json [first_seed [n_seeds [want_per_case]]]
Prints one line per (case, seed) and writes {"clean": {"kd/weighted": [seeds...], ...}, "scan": [...]} to OUT.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from _gpu_util import fp64_gpu_grads, fp64_oracle_grads, fp64_rel_errors  # noqa: E402
from kdrt import ops  # noqa: E402
from kdrt.lib import lib  # noqa: E402

CASES = [("kd", "weighted"), ("kd", "concat"), ("kd", "minimal"), ("ce", "weighted"), ("ce", "concat"), ("ce", "minimal")]


def main():
    out_path = sys.argv[1]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    want = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    torch.set_num_threads(8)
    med = lambda v: v[len(v) // 2]
    clean, scan = {}, []
    prev_arith, prev_mode = ops.get_gemm_arithmetic(), lib.kd_set_gemm_stream(2)
    try:
        for obj, fusion in CASES:
            key = f"{obj}/{fusion}"
            clean[key] = []
            for seed in range(first, first + n):
                if len(clean[key]) >= want:
                    break
                g64 = fp64_oracle_grads(fusion, obj, seed, torch.float64)
                cpu = fp64_rel_errors(g64, fp64_oracle_grads(fusion, obj, seed, torch.float32))
                ok = med(cpu) <= 1e-5 and cpu[-1] <= 5e-5
                cols = [f"cpu32 med {med(cpu):.1e} max {cpu[-1]:.1e}"]
                for arith in ("split", "fp32"):
                    ops.set_gemm_arithmetic(arith)
                    for mode in (0, 1, 2):
                        lib.kd_set_gemm_stream(mode)
                        gpu = fp64_rel_errors(g64, fp64_gpu_grads(fusion, obj, seed))
                        # the test's own criterion, with margin: a clean seed sits at <= 1e-5 / 5e-5 everywhere
                        ok = ok and med(gpu) <= 1e-5 and gpu[-1] <= 4e-5
                        cols.append(f"{arith}/s{mode} med {med(gpu):.1e} max {gpu[-1]:.1e}")
                    if arith == "split":            # round 4: the weight-gradient forms (tiled everywhere / every role-specialised instance)
                        lib.kd_set_gemm_stream(2)
                        for wg in (0, 2):
                            prev_wg = lib.kd_set_wgrad_rs(wg)
                            gpu = fp64_rel_errors(g64, fp64_gpu_grads(fusion, obj, seed))
                            lib.kd_set_wgrad_rs(prev_wg)
                            ok = ok and med(gpu) <= 1e-5 and gpu[-1] <= 4e-5
                            cols.append(f"split/wgrad{wg} med {med(gpu):.1e} max {gpu[-1]:.1e}")
                line = f"{key} seed {seed}: " + " | ".join(cols) + (" | CLEAN" if ok else " | flip")
                print(line, flush=True)
                scan.append(line)
                if ok:
                    clean[key].append(seed)
                with open(out_path, "w") as f:          # keep the file current: a long scan shows progress
                    json.dump({"clean": clean, "scan": scan}, f, indent=1)
    finally:
        ops.set_gemm_arithmetic(prev_arith)
        lib.kd_set_gemm_stream(prev_mode)
    short = [k for k, v in clean.items() if len(v) < want]
    print("clean seeds:", json.dumps(clean), "| short of", want, ":", short, flush=True)


if __name__ == "__main__":
    main()
