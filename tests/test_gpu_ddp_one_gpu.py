"""The data-parallel GPU product path with 2 ranks on ONE MI355X (gloo between them): HIP backward -> GradSink ->
BucketedAllReduce -> fused AdamW equals two independent single-rank runs whose gradients are averaged; buckets fire
head -> fusion/FPN/LiDAR -> camera; BatchNorm statistics stay per rank; rank 0's weights are broadcast.  Also the
reference-style entry script under torchrun with a frame count that does not divide by the world size, and
`bench.py --gpus 2` starting its own ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
PKG = os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(script, nproc, env, cwd=None, args=()):
    e = dict(os.environ, **env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["OMP_NUM_THREADS"] = "2"
    e["PYTHONPATH"] = PKG + os.pathsep + e.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), script, *args]
    r = subprocess.run(cmd, cwd=cwd, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_two_ranks_equal_two_replicas_with_averaged_gradients(tmp_path):
    _torchrun(os.path.join(HERE, "_ddp_gpu_worker.py"), 2, {"KD_DDP_OUT": str(tmp_path)})
    res = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r in res:
        assert r["orders"] == [[2, 1, 0], [2, 1, 0]], r["orders"]        # bucket launch order, both steps
        assert r["grad_err"] <= 1e-6, json.dumps(r)                                    # summed gradients == replica 0 + replica 1
        assert r["param_err"] <= 1e-7, r                                   # one AdamW step on the averaged gradient
        assert r["bn_own"] == 0.0 and r["bn_other"] > 0.0, r               # BatchNorm buffers stay per rank
        assert r["grad_scale"] == 0.5 and r["ranks_agree"]
    assert res[0]["total"] != res[1]["total"]                              # different shards of the batch


@pytest.mark.parametrize("k", (2, 4))
def test_k_ranks_against_reference_replicas(tmp_path, k):
    """the data-parallel step against k REFERENCE replicas (oracle/make_golden.py section 11)."""
    _torchrun(os.path.join(HERE, "_ddp_ref_worker.py"), k, {"KD_DDP_OUT": str(tmp_path)})
    for r in range(k):
        res = json.load(open(tmp_path / f"ref_rank{r}.json"))
        assert res["loss_err"] < 1e-4, res
        assert res["bn_stem_err"] < 1e-5 and res["bn_lidar_err"] < 1e-4, res      # per-replica BatchNorm statistics
        assert not res["bad_grad_digests"] and not res["bad_param_digests"] and not res["bad_params_vs_adamw_on_own_grads"], res
        assert res["head_cls_err"] < 2e-3 and res["stem_err"] < 5e-3, res
        assert res["grad_scale"] == 1.0 / k


def test_fusion_ablation_entry_script_under_torchrun_with_ragged_shards(tmp_path):
    """train_with_fusion_ablation.py, 2 ranks, 9 training frames (scenes of 3 frames, 3 train scenes): frames are
    sharded in equal counts, both ranks run the same number of steps, CE training is synchronised (base Trainer),
    only rank 0 writes files."""
    from _fake_pandaset import write_tree
    data = tmp_path / "data"
    write_tree(str(data), scenes=("001", "002", "003", "004"), frames_per_scene=3, n_points=(6000, 900, 1500), missing=False,
               degenerate=False)
    work = tmp_path / "work"
    work.mkdir()
    env = {"KD_DATA_ROOT": str(data), "KD_EPOCHS": "2", "KD_BATCH_SIZE": "2", "KD_REHEARSE_ON_ONE_GPU": "1"}
    out = _torchrun(os.path.join(PKG, "train_with_fusion_ablation.py"), 2, env, cwd=str(work))
    assert out.count("BEST FUSION") == 1                                   # one console
    res = json.load(open(work / "fusion_ablation_results.json"))
    assert set(res) == {"concat", "minimal", "weighted"} and all(0.0 <= v["miou"] <= 1.0 for v in res.values())
    hist = json.load(open(work / "checkpoints" / "fusion_ablation_weighted" / "training_history.json"))
    assert len(hist["train_loss"]) == 2


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts 2 ranks itself, they verify
    WORLD_SIZE == --gpus and count each other through an all-reduce.  (Both on cuda:0 over gloo: a rehearsal.)"""
    e = dict(os.environ, KD_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--points",
           "2048", "--no-cpu-baseline", "--student-fusion", "minimal"]
    r = subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2 and len(out["ranks"]) == 2
    assert out["config"]["student_fusion"] == "minimal" and out["config"]["global_batch"] == 8
    assert out["checks"]["finite"] and out["checks"]["selfcheck"]["ok"]
    # N > 1 lines carry the communication evidence: bucket sizes and the time the compute stream waited for the reductions
    comm = out["comm"]
    assert len(comm["bucket_bytes"]) == 3 and sum(comm["bucket_bytes"]) >= 4 * 494978 and comm["collectives_per_step"] == 3
    assert comm["steps_measured"] == 2 and 0.0 <= comm["allreduce_exposed_ms_per_step"] < 1e4
    assert comm["allreduce_exposed_ms_per_step_max_over_ranks"] >= comm["allreduce_exposed_ms_per_step"] - 1e-9
    # a launcher that started the wrong number of ranks is refused
    e2 = dict(e, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r2 = subprocess.run(cmd, env=e2, capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stdout + r2.stderr)
