"""End to end on the GPU: the reference's two entry scripts (train_with_fusion_ablation.py, train_pandaset.py) run
unchanged over a PandaSet-format tree -- real reader, device batch preparation, Trainer / KDTrainer, checkpoints, the
results JSON with the reference's published parameter counts (fusion_ablation_results.json:4-5,9-10,14-15)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd")


def _run(script, cwd, env):
    e = dict(os.environ, **env)
    e["PYTHONPATH"] = PKG + os.pathsep + e.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(PKG, script)], cwd=cwd, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_fusion_ablation_then_kd_from_its_teacher(tmp_path):
    from _fake_pandaset import write_tree
    data = tmp_path / "data"
    write_tree(str(data), scenes=("001", "002", "003", "004", "005"), frames_per_scene=2, n_points=(6000, 900), missing=False,
               degenerate=False)
    work = tmp_path / "work"
    work.mkdir()
    env = {"KD_DATA_ROOT": str(data), "KD_EPOCHS": "1", "KD_BATCH_SIZE": "2"}
    out = _run("train_with_fusion_ablation.py", str(work), env)
    assert "BEST FUSION" in out
    res = json.load(open(work / "fusion_ablation_results.json"))
    assert {k: (v["total_params"], v["fusion_params"]) for k, v in res.items()} == {
        "concat": ("573,442", "161,920"), "minimal": ("494,978", "93,056"), "weighted": ("528,132", "126,210")}
    assert all(0.0 <= v["miou"] <= 1.0 for v in res.values())
    teacher = work / "checkpoints" / "fusion_ablation_concat" / "best.pth"
    assert teacher.exists()
    # second run: the concat checkpoint becomes the frozen teacher of every variant (KD training)
    out = _run("train_with_fusion_ablation.py", str(work), dict(env, KD_TEACHER=str(teacher)))
    assert "BEST FUSION" in out


def test_train_pandaset_entry_point(tmp_path):
    """train_pandaset.py as the reference ships it: 3-class concat model, 30 epochs, checkpoints under
    checkpoints/pandaset_weighted (train_pandaset.py:79-163)."""
    from _fake_pandaset import write_tree
    data = tmp_path / "data"
    write_tree(str(data), scenes=("001", "002", "003", "004", "005"), frames_per_scene=2, n_points=(5200, 800), missing=False,
               degenerate=False)
    work = tmp_path / "work"
    work.mkdir()
    out = _run("train_pandaset.py", str(work), {"KD_DATA_ROOT": str(data)})
    assert "Total params" in out
    assert (work / "checkpoints" / "pandaset_weighted" / "latest.pth").exists()
    hist = json.load(open(work / "checkpoints" / "pandaset_weighted" / "training_history.json"))
    assert len(hist["train_loss"]) == 30 and hist["train_loss"][-1] < hist["train_loss"][0]
