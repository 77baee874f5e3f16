"""The eval-mode BatchNorm coefficient cache (kdrt/units.py:_coeffs) must never serve coefficients of weights that
have since been rewritten.  The product path writes parameters and running statistics through raw device pointers
(fused AdamW, kd_bn_finalize_train, hipGraph replays), which torch's tensor versions do not see -- round 1 keyed the
cache on `_version` only and every validate() after the first one in Trainer.train() used stale BatchNorm coefficients.
Check: validate -> train -> validate must equal, bit for bit, a FRESH model loaded from the trained state_dict."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _loaders(n=8, bs=4):
    from torch.utils.data import DataLoader
    from src.data_loading.pandaset_dataset import SyntheticPandaSet
    ds = SyntheticPandaSet(n_frames=n, num_points=1024, image_size=64, bev_size=16, seed=3, pad_tail=64)
    return DataLoader(ds, batch_size=bs, shuffle=False), DataLoader(ds, batch_size=bs, shuffle=False)


def _eval_logits(model, loader):
    model.eval()
    outs = []
    with torch.no_grad():
        for b in loader:
            outs.append(model(b["image"].cuda(), b["points"].cuda()).clone())
    return torch.cat(outs)


@pytest.mark.parametrize("kd", [False, True])
def test_validate_train_validate_uses_fresh_bn_coefficients(tmp_path, kd):
    from _gpu_util import build_product
    from src.training.trainer import KDTrainer, Trainer
    tl, vl = _loaders()
    torch.manual_seed(0)
    model = build_product("weighted", 16)
    kw = dict(lr=1e-2, weight_decay=1e-3, save_dir=str(tmp_path), class_weights=[0.4, 3.5], num_epochs=3)
    if kd:
        teacher = build_product("concat", 16)
        tr = KDTrainer(model, teacher, tl, vl, torch.device("cuda"), **kw)
    else:
        tr = Trainer(model, tl, vl, torch.device("cuda"), **kw)
    first = _eval_logits(model, vl)            # fills the per-BatchNorm eval cache
    v0 = tr.validate()
    for _ in range(2):
        tr.train_epoch()                       # AdamW + running-statistics updates through raw pointers
    v1 = tr.validate()
    second = _eval_logits(model, vl)
    assert not torch.equal(first, second)
    fresh = build_product("weighted", 16)      # no cache entries: coefficients are computed from the loaded tensors
    fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    want = _eval_logits(fresh, vl)
    assert torch.equal(second, want), (second - want).abs().max().item()
    assert v0[0] != v1[0]
    # train again, validate again: still fresh
    tr.train_epoch()
    third = _eval_logits(model, vl)
    fresh.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    assert torch.equal(third, _eval_logits(fresh, vl))


def test_frozen_teacher_keeps_its_cache_across_student_steps():
    """The point of the cache: a frozen teacher's 31 coefficient vectors are computed once, not every step."""
    import kd_oracle as O
    from _gpu_util import build_product, load_random_state
    from kdrt.kd import KDStep
    from kdrt.optim import FusedAdamW
    teacher = build_product("concat", 16); load_random_state(teacher, "concat", 11)
    student = build_product("weighted", 16); load_random_state(student, "weighted", 12)
    student.train()
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
    step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5]).cuda())
    images, pts, labels = (t.cuda() for t in O.make_inputs(2, 64, 512, 16, 4, pad_tail=40))
    step(images, pts, labels)
    bns = [m for m in teacher.modules() if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d))]
    entries = [id(m._kd_eval_cache[1]) for m in bns if hasattr(m, "_kd_eval_cache")]
    assert len(entries) >= 20
    step(images, pts, labels)
    assert entries == [id(m._kd_eval_cache[1]) for m in bns if hasattr(m, "_kd_eval_cache")]
