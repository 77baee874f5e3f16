"""GPU parity of the input-preparation kernels (csrc/kd_input.hip) -- integer / byte work, so BIT-EXACT against the
vectors the reference's own functions produced (tests/golden/bev_raster.npz, pandaset_frames.npz) and against the
oracle on fresh seeded inputs; plus size-independent properties at a full PandaSet sweep."""
import numpy as np
import pytest
import torch

import data_oracle as D
from _util import golden

pytestmark = pytest.mark.gpu

CASES = ["binary_64", "multiclass_first_wins_16", "rect_grid_float_range", "boundaries", "all_outside", "empty",
         "all_zero_labels", "full_sweep"]


def test_remap_semantic_matches_reference():
    from src.data_loading.pandaset_dataset import remap_semantic
    g = golden("bev_raster.npz")
    out = remap_semantic(g["remap_in"])
    assert isinstance(out, np.ndarray) and out.dtype == np.int64 and np.array_equal(out, g["remap_out"])
    t = remap_semantic(torch.from_numpy(g["full_sweep_raw"]))
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), g["full_sweep/labels"])
    assert remap_semantic(np.zeros(0, np.int64)).shape == (0,)


@pytest.mark.parametrize("case", CASES)
def test_rasterize_bev_matches_reference(case):
    from src.data_loading.pandaset_dataset import rasterize_bev
    g = golden("bev_raster.npz")
    grid, rng = tuple(int(v) for v in g[case + "/grid"]), tuple(float(v) for v in g[case + "/range"])
    m = rasterize_bev(g[case + "/x"], g[case + "/y"], g[case + "/labels"], grid_size=grid, pc_range=rng)
    assert isinstance(m, np.ndarray) and m.dtype == np.int64 and m.shape == grid
    assert np.array_equal(m, g[case + "/mask"])


def test_rasterize_batch_is_per_frame_and_order_defined():
    """Ragged batch in one launch == each frame alone; first-non-zero-wins is defined by INPUT order: reversing the
    points of a multi-class frame must reproduce the oracle on the reversed order (and differs from the forward one)."""
    from src.data_loading.pandaset_dataset import rasterize_bev_batch
    g = golden("bev_raster.npz")
    names = ["binary_64", "empty", "boundaries", "all_zero_labels"]
    xs, ys, ls = [g[n + "/x"] for n in names], [g[n + "/y"] for n in names], [g[n + "/labels"] for n in names]
    m = rasterize_bev_batch(xs, ys, ls, (64, 64), (-50, 50, -50, 50)).cpu().numpy()
    for k, n in enumerate(names):
        assert np.array_equal(m[k], g[n + "/mask"]), n
    n = "multiclass_first_wins_16"
    x, y, lab = g[n + "/x"][::-1].copy(), g[n + "/y"][::-1].copy(), g[n + "/labels"][::-1].copy()
    rev = rasterize_bev_batch([x], [y], [lab], (16, 16), (-50, 50, -50, 50))[0].cpu().numpy()
    assert np.array_equal(rev, D.rasterize_bev(x, y, lab, (16, 16), (-50, 50, -50, 50)))
    assert not np.array_equal(rev, g[n + "/mask"])


def test_full_sweep_properties():
    """169k points (one PandaSet sweep): remap fused into the rasteriser == remap then rasterise; permuting the points
    leaves a BINARY mask unchanged (first-non-zero == any); mask is the per-cell OR of labels."""
    from src.data_loading.pandaset_dataset import rasterize_bev_batch
    g = golden("bev_raster.npz")
    x, y, raw = g["full_sweep/x"], g["full_sweep/y"], g["full_sweep_raw"]
    fused = rasterize_bev_batch([x], [y], [raw], remap=True)[0].cpu().numpy()
    assert np.array_equal(fused, g["full_sweep/mask"])
    perm = np.random.RandomState(0).permutation(x.size)
    assert np.array_equal(rasterize_bev_batch([x[perm]], [y[perm]], [raw[perm]], remap=True)[0].cpu().numpy(), fused)
    keep, row, col = D.bev_cells(x, y, (64, 64), (-50, 50, -50, 50))
    ref = np.zeros((64, 64), np.int64)
    np.maximum.at(ref, (row, col), g["full_sweep/labels"][keep])
    assert np.array_equal(fused, ref)


def test_points_and_image_preparation():
    from src.data_loading.pandaset_dataset import image_to_chw, prepare_points
    r = np.random.RandomState(3)
    cols = [r.randn(300).astype(np.float32) for _ in range(4)]
    assert np.array_equal(prepare_points(*cols, 400).cpu().numpy(), D.prepare_points(*cols, 400))
    assert np.array_equal(prepare_points(*cols, 300).cpu().numpy(), D.prepare_points(*cols, 300))
    empty = [np.zeros(0, np.float32)] * 4
    assert np.array_equal(prepare_points(*empty, 16).cpu().numpy(), np.zeros((16, 4), np.float32))
    sub = prepare_points(*cols, 128).cpu().numpy()                      # subsample: 128 distinct rows of the input
    have = {row.tobytes() for row in np.stack(cols, 1)}
    assert sub.shape == (128, 4) and all(row.tobytes() in have for row in sub) and len({row.tobytes() for row in sub}) == 128
    img = r.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    assert np.array_equal(image_to_chw(img).cpu().numpy(), D.image_to_chw(img))
    allv = np.arange(256, dtype=np.uint8).repeat(3).reshape(16, 16, 3)  # every byte value: x/255 exactly as numpy rounds it
    assert np.array_equal(image_to_chw(allv).cpu().numpy(), D.image_to_chw(allv))


def test_pandaset_dataset_matches_reference_reader(tmp_path):
    """Fake PandaSet tree -> PandaSetDataset / create_pandaset_dataloaders against what the reference's own
    PandaSetDataset.__getitem__ returned for the same files (tests/golden/pandaset_frames.npz)."""
    from _fake_pandaset import write_tree
    from src.data_loading.pandaset_dataset import PandaSetDataset, create_pandaset_dataloaders
    g = golden("pandaset_frames.npz")
    scenes = write_tree(str(tmp_path))
    ds = PandaSetDataset(str(tmp_path), scenes, max_points=400, verbose=False)
    assert len(ds) == int(g["len"])
    for k in range(len(ds)):
        s = ds[k]
        assert s["sample_token"] == str(g["tokens"][k])
        assert s["image"].shape == (3, 256, 256) and s["image"].dtype == torch.float32
        assert np.array_equal((s["image"].cpu().numpy() * 255.0).round().astype(np.uint8), g[f"{k}/image"])
        assert abs(s["image"].double().sum().item() - float(g[f"{k}/image_f32_sum"])) < 1e-6
        assert s["segmentation"].dtype == torch.int64 and np.array_equal(s["segmentation"].cpu().numpy(), g[f"{k}/segmentation"])
        want, got = g[f"{k}/points"], s["points"].cpu().numpy()
        assert got.shape == want.shape == (400, 4)
        if k != 1:                                                       # frames with <= 400 points: padded, exact
            assert np.array_equal(got, want, equal_nan=True)
    # batched loaders: worker processes do host I/O only, device preparation per batch in this process
    tl, vl = create_pandaset_dataloaders(str(tmp_path), scenes[:1], scenes[1:], batch_size=2, num_workers=2, verbose=False)
    assert len(tl) == 1 and len(vl) == 1
    b = next(iter(vl))
    assert b["image"].shape == (2, 3, 256, 256) and b["image"].is_cuda
    assert b["points"].shape == (2, 5000, 4) and b["segmentation"].shape == (2, 64, 64)
    assert b["sample_token"] == ["002_00", "002_01"]
    for j in range(2):
        assert np.array_equal(b["segmentation"][j].cpu().numpy(), g[f"{2 + j}/segmentation"])
    # host-tensor mode for the reference's analysis scripts (they call .numpy() on the batch)
    _, vl_cpu = create_pandaset_dataloaders(str(tmp_path), scenes[:1], scenes[1:], batch_size=2, num_workers=0, verbose=False,
                                            to_cpu=True)
    bc = next(iter(vl_cpu))
    assert not bc["image"].is_cuda and np.array_equal(bc["segmentation"].numpy(), b["segmentation"].cpu().numpy())


def test_trainer_consumes_device_loader(tmp_path):
    """The reference's training loop (trainer.py:68-95) over the real-format reader."""
    from _fake_pandaset import write_tree
    from _gpu_util import build_product
    from src.data_loading.pandaset_dataset import create_pandaset_dataloaders
    from src.training.trainer import Trainer
    scenes = write_tree(str(tmp_path / "data"), n_points=(6000, 700), degenerate=False)   # NaN points would poison train-mode BN
    tl, vl = create_pandaset_dataloaders(str(tmp_path / "data"), scenes, scenes, batch_size=2, num_workers=0, verbose=False)
    torch.manual_seed(0)
    tr = Trainer(build_product("weighted", 64), tl, vl, torch.device("cuda"), save_dir=str(tmp_path / "ck"),
                 class_weights=[0.4, 3.5], num_epochs=2)
    loss, m = tr.train_epoch()
    vloss, vm = tr.validate()
    assert np.isfinite(loss) and np.isfinite(vloss) and 0.0 <= vm["miou"] <= 1.0
