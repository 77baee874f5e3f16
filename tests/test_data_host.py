"""Host half of the PandaSet reader (indexing + load_raw: the part DataLoader workers run) -- no GPU needed.
Indexing rules follow pandaset_dataset.py:71-99: scenes lacking a sub-directory and frames lacking any of the
three files are skipped; frames sort by name."""
import numpy as np
import pytest
import torch

from _util import golden


def test_index_and_raw_frames(tmp_path):
    from _fake_pandaset import write_tree
    from src.data_loading.pandaset_dataset import PandaSetDataset, _RawFrames
    g = golden("pandaset_frames.npz")
    scenes = write_tree(str(tmp_path))
    ds = PandaSetDataset(str(tmp_path), scenes, max_points=400, verbose=False)
    assert len(ds) == int(g["len"]) == 4
    assert [f"{s['scene']}_{s['frame']}" for s in ds.samples] == [str(t) for t in g["tokens"]]
    assert ds.pc_range == (-50, 50, -50, 50) and ds.grid_size == (64, 64) and ds.image_size == (256, 256)
    raw = _RawFrames(ds)
    for k in range(len(raw)):
        r = raw[k]
        assert r["image_u8"].shape == (256, 256, 3) and r["image_u8"].dtype == np.uint8
        assert np.array_equal(r["image_u8"].transpose(2, 0, 1), g[f"{k}/image"])       # what the reference turned into /255 floats
        assert r["x"].dtype == np.float32 and r["class"].dtype == np.int64 and r["x"].shape == r["class"].shape
        if r["x"].size <= 400:
            pts = np.stack([r["x"], r["y"], r["z"], r["i"]], 1)
            assert np.array_equal(pts, g[f"{k}/points"][: r["x"].size], equal_nan=True)
    assert PandaSetDataset(str(tmp_path), ["not_a_scene", "absent"], verbose=False).samples == []


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_device_stage_refuses_to_run_on_cpu(tmp_path):
    from _fake_pandaset import write_tree
    from kdrt import KDError
    from src.data_loading.pandaset_dataset import PandaSetDataset, rasterize_bev
    scenes = write_tree(str(tmp_path))
    ds = PandaSetDataset(str(tmp_path), scenes, max_points=400, verbose=False)
    with pytest.raises(KDError):
        ds[0]
    with pytest.raises(KDError):
        rasterize_bev(np.zeros(4, np.float32), np.zeros(4, np.float32), np.ones(4, np.int64))


def test_synthetic_fallback_loader_contract():
    from src.data_loading.pandaset_dataset import create_pandaset_dataloaders
    tl, vl = create_pandaset_dataloaders("/nonexistent/pandaset", ["001"], ["002"], batch_size=2, num_workers=0, verbose=False)
    b = next(iter(vl))
    assert b["image"].shape == (2, 3, 256, 256) and b["points"].shape == (2, 5000, 4)
    assert b["segmentation"].shape == (2, 64, 64) and b["segmentation"].dtype == torch.int64
