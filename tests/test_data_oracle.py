"""The data-preparation oracle (oracle/data_oracle.py) against vectors the reference's own functions produced
(tests/golden/bev_raster.npz, pandaset_frames.npz; generator oracle/make_golden_data.py).  Integer work: bit-exact."""
import os

import numpy as np
import pytest

import data_oracle as D
from _util import golden


def test_remap_semantic():
    g = golden("bev_raster.npz")
    assert np.array_equal(D.remap_semantic(g["remap_in"]), g["remap_out"])
    assert D.remap_semantic(g["remap_in"]).dtype == np.int64


@pytest.mark.parametrize("case", ["binary_64", "multiclass_first_wins_16", "rect_grid_float_range", "boundaries",
                                  "all_outside", "empty", "all_zero_labels", "full_sweep"])
def test_rasterize_bev(case):
    g = golden("bev_raster.npz")
    assert case in set(g["cases"])
    grid, rng = tuple(int(v) for v in g[case + "/grid"]), tuple(float(v) for v in g[case + "/range"])
    m = D.rasterize_bev(g[case + "/x"], g[case + "/y"], g[case + "/labels"], grid, rng)
    assert m.dtype == np.int64 and m.shape == grid
    assert np.array_equal(m, g[case + "/mask"])
    if case == "full_sweep":
        assert np.array_equal(D.remap_semantic(g["full_sweep_raw"]), g[case + "/labels"])


def test_fake_pandaset_frames(tmp_path):
    """Oracle restatement of PandaSetDataset.__getitem__ (pandaset_dataset.py:104-141) on the seeded fake tree."""
    import pandas as pd
    from PIL import Image
    from _fake_pandaset import write_tree
    g = golden("pandaset_frames.npz")
    root = str(tmp_path)
    write_tree(root)
    toks = [str(t) for t in g["tokens"]]
    assert int(g["len"]) == len(toks) == 4                        # frame 99 (no lidar/semseg) and "not_a_scene" are skipped
    for k, tok in enumerate(toks):
        sid, fid = tok.split("_")
        img = Image.open(os.path.join(root, sid, "camera", "front_camera", fid + ".jpg")).convert("RGB").resize((256, 256), Image.BILINEAR)
        chw = D.image_to_chw(np.asarray(img))
        assert np.array_equal((chw * 255.0).round().astype(np.uint8), g[f"{k}/image"])
        assert abs(float(chw.astype(np.float64).sum()) - float(g[f"{k}/image_f32_sum"])) < 1e-6
        df = pd.read_pickle(os.path.join(root, sid, "lidar", fid + ".pkl"))
        x, y, z, i = (df[c].to_numpy(dtype=np.float32) for c in "xyzi")
        raw = pd.read_pickle(os.path.join(root, sid, "annotations", "semseg", fid + ".pkl"))["class"].to_numpy(dtype=np.int64)
        assert np.array_equal(D.rasterize_bev(x, y, D.remap_semantic(raw)), g[f"{k}/segmentation"])
        want = g[f"{k}/points"]
        if x.size <= 400:
            assert np.array_equal(D.prepare_points(x, y, z, i, 400), want, equal_nan=True)
        else:                                                      # random subset without replacement of the frame's points
            have = {r.tobytes() for r in np.stack([x, y, z, i], 1)}
            assert all(r.tobytes() in have for r in want) and len({r.tobytes() for r in want}) == 400
