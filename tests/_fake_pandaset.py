"""Writes a tiny PandaSet-shaped directory tree (the on-disk layout pandaset_dataset.py:71-99 indexes):
    <root>/<scene>/camera/front_camera/<frame>.jpg
    <root>/<scene>/lidar/<frame>.pkl            pandas DataFrame with columns x, y, z, i  (float64 on disk)
    <root>/<scene>/annotations/semseg/<frame>.pkl  DataFrame with column class (int64)
Seeded, so the golden generator and the tests see the same bytes (same PIL/pandas build in both)."""
import os

import numpy as np


def frame_arrays(seed: int, n_points: int, degenerate: bool = True):
    r = np.random.RandomState(seed)
    x = r.randn(n_points) * 40.0
    y = r.randn(n_points) * 40.0
    z = r.randn(n_points) * 4.0 - 1.0
    inten = r.randint(0, 256, n_points).astype(np.float64)          # real PandaSet intensity: 0..255
    cls = r.randint(0, 43, n_points).astype(np.int64)
    if degenerate and n_points >= 8:                                               # boundary + degenerate points
        x[:8] = [50.0, -50.0, 49.99, 0.0, 50.0001, -50.0001, np.nan, 12.5]
        y[:8] = [50.0, -50.0, 0.0, 49.99, 0.0, 0.0, 1.0, np.nan]
        cls[:8] = [7, 7, 7, 7, 7, 7, 7, 7]
    img = r.randint(0, 256, (37, 53, 3)).astype(np.uint8)           # odd size: the reader resizes to 256x256
    return x, y, z, inten, cls, img


def write_tree(root, scenes=("001", "002"), frames_per_scene=2, n_points=(300, 700, 64, 0), missing=True, degenerate=True):
    import pandas as pd
    from PIL import Image
    k = 0
    for sid in scenes:
        cam = os.path.join(root, sid, "camera", "front_camera")
        lid = os.path.join(root, sid, "lidar")
        seg = os.path.join(root, sid, "annotations", "semseg")
        for d in (cam, lid, seg):
            os.makedirs(d, exist_ok=True)
        for f in range(frames_per_scene):
            x, y, z, inten, cls, img = frame_arrays(1000 + k, n_points[k % len(n_points)], degenerate)
            fid = f"{f:02d}"
            Image.fromarray(img).save(os.path.join(cam, fid + ".jpg"), quality=92)
            pd.DataFrame({"x": x, "y": y, "z": z, "i": inten}).to_pickle(os.path.join(lid, fid + ".pkl"))
            pd.DataFrame({"class": cls}).to_pickle(os.path.join(seg, fid + ".pkl"))
            k += 1
        if missing:                                                  # a frame without semseg must be skipped (:88)
            Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(cam, "99.jpg"))
    os.makedirs(os.path.join(root, "not_a_scene"), exist_ok=True)    # scene dir lacking the three sub-dirs (:76-77)
    return list(scenes) + ["not_a_scene"]
