#!/usr/bin/env python3
"""Golden vectors for the input-preparation path -- runs ONLY in the build container (needs /root/reference).

Calls the reference's own remap_semantic / rasterize_bev / PandaSetDataset.__getitem__ (module loaded by file
path; nothing is copied) on seeded inputs and a seeded fake PandaSet tree (tests/_fake_pandaset.py) and
stores inputs + expected outputs as data in tests/golden/{bev_raster,pandaset_frames}.npz.

usage:  python oracle/make_golden_data.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.dont_write_bytecode = True


def _load(ref, rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def raster_cases():
    """name -> (x f32, y f32, labels i64, grid, pc_range)"""
    r = np.random.RandomState(7)
    cases = {}
    n = 4000
    x = (r.randn(n) * 40).astype(np.float32); y = (r.randn(n) * 40).astype(np.float32)
    cases["binary_64"] = (x, y, (r.rand(n) < 0.3).astype(np.int64), (64, 64), (-50, 50, -50, 50))
    cases["multiclass_first_wins_16"] = (x, y, r.randint(0, 5, n).astype(np.int64), (16, 16), (-50, 50, -50, 50))
    cases["rect_grid_float_range"] = (x, y, r.randint(0, 3, n).astype(np.int64), (24, 40), (-40.0, 60.5, -30.25, 30.0))
    xb = np.array([50, -50, 49.99, 0, 50.0001, -50.0001, np.nan, 12.5, 0, 0, 49.999996, -49.999996], np.float32)
    yb = np.array([50, -50, 0, 49.99, 0, 0, 1.0, np.nan, 0, 0, 49.999996, -49.999996], np.float32)
    cases["boundaries"] = (xb, yb, np.array([1, 2, 3, 4, 5, 6, 7, 8, 0, 9, 10, 11], np.int64), (64, 64), (-50, 50, -50, 50))
    cases["all_outside"] = ((np.abs(x) + 51).astype(np.float32), y, np.ones(n, np.int64), (64, 64), (-50, 50, -50, 50))
    cases["empty"] = (np.zeros(0, np.float32), np.zeros(0, np.float32), np.zeros(0, np.int64), (64, 64), (-50, 50, -50, 50))
    cases["all_zero_labels"] = (x, y, np.zeros(n, np.int64), (64, 64), (-50, 50, -50, 50))
    big = 169000                                                   # a full PandaSet sweep
    xg = (r.randn(big) * 40).astype(np.float32); yg = (r.randn(big) * 40).astype(np.float32)
    cases["full_sweep"] = (xg, yg, None, (64, 64), (-50, 50, -50, 50))      # labels = remap(raw) below
    return cases, r.randint(0, 43, big).astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    ds = _load(a.ref, "src/data_loading/pandaset_dataset.py", "ref_pandaset_dataset")

    out = {}
    cases, raw_big = raster_cases()
    raw_small = np.arange(-3, 70, dtype=np.int64)
    out["remap_in"], out["remap_out"] = raw_small, ds.remap_semantic(raw_small)
    out["full_sweep_raw"] = raw_big
    for name, (x, y, lab, grid, rng) in cases.items():
        if lab is None:
            lab = ds.remap_semantic(raw_big)
        out[name + "/x"], out[name + "/y"], out[name + "/labels"] = x, y, lab
        out[name + "/grid"], out[name + "/range"] = np.array(grid), np.array(rng, np.float64)
        out[name + "/mask"] = ds.rasterize_bev(x, y, lab, grid_size=grid, pc_range=rng)
    out["cases"] = np.array(list(cases))
    np.savez_compressed(os.path.join(a.out, "bev_raster.npz"), **out)

    from _fake_pandaset import write_tree
    fr = {}
    with tempfile.TemporaryDirectory() as root:
        scenes = write_tree(root)
        d = ds.PandaSetDataset(root, scenes, max_points=400, verbose=False)
        fr["len"] = np.array(len(d))
        toks = []
        for k in range(len(d)):
            np.random.seed(k)                                      # the subsample branch draws from the global RNG
            s = d[k]
            toks.append(s["sample_token"])
            fr[f"{k}/image"] = (s["image"].numpy() * 255.0).round().astype(np.uint8)      # exact: values are u8/255
            fr[f"{k}/image_f32_sum"] = np.array(s["image"].double().sum().item())
            fr[f"{k}/points"] = s["points"].numpy()
            fr[f"{k}/segmentation"] = s["segmentation"].numpy()
        fr["tokens"] = np.array(toks)
    np.savez_compressed(os.path.join(a.out, "pandaset_frames.npz"), **fr)
    print("wrote bev_raster.npz, pandaset_frames.npz:", {k: v.shape for k, v in fr.items() if k.endswith("points")}, toks)


if __name__ == "__main__":
    main()
