"""PandaSet batch contract + a synthetic stand-in.

The real reader (reference src/data_loading/pandaset_dataset.py: JPEG + pandas pickles + a Python
BEV rasteriser) is host-side I/O outside the accelerated path and is not rebuilt here (SURVEY.md
section 2 row 7: out of scope; the dataset is not in the container).  What the hot path needs is the
batch CONTRACT it defines (pandaset_dataset.py:104-141), which `SyntheticPandaSet` reproduces:
    image        float32 [3, 256, 256] in [0, 1]
    points       float32 [5000, 4]  (x, y, z, intensity), zero-padded tail
    segmentation int64   [64, 64]   2-class BEV mask
`create_pandaset_dataloaders` keeps the reference signature; it serves synthetic frames when the
PandaSet root does not exist and raises otherwise (plug the reference's reader in there).
"""
import os

import torch
from torch.utils.data import DataLoader, Dataset


class SyntheticPandaSet(Dataset):
    def __init__(self, n_frames=64, num_points=5000, image_size=256, bev_size=64, seed=0, pad_tail=250):
        self.n, self.N, self.hw, self.g, self.seed, self.pad = n_frames, num_points, image_size, bev_size, seed, pad_tail

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        pts = torch.randn(self.N, 4, generator=g)
        pts[:, :2] *= 40.0
        pts[:, 2] = pts[:, 2] * 4.0 - 1.0
        pts[:, 3] = torch.sigmoid(pts[:, 3])
        if self.pad:
            pts[self.N - self.pad:] = 0.0
        seg = (torch.rand(self.g, self.g, generator=g) < 0.13).long()        # ~87 % background (real data)
        return {"image": torch.rand(3, self.hw, self.hw, generator=g), "points": pts, "segmentation": seg,
                "sample_token": f"synthetic_{i:06d}"}


def create_pandaset_dataloaders(root, train_scenes, val_scenes, batch_size=4, num_workers=2, verbose=True):
    if os.path.isdir(root):
        raise NotImplementedError(
            "The PandaSet file reader is host-side I/O outside the MI355X hot path and is not rebuilt; "
            "use the reference's PandaSetDataset here -- the batch contract is unchanged.")
    if verbose:
        print(f"[data] '{root}' not found: serving synthetic PandaSet-shaped frames")
    train = SyntheticPandaSet(n_frames=max(8, 8 * len(train_scenes)), seed=1)
    val = SyntheticPandaSet(n_frames=max(4, 4 * len(val_scenes)), seed=2)
    return (DataLoader(train, batch_size=batch_size, shuffle=True, num_workers=num_workers, pin_memory=True, drop_last=True),
            DataLoader(val, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True))
