"""PandaSet reader with the per-frame preparation on the MI355X -- drop-in for the reference's
src/data_loading/pandaset_dataset.py (same names, signatures, sample dictionary).

Host side (as in the reference): directory indexing (:71-99), JPEG decode + PIL bilinear resize (:105-107),
pandas pickle reads (:114-118,130-131).  Device side (csrc/kd_input.hip through the C ABI): uint8 HWC -> float32
CHW / 255 (:108-111), point stacking + zero padding or subsample gather (:119-127), remap_semantic (:13-20) and
rasterize_bev (:23-45, the reference's 44 ms/frame Python loop).  `dataset[i]` returns the reference's sample
dictionary with CUDA tensors.  The loaders from `create_pandaset_dataloaders` keep DataLoader worker processes for
the host I/O only (`load_raw`: decode + unpickle, nothing touches the GPU there) and prepare each whole batch on
the device in the consuming process (`DeviceBatchLoader`: one rasteriser launch per batch over ragged frames);
`.to(device)` in the trainer is then a no-op.

`SyntheticPandaSet` serves frames of the same contract when there is no dataset on disk:
    image        float32 [3, 256, 256] in [0, 1]
    points       float32 [max_points, 4]  (x, y, z, intensity), zero-padded tail
    segmentation int64   [64, 64]   2-class BEV mask
"""
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from kdrt import KDError
from kdrt.lib import lib
from kdrt.ops import P, stream, workspace

_DRIVABLE = {6, 7, 8, 9, 10, 12}          # Ground, Road, Lane markings, Stop lines, Other markings, Driveway (:13)
_DRIVABLE_BITS = sum(1 << k for k in _DRIVABLE)


def _dev(t, dtype):
    """numpy array / tensor -> contiguous CUDA tensor of `dtype` (the dtype conversion is the reference's host-side
    `to_numpy(dtype=...)`; everything after it runs on the device)."""
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.require(t, requirements=["C", "W"]))
    t = t.to(dtype)
    if not torch.cuda.is_available():
        raise KDError("input preparation runs on the MI355X; there is no CPU fallback")
    return t.cuda().contiguous()


def _f32(v) -> float:
    return float(np.float32(v))


def rasterize_bev_batch(xs: Sequence, ys: Sequence, classes: Sequence, grid_size=(64, 64), pc_range=(-50, 50, -50, 50),
                        remap: bool = False) -> torch.Tensor:
    """B ragged frames in one launch -> int64 [B, H, W] on the device.  remap=True applies remap_semantic first."""
    B = len(xs)
    if B == 0:
        raise KDError("rasterize_bev_batch needs at least one frame")
    lens = [int(np.shape(x)[0]) for x in xs]
    x = torch.cat([_dev(v, torch.float32).reshape(-1) for v in xs])
    y = torch.cat([_dev(v, torch.float32).reshape(-1) for v in ys])
    c = torch.cat([_dev(v, torch.int64).reshape(-1) for v in classes])
    if not (x.numel() == y.numel() == c.numel() == sum(lens)):
        raise KDError("x, y and labels must have the same length per frame")
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64).cuda()
    H, W = int(grid_size[0]), int(grid_size[1])
    x0, x1, y0, y1 = pc_range
    mask = torch.empty(B, H, W, dtype=torch.int64, device="cuda")
    nbytes = lib.kd_bev_rasterize_ws_bytes(B, H, W)
    ws = workspace(nbytes, mask.device)
    lib.call("kd_bev_rasterize", P(x), P(y), P(c), P(off), B, x.numel(), 1 if remap else 0, _DRIVABLE_BITS, H, W,
             _f32(x0), _f32(x1 - x0), _f32(x1), _f32(y0), _f32(y1 - y0), _f32(y1), P(ws), nbytes, P(mask), stream())
    return mask


def remap_semantic(raw_ids):
    """PandaSet raw class ids -> {0 = background, 1 = drivable incl. lanes}.  numpy in -> numpy out (as the
    reference), tensor in -> CUDA tensor out; computed on the device either way."""
    ids = _dev(raw_ids, torch.int64)
    flat = ids.reshape(-1)
    out = torch.empty_like(flat)
    lib.call("kd_semantic_remap", P(flat), flat.numel(), _DRIVABLE_BITS, P(out), stream())
    out = out.reshape(ids.shape)
    return out.cpu().numpy() if isinstance(raw_ids, np.ndarray) else out


def rasterize_bev(x, y, labels, grid_size: Tuple[int, int] = (64, 64),
                  pc_range: Tuple[float, float, float, float] = (-50, 50, -50, 50)):
    """Per-point labels -> BEV mask, first non-zero label per cell in point order.  numpy in -> numpy out."""
    m = rasterize_bev_batch([x], [y], [labels], grid_size, pc_range, remap=False)[0]
    return m.cpu().numpy() if isinstance(x, np.ndarray) else m


def prepare_points(x, y, z, i, max_points: int, generator=None) -> torch.Tensor:
    """float32 [max_points, 4]: zero-padded, or a uniform subset without replacement when the sweep is longer."""
    x, y, z, i = (_dev(v, torch.float32) for v in (x, y, z, i))
    n = x.numel()
    out = torch.empty(max_points, 4, dtype=torch.float32, device="cuda")
    choice = None
    if n > max_points:
        choice = torch.randperm(n, device="cuda", generator=generator)[:max_points].contiguous()
    lib.call("kd_points_prepare", P(x), P(y), P(z), P(i), P(choice), n, max_points, P(out), stream())
    return out


def image_to_chw(img_u8_hwc) -> torch.Tensor:
    t = _dev(np.asarray(img_u8_hwc), torch.uint8)
    if t.dim() != 3 or t.shape[2] != 3:
        raise KDError(f"expected an HxWx3 uint8 image, got {tuple(t.shape)}")
    H, W = int(t.shape[0]), int(t.shape[1])
    out = torch.empty(3, H, W, dtype=torch.float32, device="cuda")
    lib.call("kd_image_u8hwc_to_f32chw", P(t), P(out), H, W, stream())
    return out


class PandaSetDataset(Dataset):
    """2-class version: background (0) and drivable (1, includes lanes)."""

    def __init__(self, root: str, scene_ids: List[str], image_size: Tuple[int, int] = (256, 256),
                 grid_size: Tuple[int, int] = (64, 64), max_points: int = 5000, verbose: bool = True):
        self.root, self.scene_ids = root, scene_ids
        self.image_size, self.grid_size, self.max_points = image_size, grid_size, max_points
        self.pc_range = (-50, 50, -50, 50)
        self.samples = self._index_scenes(verbose=verbose)
        if verbose:
            print(f"Indexed {len(self.samples)} valid samples from {len(scene_ids)} scenes")

    def _index_scenes(self, verbose: bool = True):
        found = []
        for sid in self.scene_ids:
            dirs = {"image": os.path.join(self.root, sid, "camera", "front_camera"),
                    "lidar": os.path.join(self.root, sid, "lidar"),
                    "semseg": os.path.join(self.root, sid, "annotations", "semseg")}
            if not all(os.path.isdir(d) for d in dirs.values()):
                continue
            frames = sorted(f[:-4] for f in os.listdir(dirs["image"]) if f.endswith(".jpg"))
            usable = 0
            for fid in frames:
                paths = {"image": os.path.join(dirs["image"], fid + ".jpg"), "lidar": os.path.join(dirs["lidar"], fid + ".pkl"),
                         "semseg": os.path.join(dirs["semseg"], fid + ".pkl")}
                if all(os.path.exists(p) for p in paths.values()):
                    found.append({"scene": sid, "frame": fid, **paths})
                    usable += 1
            if verbose:
                print(f"Scene {sid}: {usable}/{len(frames)} frames usable")
        return found

    def __len__(self):
        return len(self.samples)

    def load_raw(self, idx: int) -> Dict[str, object]:
        """Host I/O only (safe in DataLoader workers): decoded + resized uint8 image, float32 point columns,
        int64 raw class ids."""
        import pandas as pd
        from PIL import Image
        s = self.samples[idx]
        img = Image.open(s["image"]).convert("RGB").resize(self.image_size, Image.BILINEAR)
        lidar = pd.read_pickle(s["lidar"])
        cols = {c: lidar[c].to_numpy(dtype=np.float32) for c in ("x", "y", "z", "i")}
        raw_ids = pd.read_pickle(s["semseg"])["class"].to_numpy(dtype=np.int64)
        return {"image_u8": np.asarray(img), **cols, "class": raw_ids, "sample_token": f"{s['scene']}_{s['frame']}"}

    def prepare_batch(self, raws: Sequence[Dict[str, object]]) -> Dict[str, object]:
        """Device stage for a list of `load_raw` results -> the collated batch the trainers consume."""
        xs = [_dev(r["x"], torch.float32) for r in raws]
        ys = [_dev(r["y"], torch.float32) for r in raws]
        seg = rasterize_bev_batch(xs, ys, [r["class"] for r in raws], self.grid_size, self.pc_range, remap=True)
        pts = torch.stack([prepare_points(x, y, r["z"], r["i"], self.max_points) for x, y, r in zip(xs, ys, raws)])
        img = torch.stack([image_to_chw(r["image_u8"]) for r in raws])
        return {"image": img, "points": pts, "segmentation": seg, "sample_token": [r["sample_token"] for r in raws]}

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        b = self.prepare_batch([self.load_raw(idx)])
        return {"image": b["image"][0], "points": b["points"][0], "segmentation": b["segmentation"][0],
                "sample_token": b["sample_token"][0]}


class _RawFrames(Dataset):
    """What the worker processes see: host I/O only."""

    def __init__(self, ds: PandaSetDataset):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, idx):
        return self.ds.load_raw(idx)


class RankShardSampler(torch.utils.data.Sampler):
    """Frame indices of ONE rank of a data-parallel job.  `equal=True` (training): every rank gets exactly
    floor(n / world) frames per epoch -- the epoch's permutation (seeded by `seed + epoch`, identical on all ranks) is cut
    to a multiple of `world` and dealt round-robin -- so that, with drop_last batching, all ranks run the SAME number of
    steps and none is left waiting in a gradient all-reduce.  `equal=False` (validation: no collectives inside the
    loop): rank r takes indices r, r + world, ... of the unshuffled set; every frame is seen exactly once job-wide."""

    def __init__(self, n: int, rank: int, world: int, shuffle: bool, equal: bool, seed: int = 0):
        if not 0 <= rank < world:
            raise KDError(f"rank {rank} outside world of {world}")
        self.n, self.rank, self.world, self.shuffle, self.equal, self.seed, self.epoch = n, rank, world, shuffle, equal, seed, 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def _order(self):
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            return torch.randperm(self.n, generator=g).tolist()
        return list(range(self.n))

    def __len__(self):
        if self.equal:
            return self.n // self.world
        return (self.n - self.rank + self.world - 1) // self.world

    def __iter__(self):
        order = self._order()
        if self.equal:
            order = order[: (self.n // self.world) * self.world]
        return iter(order[self.rank::self.world])


def _dist_rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _EpochLoader:
    """DataLoader + `set_epoch` forwarded to a RankShardSampler (the trainers call it once per epoch)."""

    def __init__(self, loader: DataLoader, sampler=None):
        self._loader, self._sampler = loader, sampler
        self.dataset, self.batch_size = loader.dataset, loader.batch_size

    def set_epoch(self, epoch: int):
        if self._sampler is not None:
            self._sampler.set_epoch(epoch)

    def __len__(self):
        return len(self._loader)

    def __iter__(self):
        return iter(self._loader)


class DeviceBatchLoader:
    """DataLoader over raw host frames (any num_workers) + per-batch device preparation in the consumer.
    With `world > 1` the frames are sharded over ranks (RankShardSampler); the training loader then also drops the
    ragged last batch, so every rank runs the same number of steps."""

    def __init__(self, ds: PandaSetDataset, batch_size: int, shuffle: bool, num_workers: int, to_cpu: bool = False,
                 rank: int = 0, world: int = 1, train: bool = None):
        self.dataset = ds
        self.batch_size = batch_size
        self.to_cpu = to_cpu
        train = shuffle if train is None else train
        self._sampler = None
        if world > 1:
            self._sampler = RankShardSampler(len(ds), rank, world, shuffle=shuffle, equal=train)
            self._loader = DataLoader(_RawFrames(ds), batch_size=batch_size, sampler=self._sampler, num_workers=num_workers,
                                      collate_fn=list, drop_last=train)
        else:
            self._loader = DataLoader(_RawFrames(ds), batch_size=batch_size, shuffle=shuffle, num_workers=num_workers,
                                      collate_fn=list)

    def set_epoch(self, epoch: int):
        if self._sampler is not None:
            self._sampler.set_epoch(epoch)

    def __len__(self):
        return len(self._loader)

    def __iter__(self):
        for raws in self._loader:
            b = self.dataset.prepare_batch(raws)
            if self.to_cpu:                 # for host-side analysis scripts that call .numpy() on the batch tensors
                b = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items()}
            yield b


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


def create_pandaset_dataloaders(root: str, train_scenes: List[str], val_scenes: List[str], batch_size: int = 4,
                                num_workers: int = 0, verbose: bool = True, to_cpu: bool = None):
    """Reference signature (pandaset_dataset.py:144-160) plus `to_cpu`: batches stay on the GPU by default (the trainers'
    `.to(device)` is then free); to_cpu=True (or KD_LOADER_TO_CPU=1) returns host tensors for the reference's analysis
    scripts, which call `.numpy()` on them (test_dataset_distribution.py:22, verify_2class_distribution.py)."""
    if to_cpu is None:
        to_cpu = os.environ.get("KD_LOADER_TO_CPU") == "1"
    # under torch.distributed (one process per GPU) the FRAMES are sharded over ranks, equal counts per rank for training
    rank, world = _dist_rank_world()
    if os.path.isdir(root):
        train_ds = PandaSetDataset(root, train_scenes, verbose=verbose)
        val_ds = PandaSetDataset(root, val_scenes, verbose=verbose)
        return (DeviceBatchLoader(train_ds, batch_size, shuffle=True, num_workers=num_workers, to_cpu=to_cpu, rank=rank, world=world),
                DeviceBatchLoader(val_ds, batch_size, shuffle=False, num_workers=num_workers, to_cpu=to_cpu, rank=rank, world=world))
    if verbose:
        print(f"[data] '{root}' not found: serving synthetic PandaSet-shaped frames")
    train = SyntheticPandaSet(n_frames=max(8, 8 * len(train_scenes)), seed=1)
    val = SyntheticPandaSet(n_frames=max(4, 4 * len(val_scenes)), seed=2)
    if world > 1:
        ts = RankShardSampler(len(train), rank, world, shuffle=True, equal=True)
        vs = RankShardSampler(len(val), rank, world, shuffle=False, equal=False)
        return (_EpochLoader(DataLoader(train, batch_size=batch_size, sampler=ts, num_workers=num_workers, pin_memory=True, drop_last=True), ts),
                _EpochLoader(DataLoader(val, batch_size=batch_size, sampler=vs, num_workers=num_workers, pin_memory=True), vs))
    return (DataLoader(train, batch_size=batch_size, shuffle=True, num_workers=num_workers, pin_memory=True, drop_last=True),
            DataLoader(val, batch_size=batch_size, shuffle=False, num_workers=num_workers, pin_memory=True))
