"""MI355X-native fusion / head / full model -- drop-in for the reference's src/models/fusion_module.py.

Class names, constructor signatures, attribute names (`camera_proj` vs `cam_proj`, `fuse`,
`attention`, `laterals`, `post`, `block`, `cls`, ...), registration order and state_dict keys follow
the reference (fusion_module.py:8-286).  Forward passes run through kdrt's HIP Functions; the
layer objects only hold parameters.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from kdrt import units as U
from kdrt.ops import ACT_RELU


def _pw_unit(seq, i=0):
    return U.UnitSpec("pw", seq[i], seq[i + 1], ACT_RELU)


def _dw_unit(seq, i=0):
    return U.UnitSpec("dw", seq[i], seq[i + 1], ACT_RELU)


class Conv1x1(nn.Module):
    def __init__(self, in_ch, out_ch, bias=False):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_ch, out_ch, kernel_size=1, bias=bias), nn.BatchNorm2d(out_ch), nn.ReLU())

    def unit(self):
        return _pw_unit(self.conv)

    def forward(self, x):
        return U.run_chain(x, [self.unit()], False, self.training)


class DWSeparableConv(nn.Module):
    """Depthwise 3x3 + BN + ReLU, pointwise 1x1 + BN + ReLU."""

    def __init__(self, in_ch, out_ch, stride=1):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(in_ch, in_ch, kernel_size=3, stride=stride, padding=1, groups=in_ch, bias=False),
            nn.BatchNorm2d(in_ch), nn.ReLU(),
            nn.Conv2d(in_ch, out_ch, kernel_size=1, bias=False), nn.BatchNorm2d(out_ch), nn.ReLU())

    def units(self):
        return [_dw_unit(self.net, 0), _pw_unit(self.net, 3)]

    def forward(self, x):
        return U.run_chain(x, self.units(), False, self.training)


class CameraFPNLite(nn.Module):
    def __init__(self, in_channels_by_stage: Dict[str, int], target_channels: int = 128,
                 stages_to_use: Optional[List[str]] = None, target_size: Optional[Tuple[int, int]] = None):
        super().__init__()
        self.stages_to_use = stages_to_use or list(in_channels_by_stage.keys())
        self.laterals = nn.ModuleDict()
        for s in self.stages_to_use:
            self.laterals[s] = Conv1x1(in_channels_by_stage[s], target_channels)
        self.post = DWSeparableConv(target_channels, target_channels)
        self.target_size = target_size

    def forward(self, feats: Dict[str, torch.Tensor]) -> torch.Tensor:
        xs = [feats[s] for s in self.stages_to_use]
        lats = [self.laterals[s].unit() for s in self.stages_to_use]
        return U.run_fpn(xs, lats, self.post.units(), self.training, self.target_size)


def _match_size(cam_feat, lidar_feat):
    """Bilinear (align_corners=False) resize of the LiDAR map to the camera map when they differ
    (reference fusion_module.py:239-240 and the fusion blocks' own forwards)."""
    if cam_feat.shape[-2:] != lidar_feat.shape[-2:]:
        lidar_feat = U.run_resize(lidar_feat, cam_feat.shape[-2:])
    return lidar_feat


class ConcatenationFusion(nn.Module):
    def __init__(self, camera_channels=128, lidar_channels=128, out_channels=256):
        super().__init__()
        self.camera_proj = Conv1x1(camera_channels, camera_channels)
        self.lidar_proj = Conv1x1(lidar_channels, lidar_channels)
        in_cat = camera_channels + lidar_channels
        self.fuse = nn.Sequential(
            nn.Conv2d(in_cat, in_cat, kernel_size=3, padding=1, groups=in_cat, bias=False), nn.BatchNorm2d(in_cat),
            nn.ReLU(), nn.Conv2d(in_cat, out_channels, kernel_size=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU())

    def run(self, cam_feat, lidar_feat, want_pre=True):
        lidar_feat = _match_size(cam_feat, lidar_feat)
        return U.run_concat_fuse(cam_feat, lidar_feat, self.camera_proj.unit(), self.lidar_proj.unit(),
                                 [_dw_unit(self.fuse, 0), _pw_unit(self.fuse, 3)], self.training, want_pre)

    def forward(self, cam_feat, lidar_feat):
        return self.run(cam_feat, lidar_feat, want_pre=False)[0]


class MinimalFusion(nn.Module):
    def __init__(self, cam_ch=128, lidar_ch=128, out_ch=128):
        super().__init__()
        self.cam_proj = Conv1x1(cam_ch, out_ch)
        self.lidar_proj = Conv1x1(lidar_ch, out_ch)

    def forward(self, cam_feat, lidar_feat):
        lidar_feat = _match_size(cam_feat, lidar_feat)
        return U.run_minimal_fuse(cam_feat, lidar_feat, self.cam_proj.unit(), self.lidar_proj.unit(), self.training)


class WeightedFusion(nn.Module):
    def __init__(self, cam_ch=128, lidar_ch=128, out_ch=128):
        super().__init__()
        self.cam_proj = Conv1x1(cam_ch, out_ch)
        self.lidar_proj = Conv1x1(lidar_ch, out_ch)
        self.attention = nn.Sequential(nn.Conv2d(out_ch * 2, out_ch, kernel_size=1), nn.ReLU(),
                                       nn.Conv2d(out_ch, 2, kernel_size=1), nn.Softmax(dim=1))

    def forward(self, cam_feat, lidar_feat):
        lidar_feat = _match_size(cam_feat, lidar_feat)
        return U.run_weighted_fuse(cam_feat, lidar_feat, self.cam_proj.unit(), self.lidar_proj.unit(),
                                   self.attention[0], self.attention[2], self.training)


class LightweightSegmentationHead(nn.Module):
    """Two ConvTranspose2d x2 upsampling stages + 3x3 classifier (`output_mode="x4"`)."""

    def __init__(self, in_channels=256, num_classes=2):
        super().__init__()
        self.up1 = nn.Sequential(nn.ConvTranspose2d(in_channels, 64, kernel_size=4, stride=2, padding=1, bias=False),
                                 nn.BatchNorm2d(64), nn.ReLU())
        self.up2 = nn.Sequential(nn.ConvTranspose2d(64, 16, kernel_size=4, stride=2, padding=1, bias=False),
                                 nn.BatchNorm2d(16), nn.ReLU())
        self.cls = nn.Conv2d(16, num_classes, kernel_size=3, padding=1)

    def units(self):
        return [U.UnitSpec("ct", self.up1[0], self.up1[1], ACT_RELU),
                U.UnitSpec("ct", self.up2[0], self.up2[1], ACT_RELU)]

    def forward(self, x):
        return U.run_x4_head(x, self.units(), self.cls, self.training)


class SameResolutionSegmentationHead(nn.Module):
    def __init__(self, in_channels=256, num_classes=2):
        super().__init__()
        self.block = nn.Sequential(DWSeparableConv(in_channels, 64), DWSeparableConv(64, 32))
        self.cls = nn.Conv2d(32, num_classes, kernel_size=1)

    def forward(self, x):
        return U.run_same_head(x, self.block[0].units() + self.block[1].units(), self.cls, self.training)


class CompleteSegmentationModel(nn.Module):
    def __init__(self, camera_encoder: nn.Module, lidar_encoder: nn.Module, num_classes: int = 2,
                 fusion_type: str = "concat", fusion_out_channels: int = 256,
                 camera_fpn_stages: Optional[List[str]] = None, camera_fpn_channels: int = 128,
                 output_mode: str = "same"):
        super().__init__()
        U.stale_cache_guard(self)
        self.camera_encoder = camera_encoder
        self.lidar_encoder = lidar_encoder
        self.fusion_type = fusion_type
        self.output_mode = output_mode
        self.use_multiscale = getattr(camera_encoder, "return_multiscale", False)
        self.camera_fpn = None
        if self.use_multiscale:
            self.camera_fpn = CameraFPNLite(in_channels_by_stage=camera_encoder.get_feature_info(),
                                            target_channels=camera_fpn_channels, stages_to_use=camera_fpn_stages)
            # tell the encoder which multiscale maps nobody reads (it may then skip materialising them in training)
            # maps of the encoder nobody downstream reads: passed PER CALL (`_skip_stages`), never set on the encoder -- called
            # directly, the encoder returns the reference's full key set in train() and eval() alike (camera_encoder.py:105-115)
            self._cam_skip = tuple(s for s in camera_encoder.get_feature_info() if s not in self.camera_fpn.stages_to_use)
            cam_ch = camera_fpn_channels
        else:
            cam_ch = getattr(camera_encoder, "out_channels", 128)
        lidar_ch = getattr(getattr(lidar_encoder, "encoder", lidar_encoder), "feature_dim", 128)
        if fusion_type == "concat":
            self.fusion = ConcatenationFusion(cam_ch, lidar_ch, fusion_out_channels)
            head_in = fusion_out_channels
        elif fusion_type == "minimal":
            self.fusion = MinimalFusion(cam_ch=cam_ch, lidar_ch=lidar_ch, out_ch=cam_ch)
            head_in = cam_ch
        elif fusion_type == "weighted":
            self.fusion = WeightedFusion(cam_ch=cam_ch, lidar_ch=lidar_ch, out_ch=cam_ch)
            head_in = cam_ch
        else:
            raise ValueError(f"Unknown fusion_type: {fusion_type}")
        if output_mode == "x4":
            self.head = LightweightSegmentationHead(in_channels=head_in, num_classes=num_classes)
        elif output_mode == "same":
            self.head = SameResolutionSegmentationHead(in_channels=head_in, num_classes=num_classes)
        else:
            raise ValueError(f"Unknown output_mode: {output_mode}")

    def forward(self, images: torch.Tensor, points: torch.Tensor, return_intermediates=False):
        """`return_intermediates`: the reference's bool (True: all five maps of fusion_module.py:260-262), or -- an extension --
        a collection of their names, in which case only those are produced (the KD step asks for camera_feat / lidar_feat /
        logits and so spares the concat block a 2 GB pass for a `pre_fusion` nobody reads)."""
        names = None
        if return_intermediates and not isinstance(return_intermediates, bool):
            names = set(return_intermediates)
            unknown = names - {"camera_feat", "lidar_feat", "pre_fusion", "post_fusion", "logits"}
            if unknown:
                raise ValueError(f"unknown intermediates {sorted(unknown)}")
        if self.use_multiscale and getattr(self.camera_encoder, "_kd_accepts_skip", False):
            cam_raw = self.camera_encoder(images, _skip_stages=self._cam_skip)
        else:
            cam_raw = self.camera_encoder(images)
        cam_feat = self.camera_fpn(cam_raw) if isinstance(cam_raw, dict) else cam_raw
        lidar_feat = _match_size(cam_feat, self.lidar_encoder(points))       # fusion_module.py:238-240
        if isinstance(self.fusion, ConcatenationFusion):
            want_pre = bool(return_intermediates) and (names is None or "pre_fusion" in names)
            fused, pre_fusion = self.fusion.run(cam_feat, lidar_feat, want_pre=want_pre)
        else:
            fused = pre_fusion = self.fusion(cam_feat, lidar_feat)
        logits = self.head(fused)
        if return_intermediates:
            mids = {"camera_feat": cam_feat, "lidar_feat": lidar_feat, "pre_fusion": pre_fusion,
                    "post_fusion": fused, "logits": logits}
            return logits, (mids if names is None else {k: v for k, v in mids.items() if k in names})
        return logits

    def get_architecture_summary(self):
        n = lambda m: sum(p.numel() for p in m.parameters())
        fusion = n(self.fusion) + (n(self.camera_fpn) if self.camera_fpn is not None else 0)
        return {"camera_params": f"{n(self.camera_encoder):,}", "lidar_params": f"{n(self.lidar_encoder):,}",
                "fusion_params": f"{fusion:,}", "head_params": f"{n(self.head):,}", "total_params": f"{n(self):,}",
                "fusion_type": self.fusion_type, "output_mode": self.output_mode, "use_multiscale": self.use_multiscale}
