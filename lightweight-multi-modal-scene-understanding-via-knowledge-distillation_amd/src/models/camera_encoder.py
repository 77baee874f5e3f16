"""MI355X-native TwinLite camera encoder -- drop-in for the reference's src/models/camera_encoder.py.

Same classes, constructor signatures, attribute names, parameter registration order and
state_dict keys (reference camera_encoder.py:9-123), so seeds, checkpoints and callers carry
over.  The layer objects below are parameter containers only: forward never calls them, it hands
their tensors to the gfx950 kernels through kdrt (MFMA GEMMs for the 1x1 convs -- fp32 storage and
accumulation, products as bf16x3 split pieces on the bf16 matrix pipe by default, exact-fp32 MFMA with KD_GEMM=fp32;
NHWC stencils for the depthwise/stem convs, BatchNorm folded into the consumers' loads).
"""
import torch
import torch.nn as nn

from kdrt import units as U
from kdrt.ops import ACT_NONE, ACT_RELU6


def _conv_bn(layers, cin, cout, k, stride, groups, act):
    layers.append(nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, groups=groups, bias=False))
    layers.append(nn.BatchNorm2d(cout))
    if act:
        layers.append(nn.ReLU6())


class InvertedResidual(nn.Module):
    """MobileNetV2 block: [1x1 expand+BN+ReLU6] -> 3x3 depthwise+BN+ReLU6 -> 1x1 project+BN (+x).
    Keys follow the reference's nn.Sequential `conv` (camera_encoder.py:19-44)."""

    def __init__(self, in_channels, out_channels, stride=1, expansion_ratio=6):
        super().__init__()
        self.use_residual = stride == 1 and in_channels == out_channels
        hidden = int(round(in_channels * expansion_ratio))
        layers = []
        if expansion_ratio != 1:
            _conv_bn(layers, in_channels, hidden, 1, 1, 1, True)
        _conv_bn(layers, hidden, hidden, 3, stride, hidden, True)
        _conv_bn(layers, hidden, out_channels, 1, 1, 1, False)
        self.conv = nn.Sequential(*layers)

    def _units(self):
        mods = list(self.conv)
        units, i = [], 0
        while i < len(mods):
            conv, bn = mods[i], mods[i + 1]
            has_act = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU6)
            kind = "dw" if conv.groups > 1 else "pw"
            units.append(U.UnitSpec(kind, conv, bn, ACT_RELU6 if has_act else ACT_NONE))
            i += 3 if has_act else 2
        return units

    def forward(self, x):
        return U.run_chain(x, self._units(), self.use_residual, self.training)


class TwinLiteEncoder(nn.Module):
    def __init__(self, in_channels=3, base_channels=32, return_multiscale=False):
        super().__init__()
        U.stale_cache_guard(self)
        self.return_multiscale = return_multiscale
        stem = []
        _conv_bn(stem, in_channels, base_channels, 3, 2, 1, True)
        self.stem = nn.Sequential(*stem)
        b = base_channels
        self.stage1 = InvertedResidual(b, b, stride=1, expansion_ratio=1)
        self.stage2 = InvertedResidual(b, b * 2, stride=2, expansion_ratio=6)
        self.stage3 = InvertedResidual(b * 2, b * 2, stride=1, expansion_ratio=6)
        self.stage4 = InvertedResidual(b * 2, b * 4, stride=2, expansion_ratio=6)
        self.stage5 = InvertedResidual(b * 4, b * 4, stride=1, expansion_ratio=6)
        self.feature_channels = {"stage2": b * 2, "stage3": b * 2, "stage4": b * 4, "stage5": b * 4}
        self.out_channels = b * 4

    _kd_accepts_skip = True

    def forward(self, x, _skip_stages=()):
        """`_skip_stages` (not in the reference's signature; keyword-only use by CompleteSegmentationModel): multiscale maps the
        CALLER promises not to read -- they are left out of the returned dict.  Without it the key set is the reference's,
        {stage2, stage3, stage4, stage5}, in train() and eval() alike."""
        stem = U.UnitSpec("stem", self.stem[0], self.stem[1], ACT_RELU6)
        # Training: a block that feeds ONLY the residual block after it hands its output over un-materialised (PairChainFn):
        # the stem -> stage1 always; stage2 -> stage3 unless somebody reads stage2's map (the owning model says per call which
        # multiscale maps it will not read; by default every map is produced, as in the reference).
        w0 = (x.shape[-1] - 1) // 2 + 1
        u1 = self.stage1._units()
        if self.stage1.use_residual and U.chain_pair_ok([stem], u1, self.training, w0):
            x1 = U.run_chain_pair(x, [stem], u1, self.training)
        else:
            x1 = self.stage1(U.run_chain(x, [stem], False, self.training))
        x2 = None
        u2, u3 = self.stage2._units(), self.stage3._units()
        skip2 = (not self.return_multiscale) or "stage2" in _skip_stages
        if skip2 and self.stage3.use_residual and not self.stage2.use_residual and U.chain_pair_ok(u2, u3, self.training, 0):
            x3 = U.run_chain_pair(x1, u2, u3, self.training)
        else:
            x2 = self.stage2(x1)
            x3 = self.stage3(x2)
        x4 = self.stage4(x3)
        x5 = self.stage5(x4)
        if self.return_multiscale and U.grad_routing(self.training):
            # x3 / x4 feed the next stage AND the FPN: let the FPN's backward (which runs first) deposit their gradients for the
            # next stage's data-gradient kernel instead of an autograd accumulation pass over each map (kdrt/units.py: FPNFn)
            if not self.stage4.use_residual:
                x3._kd_deposit = True
            x4._kd_deposit = True
            if self.stage5.use_residual:
                x5._kd_residual_of = x4
        if self.return_multiscale:
            maps = {"stage2": x2, "stage3": x3, "stage4": x4, "stage5": x5}
            return {k: v for k, v in maps.items() if v is not None}
        return x5

    def get_feature_info(self):
        return self.feature_channels

    def count_parameters(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)
