"""The reference's own test scripts, replayed against the drop-in modules on the GPU: same constructor calls, inputs and
checks as test_camera_encoder.py:6-52, test_lidar_encoder.py:263-310 (the live part: x4 / same output modes) and
test_fusion_module.py:12-66 (intermediates, stock `F.cross_entropy` + autograd through the HIP Functions)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]


def test_camera_encoder_script():
    from src.models.camera_encoder import TwinLiteEncoder
    dev = torch.device("cuda")
    model = TwinLiteEncoder().to(dev).eval()
    assert model.count_parameters() == 363520 and model.out_channels == 128
    with torch.no_grad():
        assert model(torch.randn(1, 3, 256, 256, device=dev)).shape == (1, 128, 32, 32)
        assert model(torch.randn(1, 3, 512, 512, device=dev)).shape == (1, 128, 64, 64)
        assert model(torch.randn(4, 3, 128, 128, device=dev)).shape == (4, 128, 16, 16)
        ms = TwinLiteEncoder(return_multiscale=True).to(dev).eval()(torch.randn(1, 3, 256, 256, device=dev))
    assert {k: tuple(v.shape) for k, v in ms.items()} == {"stage2": (1, 64, 64, 64), "stage3": (1, 64, 64, 64),
                                                           "stage4": (1, 128, 32, 32), "stage5": (1, 128, 32, 32)}
    assert model.get_feature_info() == {"stage2": 64, "stage3": 64, "stage4": 128, "stage5": 128}


def test_output_modes_script():
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.lidar_encoder import LiDAREncoder
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cam = TwinLiteEncoder(return_multiscale=True).to(dev)
    lid = LiDAREncoder(encoder_type="spatial", grid_size=(64, 64), use_vectorized=True).to(dev)
    images, points = torch.randn(2, 3, 256, 256, device=dev), torch.randn(2, 5000, 4, device=dev)
    kw = dict(num_classes=3, fusion_type="concat", fusion_out_channels=256, camera_fpn_stages=["stage3", "stage4", "stage5"],
              camera_fpn_channels=128)
    with torch.no_grad():                                   # modules left in train mode, as the script does
        assert CompleteSegmentationModel(cam, lid, output_mode="x4", **kw).to(dev)(images, points).shape == (2, 3, 256, 256)
        assert CompleteSegmentationModel(cam, lid, output_mode="same", **kw).to(dev)(images, points).shape == (2, 3, 64, 64)


def test_fusion_module_script():
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.lidar_encoder import LiDAREncoder
    torch.manual_seed(0)
    dev = torch.device("cuda")
    cam = TwinLiteEncoder(return_multiscale=True).to(dev).eval()
    lid = LiDAREncoder(encoder_type="spatial", grid_size=(64, 64), use_vectorized=True).to(dev).eval()
    model = CompleteSegmentationModel(camera_encoder=cam, lidar_encoder=lid, num_classes=3, fusion_type="concat",
                                      fusion_out_channels=256, camera_fpn_stages=["stage3", "stage4", "stage5"],
                                      camera_fpn_channels=128).to(dev)
    assert set(model.get_architecture_summary()) >= {"total_params", "fusion_params"}
    B = 2
    images, points = torch.randn(B, 3, 256, 256, device=dev), torch.randn(B, 5000, 4, device=dev)
    with torch.no_grad():
        logits, mids = model(images, points, return_intermediates=True)
    assert logits.shape == (B, 3, 64, 64)                   # default output_mode="same": the BEV grid (the script's 256x256 is stale)
    assert mids["camera_feat"].shape == mids["lidar_feat"].shape == (B, 128, 64, 64)
    assert mids["pre_fusion"].shape == (B, 256, 64, 64) and mids["post_fusion"].shape == (B, 256, 64, 64)
    model.train()
    images.requires_grad_(True)
    logits = model(images, points, return_intermediates=False)
    labels = torch.randint(low=0, high=3, size=(B, 64, 64), device=dev)
    loss = F.cross_entropy(logits, labels)                   # stock loss + autograd through the HIP Functions
    loss.backward()
    assert torch.isfinite(loss)
    dec = sum((p.grad is not None and p.grad.abs().sum().item() > 0) for p in model.head.parameters())
    fus = sum((p.grad is not None and p.grad.abs().sum().item() > 0) for p in model.fusion.parameters())
    assert dec > 0 and fus > 0
