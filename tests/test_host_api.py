"""CPU tests of the host-side mirror of the reference interface: class names, constructor
signatures, state_dict keys/shapes/dtypes, registration order (seed-0 default init digests),
parameter counts, error conventions -- and that the product path refuses CPU tensors loudly."""
import numpy as np
import pytest
import torch

from _util import digest, golden

FUSIONS = (("concat", 256), ("minimal", 128), ("weighted", 128))


def _build(fusion, oc, grid=64, num_classes=2, output_mode="same"):
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.lidar_encoder import LiDAREncoder
    cam = TwinLiteEncoder(return_multiscale=True)
    lid = LiDAREncoder(encoder_type="spatial", grid_size=(grid, grid), use_vectorized=True)
    return CompleteSegmentationModel(cam, lid, num_classes=num_classes, fusion_type=fusion, fusion_out_channels=oc,
                                     camera_fpn_stages=["stage3", "stage4", "stage5"], camera_fpn_channels=128,
                                     output_mode=output_mode)


def test_camera_encoder_default_init_matches_reference():
    from src.models.camera_encoder import TwinLiteEncoder
    pins = golden("pins.npz")
    torch.manual_seed(0)
    enc = TwinLiteEncoder()
    sd = enc.state_dict()
    assert list(sd.keys()) == [str(k) for k in pins["cam_keys"]]
    assert enc.count_parameters() == int(pins["cam_params"]) == 363520      # test_camera_encoder.py output
    assert enc.out_channels == 128
    assert enc.get_feature_info() == {"stage2": 64, "stage3": 64, "stage4": 128, "stage5": 128}
    for v, want in zip(sd.values(), pins["cam_digest"]):
        np.testing.assert_allclose(digest(v.float()), want, rtol=0, atol=0)


@pytest.mark.parametrize("fusion,oc", FUSIONS)
def test_full_model_state_dict_contract(fusion, oc):
    pins = golden("pins.npz")
    torch.manual_seed(0)
    m = _build(fusion, oc)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in pins[f"{fusion}_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in pins[f"{fusion}_shapes"]]
    assert [str(v.dtype) for v in sd.values()] == [str(s) for s in pins[f"{fusion}_dtypes"]]
    for k, v, want in zip(sd.keys(), sd.values(), pins[f"{fusion}_digest"]):
        np.testing.assert_allclose(digest(v.float()), want, rtol=0, atol=0, err_msg=k)
    s = m.get_architecture_summary()
    assert int(s["total_params"].replace(",", "")) == int(pins[f"{fusion}_total"])
    assert int(s["fusion_params"].replace(",", "")) == int(pins[f"{fusion}_fusion"])
    assert s["fusion_type"] == fusion and s["output_mode"] == "same" and s["use_multiscale"] is True


def test_error_conventions():
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.lidar_encoder import LiDAREncoder, PointPillarsLiDAREncoder
    with pytest.raises(ValueError):
        LiDAREncoder(encoder_type="nope")
    with pytest.raises(ValueError):
        _build("bogus", 128)
    with pytest.raises(ValueError):
        _build("concat", 256, output_mode="x8")
    with pytest.raises(ImportError):
        PointPillarsLiDAREncoder()
    lid = LiDAREncoder(encoder_type="pointpillars", grid_size=(64, 64))      # silent fallback, lidar_encoder.py:201-205
    assert lid.encoder_type == "spatial" and lid.get_output_shape() == (128, 64, 64)
    m = _build("concat", 256, num_classes=3, output_mode="x4")               # constructible (state_dict contract)
    assert "head.up1.0.weight" in m.state_dict()


def test_product_path_refuses_cpu_tensors():
    from kdrt import KDError
    m = _build("weighted", 128, grid=16)
    with pytest.raises(KDError):
        m(torch.rand(1, 3, 64, 64), torch.rand(1, 32, 4))


def test_run_mode_tells_inference_from_eval_with_autograd():
    """Inside autograd.Function.forward grad mode is always off, so the wrappers decide the execution mode."""
    from kdrt import units
    assert units.run_mode(True) == 1
    assert units.run_mode(False) == 0
    with torch.no_grad():
        assert units.run_mode(False) == 2 and units.run_mode(True) == 1


def test_point_sort_sharing_is_off_by_default_and_scoped():
    from kdrt import units
    assert units._sort_sharing is False and not units._sort_cache
    units.share_point_bins(True)
    try:
        assert units._sort_sharing is True
        units._sort_cache["points"] = ("key", None, None)
    finally:
        units.share_point_bins(False)
    assert units._sort_sharing is False and not units._sort_cache        # leaving the bracket drops the entries
