"""`SpatialLiDAREncoder(use_vectorized=False)` (the reference's Python double loop,
lidar_encoder.py:101-143) returns bit-for-bit what `forward_vectorized` returns; here both flags run the same device
path, so outputs and gradients must be identical bits, in training and in eval mode, through the dispatcher too."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]


@pytest.mark.parametrize("training", [True, False])
def test_use_vectorized_false_gives_the_same_bits(training):
    import kd_oracle as O
    from src.models.lidar_encoder import LiDAREncoder, SpatialLiDAREncoder
    torch.manual_seed(3)
    a = SpatialLiDAREncoder(grid_size=(16, 16), use_vectorized=True).cuda()
    b = SpatialLiDAREncoder(grid_size=(16, 16), use_vectorized=False).cuda()
    b.load_state_dict(a.state_dict())
    assert a.use_vectorized is True and b.use_vectorized is False
    _, pts, _ = O.make_inputs(2, 64, 700, 16, 9, pad_tail=50)
    pts = pts.cuda()
    pts[0, :3, :2] = torch.tensor([[50.0, 50.0], [-50.0, -50.0], [49.99, 0.0]]).cuda()
    if not training:                                    # (a NaN point poisons train-mode BatchNorm statistics, upstream too)
        pts[0, 3, 0] = float("nan")
    pts[1, 10:20] = pts[1, 10]                          # duplicates: tie-split gradient
    a.train(training); b.train(training)
    ya, yb = a(pts), b(pts)
    assert torch.equal(ya, yb)
    if training:
        g = torch.randn_like(ya)
        ya.backward(g); yb.backward(g)
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert torch.equal(p.grad, q.grad), n
        assert torch.equal(a.point_mlp[1].running_mean, b.point_mlp[1].running_mean)
    # the dispatcher forwards the flag (lidar_encoder.py:194-199) 
    d = LiDAREncoder(encoder_type="spatial", use_vectorized=False, grid_size=(16, 16)).cuda()
    assert d.use_vectorized is False and d.encoder.use_vectorized is False
    d.encoder.load_state_dict(a.state_dict())
    d.train(training)
    assert torch.equal(d(pts).detach(), yb.detach())


@pytest.mark.parametrize("mode", ("eval", "train"))
def test_use_vectorized_false_against_reference_iterative_golden(mode):
    """tests/golden/lidar_iterative.npz: the reference's own forward_iterative outputs on the edge-case points."""
    from _util import golden
    import kd_oracle as O
    from src.models.lidar_encoder import SpatialLiDAREncoder
    gd = golden("lidar_iterative.npz")
    enc = SpatialLiDAREncoder(grid_size=(16, 16), use_vectorized=False).cuda()
    st = O.randomize_state({k: v.cpu() for k, v in enc.state_dict().items()}, 3)
    enc.load_state_dict(st)
    enc.train(mode == "train")
    with torch.no_grad():
        y = enc(torch.from_numpy(gd["points"]).cuda())
    want = torch.from_numpy(gd[f"iter_{mode}_out"])
    assert (y.cpu() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    assert torch.equal(y.cpu() == 0, want == 0)                    # the same cells are empty: binning is bit-exact
