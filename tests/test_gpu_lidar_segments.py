"""Cell-sorted (segmented) scatter-max against the atomic scatter-max entry points, through the C ABI.
Both implement lidar_encoder.py:57-99 (BEV binning + amax + the even tie split of its backward); the
atomic pair is already pinned to the oracle and the golden vectors (test_gpu_parity.py / test_gpu_units.py),
so here the two are compared with each other BIT FOR BIT on inputs built to hit the awkward cases:
exact ties (duplicated points), empty cells, out-of-range and NaN points, a padded tail that lands
thousands of points in one cell, every supported width."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu]

RNG = (-50.0, 50.0, -50.0, 50.0)


def _P(t):
    return ctypes.c_void_p(t.data_ptr())


def _inputs(B, N, C, seed, pad=0, dup=0, nan=0, sigma=40.0):
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(B, N, 4, generator=g) * torch.tensor([sigma, sigma, 2.0, 1.0])
    if pad:
        pts[:, N - pad:] = 0.0                       # zero padding: all in the cell that holds (0, 0)
    y = torch.randn(B * N, C, generator=g)
    if dup:
        src = torch.randint(0, N - pad - dup, (dup,), generator=g)
        pts[:, N - pad - dup:N - pad] = pts[:, src]  # same cell ...
        yv = y.view(B, N, C)
        yv[:, N - pad - dup:N - pad] = yv[:, src]    # ... and same features: exact ties on every channel
    if nan:
        pts[:, :nan, 0] = float("nan")
        pts[:, nan:2 * nan, 1] = float("inf")
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.2
    mean = torch.randn(C, generator=g) * 0.1
    invstd = torch.rand(C, generator=g) + 0.5
    return [t.cuda().contiguous() for t in (pts.view(B * N, 4), y, sc, sh, mean, invstd)]


def _sort(lib, pts, B, N, H, W):
    dev = pts.device
    row = torch.empty(B * N, device=dev, dtype=torch.int32)
    start = torch.empty(B * H * W + 1, device=dev, dtype=torch.int32)
    perm = torch.full((B * N,), -7, device=dev, dtype=torch.int32)
    nb = lib.kd_lidar_cell_sort_ws_bytes(B, N, H, W)
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    lib.call("kd_lidar_cell_sort", _P(pts), B, N, H, W, *RNG, _P(row), _P(start), _P(perm), _P(ws), nb, None)
    return row, start, perm


@pytest.mark.parametrize("B,N,H,W,pad,nan", [(1, 257, 4, 4, 0, 0), (3, 5000, 16, 16, 700, 40), (2, 9000, 64, 64, 0, 13),
                                             (2, 40, 32, 32, 0, 0), (5, 3001, 33, 17, 100, 5)])
def test_cell_sort_groups_the_in_range_points_by_row(B, N, H, W, pad, nan):
    from kdrt.lib import lib
    pts = _inputs(B, N, 64, 3, pad=pad, nan=nan)[0]
    row, start, perm = _sort(lib, pts, B, N, H, W)
    cell = torch.empty(B * N, device="cuda", dtype=torch.int32)
    lib.call("kd_lidar_bev_index", _P(pts), _P(cell), B * N, H, W, *RNG, None)
    torch.cuda.synchronize()
    cell, row, start, perm = cell.cpu().numpy(), row.cpu().numpy(), start.cpu().numpy(), perm.cpu().numpy()
    want_row = np.where(cell >= 0, (np.arange(B * N) // N) * (H * W) + cell, -1)
    assert np.array_equal(row, want_row)
    counts = np.bincount(want_row[want_row >= 0], minlength=B * H * W)
    assert start[0] == 0 and np.array_equal(np.diff(start), counts)
    nv = int(start[-1])
    assert nv == int((want_row >= 0).sum())
    used = perm[:nv]
    assert np.array_equal(np.sort(used), np.nonzero(want_row >= 0)[0])          # a permutation of the in-range ids
    assert np.array_equal(want_row[used], np.repeat(np.arange(B * H * W), counts))   # grouped by row, rows ascending
    assert np.all(perm[nv:] == -7)                                              # nothing written past the end


@pytest.mark.parametrize("B,N,H,W,pad,nan", [(1, 257, 4, 4, 0, 0), (3, 5000, 16, 16, 700, 40), (2, 9000, 64, 64, 0, 13),
                                             (2, 40, 32, 32, 0, 0), (5, 3001, 33, 17, 100, 5), (2, 1024, 8, 8, 0, 0),
                                             (1, 2049, 110, 110, 64, 3), (2, 3000, 128, 128, 200, 0),
                                             (1, 1500, 192, 192, 0, 2)])
def test_sort_points_is_the_stable_sort_by_frame_and_cell(B, N, H, W, pad, nan):
    from kdrt.lib import lib
    pts = _inputs(B, N, 64, 4, pad=pad, nan=nan)[0]
    spts = torch.full((B * N, 4), -5.0, device="cuda")
    srow = torch.full((B * N,), -9, device="cuda", dtype=torch.int32)
    start = torch.empty(B * H * W + 1, device="cuda", dtype=torch.int32)
    perm = torch.full((B * N,), -7, device="cuda", dtype=torch.int32)
    nb = lib.kd_lidar_sort_points_ws_bytes(B, N, H, W)
    ws = torch.empty(nb, device="cuda", dtype=torch.uint8)
    lib.call("kd_lidar_sort_points", _P(pts), B, N, H, W, *RNG, _P(spts), _P(srow), _P(start), _P(perm), _P(ws), nb, None)
    cell = torch.empty(B * N, device="cuda", dtype=torch.int32)
    lib.call("kd_lidar_bev_index", _P(pts), _P(cell), B * N, H, W, *RNG, None)
    torch.cuda.synchronize()
    cell = cell.cpu().numpy().astype(np.int64)
    frame = np.arange(B * N) // N
    key = np.where(cell >= 0, frame * (H * W) + cell, B * H * W + frame)     # out-of-range: after everything, by frame
    want = np.argsort(key, kind="stable")
    assert np.array_equal(perm.cpu().numpy(), want)
    # bit patterns (NaN coordinates must travel unchanged)
    assert np.array_equal(spts.cpu().numpy().view(np.int32), pts.cpu().numpy().view(np.int32)[want])
    want_row = np.where(cell >= 0, key, -1)[want]
    assert np.array_equal(srow.cpu().numpy(), want_row)
    counts = np.bincount(key[cell >= 0], minlength=B * H * W)
    assert np.array_equal(start.cpu().numpy(), np.concatenate([[0], np.cumsum(counts)]))


def test_sort_points_rejects_a_grid_too_large_for_its_histogram():
    from kdrt.lib import KDError, lib
    t = torch.zeros(64, 4, device="cuda")
    i = torch.zeros(200 * 200 + 1, device="cuda", dtype=torch.int32)
    with pytest.raises(KDError, match="bins exceed"):
        lib.call("kd_lidar_sort_points", _P(t), 1, 64, 200, 200, *RNG, _P(t), _P(i), _P(i), None, _P(i), 1 << 30, None)


@pytest.mark.parametrize("C", (64, 128, 256))
@pytest.mark.parametrize("B,N,H,W,pad,dup,nan", [(2, 3000, 16, 16, 300, 200, 20), (1, 500, 64, 64, 0, 50, 0),
                                                 (3, 20000, 64, 64, 2500, 1000, 7)])
def test_segmented_scatter_matches_atomic_scatter_bitwise(C, B, N, H, W, pad, dup, nan):
    from kdrt.lib import lib
    pts, y, sc, sh, mean, invstd = _inputs(B, N, C, 11 + C, pad=pad, dup=dup, nan=nan)
    ncells = B * H * W
    act = 1
    # atomic pair
    grid_a = torch.empty(ncells, C, device="cuda")
    lib.call("kd_lidar_scatter_max_fwd", _P(pts), _P(y), _P(sc), _P(sh), act, _P(grid_a), B, N, C, H, W, *RNG, None)
    dout = torch.randn(ncells, C, generator=torch.Generator().manual_seed(5)).cuda()
    G_a = torch.full((B * N, C), 3.0, device="cuda")
    rows_a = lib.kd_lidar_scatter_stat_rows(B * N, C)
    part_a = torch.empty(rows_a, 2, C, device="cuda")
    nb = lib.kd_lidar_scatter_bwd_ws_bytes(B, H, W, C)
    ws = torch.empty(nb, device="cuda", dtype=torch.uint8)
    lib.call("kd_lidar_scatter_max_bwd", _P(pts), _P(y), _P(sc), _P(sh), act, _P(grid_a), _P(dout), _P(mean), _P(invstd),
             _P(G_a), _P(part_a), B, N, C, H, W, *RNG, _P(ws), nb, None)
    # segmented pair
    row, start, perm = _sort(lib, pts, B, N, H, W)
    grid_s = torch.full((ncells, C), -1.0, device="cuda")          # every row must be overwritten, empty ones with 0
    lib.call("kd_lidar_seg_max_fwd", _P(y), _P(sc), _P(sh), act, _P(start), _P(perm), None, _P(grid_s), B * N, ncells, C, None)
    G_s = torch.full((B * N, C), 3.0, device="cuda")
    rows_s = lib.kd_lidar_seg_stat_rows(ncells)
    part_s = torch.empty(rows_s, 2, C, device="cuda")
    lib.call("kd_lidar_seg_max_bwd", _P(y), _P(sc), _P(sh), act, _P(grid_s), _P(dout), _P(mean), _P(invstd), _P(start), _P(perm),
             _P(row), _P(G_s), _P(part_s), B * N, ncells, C, None)
    torch.cuda.synchronize()
    assert torch.equal(grid_a.view(torch.int32), grid_s.view(torch.int32))
    assert torch.equal(G_a.view(torch.int32), G_s.view(torch.int32))
    # third form: rows physically sorted (kd_lidar_sort_points), perm = NULL
    spts = torch.empty_like(pts)
    srow = torch.empty(B * N, device="cuda", dtype=torch.int32)
    start2 = torch.empty(ncells + 1, device="cuda", dtype=torch.int32)
    perm2 = torch.empty(B * N, device="cuda", dtype=torch.int32)
    nb2 = lib.kd_lidar_sort_points_ws_bytes(B, N, H, W)
    ws2 = torch.empty(nb2, device="cuda", dtype=torch.uint8)
    lib.call("kd_lidar_sort_points", _P(pts), B, N, H, W, *RNG, _P(spts), _P(srow), _P(start2), _P(perm2), _P(ws2), nb2, None)
    ys = y[perm2.long()].contiguous()
    grid_p = torch.full((ncells, C), -1.0, device="cuda")
    lib.call("kd_lidar_seg_max_fwd", _P(ys), _P(sc), _P(sh), act, _P(start2), None, _P(srow), _P(grid_p), B * N, ncells, C, None)
    G_p = torch.full((B * N, C), 3.0, device="cuda")
    part_p = torch.empty(rows_s, 2, C, device="cuda")
    lib.call("kd_lidar_seg_max_bwd", _P(ys), _P(sc), _P(sh), act, _P(grid_p), _P(dout), _P(mean), _P(invstd), _P(start2), None,
             _P(srow), _P(G_p), _P(part_p), B * N, ncells, C, None)
    torch.cuda.synchronize()
    assert torch.equal(start2, start)
    assert torch.equal(grid_a.view(torch.int32), grid_p.view(torch.int32))
    assert torch.equal(G_a[perm2.long()].view(torch.int32), G_p.view(torch.int32))
    assert torch.allclose(part_p.double().sum(0), part_s.double().sum(0), rtol=1e-5, atol=1e-5 * float(part_s.double().sum(0).abs().max()))
    # ties really happened, and were split
    holders = (G_s != 0).sum().item()
    occupied = ((grid_s > 0) & (dout != 0)).sum().item()
    assert holders >= occupied > 0 and (dup == 0 or holders > occupied)
    sa, ss = part_a.double().sum(0), part_s.double().sum(0)
    assert torch.allclose(sa, ss, rtol=1e-5, atol=1e-5 * float(sa.abs().max()))
    # the stats are what they claim to be
    xhat = (y.double() - mean.double()) * invstd.double()
    want = torch.stack([G_s.double().sum(0), (G_s.double() * xhat).sum(0)])
    assert torch.allclose(ss, want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))


def test_segmented_scatter_is_run_to_run_deterministic():
    from kdrt.lib import lib
    B, N, H, W, C = 2, 20000, 64, 64, 128
    pts, y, sc, sh, mean, invstd = _inputs(B, N, C, 77, pad=1000, dup=500)
    dout = torch.randn(B * H * W, C, generator=torch.Generator().manual_seed(6)).cuda()
    outs = []
    for _ in range(3):
        row, start, perm = _sort(lib, pts, B, N, H, W)
        grid = torch.empty(B * H * W, C, device="cuda")
        lib.call("kd_lidar_seg_max_fwd", _P(y), _P(sc), _P(sh), 1, _P(start), _P(perm), None, _P(grid), B * N, B * H * W, C, None)
        G = torch.empty(B * N, C, device="cuda")
        part = torch.empty(lib.kd_lidar_seg_stat_rows(B * H * W), 2, C, device="cuda")
        lib.call("kd_lidar_seg_max_bwd", _P(y), _P(sc), _P(sh), 1, _P(grid), _P(dout), _P(mean), _P(invstd), _P(start), _P(perm),
                 _P(row), _P(G), _P(part), B * N, B * H * W, C, None)
        torch.cuda.synchronize()
        outs.append((grid.clone(), G.clone(), part.clone()))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))


def test_segmented_scatter_rejects_unsupported_width():
    from kdrt.lib import KDError, lib
    t = torch.zeros(64, device="cuda")
    i = torch.zeros(64, device="cuda", dtype=torch.int32)
    with pytest.raises(KDError, match="C must be 64, 128 or 256"):
        lib.call("kd_lidar_seg_max_fwd", _P(t), _P(t), _P(t), 1, _P(i), _P(i), None, _P(t), 1, 1, 96, None)


def test_gather_sorted_lists_the_in_range_points_in_cell_order():
    from kdrt.lib import lib
    B, N, H, W = 3, 7000, 32, 32
    pts = _inputs(B, N, 64, 9, pad=300, nan=11)[0]
    row, start, perm = _sort(lib, pts, B, N, H, W)
    out_pts = torch.full((B * N, 4), -5.0, device="cuda")
    out_row = torch.full((B * N,), -9, device="cuda", dtype=torch.int32)
    lib.call("kd_lidar_gather_sorted", _P(pts), _P(perm), _P(row), _P(start[B * H * W:]), _P(out_pts), _P(out_row), B * N, None)
    torch.cuda.synchronize()
    nv = int(start[-1])
    assert 0 < nv < B * N
    idx = perm[:nv].long()
    assert torch.equal(out_pts[:nv], pts[idx]) and torch.equal(out_row[:nv], row[idx])
    assert bool((out_row[:nv][1:] >= out_row[:nv][:-1]).all())
    assert bool((out_pts[nv:] == -5.0).all()) and bool((out_row[nv:] == -9).all())


@pytest.mark.parametrize("sigma", (40.0, 2.0))
@pytest.mark.parametrize("training", (False, True))
def test_lidar_encoder_same_bits_with_sorted_and_atomic_scatter(training, sigma):
    """The whole encoder (eval: compaction + fused scatter epilogue; train: forward AND parameter gradients)."""
    from kdrt import units
    from src.models.lidar_encoder import LiDAREncoder
    torch.manual_seed(3)
    enc = LiDAREncoder(encoder_type="spatial", grid_size=(32, 32)).cuda().train(training)
    # (NaN / Inf coordinates only in eval: in train mode they poison the batch statistics, in the reference too)
    # sigma = 2 m: a few cells hold more than a thousand points each (the chunked long-row kernels run)
    pts = _inputs(2, 6000, 64, 21, pad=500, dup=300, nan=0 if training else 9, sigma=sigma)[0].view(2, 6000, 4)
    res = {}
    saved = units._SCATTER_MODE, units._SCATTER_TABLES
    try:
        for mode in ("atomic", "ids", "sorted", "sorted_G"):
            units._SCATTER_MODE = mode.split("_")[0]
            units._SCATTER_TABLES = mode == "sorted"        # "sorted_G": sorted rows, gradient materialised as [points, C]
            units.clear_step_caches()
            enc.zero_grad()
            if training:
                y = enc(pts)
                (y * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
                # (conv biases in front of a BatchNorm have a zero true gradient: rounding noise, not compared)
                res[mode] = (y.detach().clone(), [p.grad.clone() for n, p in enc.named_parameters()
                                                  if not n.endswith(("point_mlp.0.bias", "point_mlp.3.bias", "point_mlp.6.bias"))])
            else:
                with torch.no_grad():
                    res[mode] = (enc(pts).clone(), [])
    finally:
        units._SCATTER_MODE, units._SCATTER_TABLES = saved
    # the table form of the scatter gradient rebuilds exactly the values the materialised form stores; only the
    # BatchNorm-backward sums of the 500-point padding cell are grouped differently (64-point chunks): rounding level
    assert torch.equal(res["sorted"][0].view(torch.int32), res["sorted_G"][0].view(torch.int32))
    for a, b in zip(res["sorted"][1], res["sorted_G"][1]):
        assert torch.allclose(a, b, rtol=2e-6, atol=2e-6 * float(a.abs().max()))
    # "ids" leaves the rows in place: same bits as the atomic form.  "sorted" permutes the rows of the point MLP, so in
    # train mode its BatchNorm statistics are summed in another order: same values to rounding.
    assert torch.equal(res["atomic"][0].view(torch.int32), res["ids"][0].view(torch.int32))
    if training:
        assert torch.allclose(res["atomic"][0], res["sorted"][0], rtol=1e-5, atol=1e-5 * float(res["atomic"][0].abs().max()))
    else:
        assert torch.equal(res["atomic"][0].view(torch.int32), res["sorted"][0].view(torch.int32))
    for mode in ("ids", "sorted"):
        for a, b in zip(res["atomic"][1], res[mode][1]):
            assert bool(torch.isfinite(a).all())
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(a.abs().max()))


@pytest.mark.parametrize("C", (64, 128))
@pytest.mark.parametrize("B,N,H,W,sigma,pad,dup", [(2, 6000, 16, 16, 40.0, 0, 300),      # ordinary scene: no long rows
                                                   (2, 9000, 32, 32, 3.0, 0, 500),       # a few cells hold ~1000 points
                                                   (3, 5000, 64, 64, 0.3, 0, 200),       # everything in 1-4 cells
                                                   (2, 3000, 16, 16, 40.0, 1500, 100),   # half the frame is zero padding
                                                   (1, 700, 8, 8, 12.0, 257, 0)])        # a 257-point row next to short ones
def test_share_table_path_matches_atomic_pair_on_concentrated_scenes(C, B, N, H, W, sigma, pad, dup):
    """Sorted rows + chunked handling of rows with more than 256 points (forward maxima, holder counts, shares,
    BatchNorm-backward sums): the grid and the gradient rebuilt from the tables equal the atomic pair's bit for bit."""
    from kdrt.lib import lib
    pts, y, sc, sh, mean, invstd = _inputs(B, N, C, 5 + C, pad=pad, dup=dup, sigma=sigma)
    ncells, P = B * H * W, B * N
    grid_a = torch.empty(ncells, C, device="cuda")
    lib.call("kd_lidar_scatter_max_fwd", _P(pts), _P(y), _P(sc), _P(sh), 1, _P(grid_a), B, N, C, H, W, *RNG, None)
    dout = torch.randn(ncells, C, generator=torch.Generator().manual_seed(5)).cuda()
    G_a = torch.empty(P, C, device="cuda")
    part_a = torch.empty(lib.kd_lidar_scatter_stat_rows(P, C), 2, C, device="cuda")
    nb = lib.kd_lidar_scatter_bwd_ws_bytes(B, H, W, C)
    ws = torch.empty(nb, device="cuda", dtype=torch.uint8)
    lib.call("kd_lidar_scatter_max_bwd", _P(pts), _P(y), _P(sc), _P(sh), 1, _P(grid_a), _P(dout), _P(mean), _P(invstd),
             _P(G_a), _P(part_a), B, N, C, H, W, *RNG, _P(ws), nb, None)
    spts, srow = torch.empty_like(pts), torch.empty(P, device="cuda", dtype=torch.int32)
    start, perm = torch.empty(ncells + 1, device="cuda", dtype=torch.int32), torch.empty(P, device="cuda", dtype=torch.int32)
    nb2 = lib.kd_lidar_sort_points_ws_bytes(B, N, H, W)
    ws2 = torch.empty(nb2, device="cuda", dtype=torch.uint8)
    lib.call("kd_lidar_sort_points", _P(pts), B, N, H, W, *RNG, _P(spts), _P(srow), _P(start), _P(perm), _P(ws2), nb2, None)
    ys = y[perm.long()].contiguous()
    grid_t = torch.full((ncells, C), -1.0, device="cuda")
    lib.call("kd_lidar_seg_max_fwd", _P(ys), _P(sc), _P(sh), 1, _P(start), None, _P(srow), _P(grid_t), P, ncells, C, None)
    share = torch.full((ncells, C), float("nan"), device="cuda")
    cnt = torch.full((ncells, C), float("nan"), device="cuda")
    part_t = torch.full((lib.kd_lidar_seg_share_stat_rows(ncells, P), 2, C), float("nan"), device="cuda")
    lib.call("kd_lidar_seg_share_bwd", _P(ys), _P(sc), _P(sh), 1, _P(grid_t), _P(dout), _P(mean), _P(invstd), _P(start), _P(srow),
             _P(share), _P(cnt), _P(part_t), P, ncells, C, None)
    torch.cuda.synchronize()
    counts = (start[1:] - start[:-1])
    if sigma < 10 or pad > 256:
        assert int(counts.max()) > 256                    # the chunked kernels really ran
    assert torch.equal(grid_a.view(torch.int32), grid_t.view(torch.int32))
    # the share table against the atomic pair's gradient: holders of (cell, channel) = its non-zero entries (dout != 0)
    rows = srow.long()
    ok = rows >= 0
    Gs = G_a[perm.long()]
    held = (Gs != 0) & ok[:, None]
    cnt_ref = torch.zeros(ncells, C, device="cuda")
    cnt_ref.index_add_(0, rows.clamp_min(0), held.float())
    nonempty = counts > 0
    want = torch.where(cnt_ref > 0, dout / cnt_ref.clamp_min(1), torch.zeros_like(dout))
    assert torch.equal(share[nonempty].view(torch.int32), want[nonempty].view(torch.int32))
    # ... and every holder's gradient is exactly its cell's share
    assert torch.equal(torch.where(held, share[rows.clamp_min(0)], torch.zeros_like(Gs)).view(torch.int32), Gs.view(torch.int32))
    assert bool(torch.isfinite(part_t).all())
    sa, st_ = part_a.double().sum(0), part_t.double().sum(0)
    assert torch.allclose(sa, st_, rtol=1e-5, atol=1e-5 * float(sa.abs().max()))
