"""Direct C-ABI checks of the GEMM family on awkward shapes (partial row / column / K tiles, M = 1, N = 4, K = 4, K not a
multiple of the 32-wide K-tile, strided operands), every prologue / epilogue, both arithmetics, against an fp64
reference computed with stock torch on the host.  Tolerance: 2e-5 of the result's largest magnitude."""
import itertools

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]
TOL = 2e-5

# (the shapes whose K and N are multiples of 32 also run the weight-resident streaming kernels in the split arithmetic:
# several column tiles, K chunks, tail slabs)
SHAPES = [(1, 4, 4), (37, 36, 20), (128, 32, 192), (300, 100, 132), (257, 192, 64), (1000, 16, 256), (513, 64, 16), (129, 768, 128),
          (1000, 64, 384), (777, 128, 128), (2100, 32, 32), (640, 64, 192), (333, 384, 64), (901, 128, 64), (450, 256, 128)]


def _act(z, act):
    return z if act == 0 else (z.clamp_min(0) if act == 1 else z.clamp(0, 6))


def _close(got, want, what):
    err = (got.double().cpu() - want).abs().max().item()
    scale = max(want.abs().max().item(), 1e-30)
    assert err <= TOL * scale, (what, err, scale)


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_forward_prologues_and_stats(M, K, N):
    from kdrt import ops
    g = torch.Generator().manual_seed(M * 131 + K * 17 + N)
    lda = K + 8                                                    # A lives inside a wider buffer
    Abuf = torch.randn(M, lda, generator=g)
    A = Abuf[:, :K]
    W, bias = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    sc, sh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.3
    Ad = Abuf.cuda()[:, :K]
    for pro, act, use_bias, epi in itertools.product((0, 1), (1, 2), (False, True), (0, 1)):
        if pro == 0 and act == 2:
            continue
        ref_in = A.double() if pro == 0 else _act(A.double() * sc.double() + sh.double(), act)
        want = ref_in @ W.double().t() + (bias.double() if use_bias else 0)
        C = torch.full((M, N + 4), 7.0, device="cuda")[:, :N]         # strided output; the pad must stay untouched
        rows = ops.lib.kd_pwconv_stat_rows_for(M, K, N, pro, epi, 0)     # rows the launch for this shape writes
        partial = torch.zeros(rows * 2 * N, device="cuda") if epi else None
        ops.pw_gemm(Ad, W.cuda(), C, M=M, K=K, N=N, pro=pro, pro_act=act, p=(sc.cuda(), sh.cuda(), None, None, None),
                    bias=bias.cuda() if use_bias else None, epi=epi, partial=partial, partial_rows=rows)
        _close(C, want, ("fwd", pro, act, use_bias, epi))
        assert torch.all(C._base[:, N:] == 7.0) if C._base is not None else True
        if epi:
            st = partial.view(rows, 2, N).double().sum(0).cpu()
            assert (st[0] - want.sum(0)).abs().max() <= 1e-4 * max(1.0, want.abs().sum(0).max().item())
            assert (st[1] - (want * want).sum(0)).abs().max() <= 1e-4 * max(1.0, (want * want).sum(0).max().item())


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_dgrad_and_wgrad(M, K, N):
    """dgrad: dX = (al*(G*mask) + be*Y + ga) . W  times act'(X*esc+esh), with the BN-backward sums; wgrad: dW = dYeff^T . Aeff."""
    from kdrt import ops
    g = torch.Generator().manual_seed(M * 7 + K * 3 + N * 11)
    G, Y = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / N ** 0.5
    al, be, ga = (torch.randn(N, generator=g) * 0.5 for _ in range(3))
    msc, msh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    esc, esh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    mean, inv = torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5
    d = lambda t: t.double()
    mask = ((d(Y) * d(msc) + d(msh)) > 0).double()
    dy = d(al) * (d(G) * mask) + d(be) * d(Y) + d(ga)
    zx = d(X) * d(esc) + d(esh)
    dx = (dy @ d(W)) * (zx > 0).double()
    c = lambda t: t.cuda()
    gin = torch.empty(M, K, device="cuda")
    rows = ops.lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, 0)           # (reduction width N, output width K)
    part = torch.zeros(rows * 2 * K, device="cuda")
    Wt = ops.transpose(c(W))                                           # [K][N]
    ops.pw_gemm(c(G), Wt, gin, M=M, K=N, N=K, A2=c(Y), pro=2, pro_act=1, p=(c(al), c(be), c(ga), c(msc), c(msh)), epi=2,
                X=c(X), esc=c(esc), esh=c(esh), emean=c(mean), einv=c(inv), epi_act=1, partial=part, partial_rows=rows)
    _close(gin, dx, "dgrad")
    st = part.view(rows, 2, K).double().sum(0).cpu()
    xhat = (d(X) - d(mean)) * d(inv)
    assert (st[0] - dx.sum(0)).abs().max() <= 1e-4 * max(1.0, dx.abs().sum(0).max().item())
    assert (st[1] - (dx * xhat).sum(0)).abs().max() <= 1e-4 * max(1.0, (dx * xhat).abs().sum(0).max().item())
    aeff = zx.clamp_min(0)
    dW = torch.empty(N, K, device="cuda")
    ops.pw_wgrad(c(G), c(X), dW, M=M, N=N, K=K, X=c(Y), d_mode=2, d_act=1, al=c(al), be=c(be), ga=c(ga), msc=c(msc), msh=c(msh),
                 a_mode=1, a_act=1, asc=c(esc), ash=c(esh))
    _close(dW, dy.t() @ aeff, "wgrad")


@pytest.mark.parametrize("M,N,K", [(5000, 192, 32), (4133, 384, 64), (3001, 768, 128), (2500, 64, 192), (2222, 64, 384), (1999, 128, 384),
                                   (3100, 128, 768), (1000, 128, 128), (900, 128, 64), (777, 64, 128), (1500, 128, 256), (1800, 256, 256),
                                   (650, 64, 256), (17, 384, 64), (40000, 384, 64)])
@pytest.mark.parametrize("d_mode,a_mode", [(0, 0), (2, 1), (2, 0), (0, 1)])
def test_role_specialised_wgrad_against_fp64_and_the_tiled_kernel(M, N, K, d_mode, a_mode):
    """kd_wgrad_rs.hip (one workgroup per CU owns a block of dW; vector waves convert, matrix waves multiply; column slices for
    the 768-wide layers; row slices with tail chunks) against an fp64 reference and against pw_wgrad_kernel."""
    from kdrt import ops
    if ops.get_gemm_arithmetic() != "split":
        pytest.skip("the role-specialised weight gradient exists in the split arithmetic only")
    g = torch.Generator().manual_seed(M + 7 * N + 13 * K + d_mode + 3 * a_mode)
    c = lambda t: t.cuda()
    G, Y, A = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    al, be, ga = (torch.randn(N, generator=g) * 0.5 for _ in range(3))
    msc, msh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    asc, ash = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    d = lambda t: t.double()
    dy = d(G) if d_mode == 0 else d(al) * (d(G) * ((d(Y) * d(msc) + d(msh)) > 0).double()) + d(be) * d(Y) + d(ga)
    aeff = d(A) if a_mode == 0 else (d(A) * d(asc) + d(ash)).clamp_min(0)
    want = dy.t() @ aeff

    def run():
        dW = torch.full((N, K), float("nan"), device="cuda")
        ops.pw_wgrad(c(G), c(A), dW, M=M, N=N, K=K, X=c(Y) if d_mode == 2 else None, d_mode=d_mode, d_act=1, al=c(al), be=c(be), ga=c(ga),
                     msc=c(msc), msh=c(msh), a_mode=a_mode, a_act=1, asc=c(asc), ash=c(ash))
        torch.cuda.synchronize()
        return dW
    prev = ops.lib.kd_set_wgrad_rs(0)
    try:
        tiled = run()
        ops.lib.kd_set_wgrad_rs(2)                   # every layer that has an instance, also those the default leaves to the tiled kernel
        rs = run()
    finally:
        ops.lib.kd_set_wgrad_rs(prev)
    _close(rs, want, "rs wgrad")
    _close(tiled, want, "tiled wgrad")
    assert (rs - tiled).abs().max().item() <= 2e-5 * want.abs().max().item()


def _both_forms(fn):
    """Run fn() with the tiled kernels, then with the streaming kernels; returns the two results."""
    from kdrt import ops
    prev = ops.lib.kd_set_gemm_stream(0)
    try:
        a = fn()
        ops.lib.kd_set_gemm_stream(2)
        b = fn()
    finally:
        ops.lib.kd_set_gemm_stream(prev)
    return a, b


@pytest.mark.parametrize("M,K,N", [(2100, 32, 32), (640, 64, 192), (777, 128, 128), (1000, 64, 384), (333, 384, 64), (129, 768, 128),
                                   (450, 256, 128), (5000, 192, 32), (1000, 32, 192), (700, 64, 384), (333, 64, 192), (4099, 64, 384)])
@pytest.mark.parametrize("epi,with_addend", [(0, False), (0, True), (2, False), (2, True)])
def test_streaming_dgrad_same_bits_as_tiled(M, K, N, epi, with_addend):
    """The streaming form of the data gradient (PRO2 operand from two streamed tensors, K chunks, several column tiles,
    residual gradient, activation mask + BatchNorm-backward sums) against the tiled kernel: the same bits in the result,
    the same sums up to summation order.  Split arithmetic only (the exact-fp32 mode has no streaming form)."""
    from kdrt import ops
    if ops.get_gemm_arithmetic() != "split":
        pytest.skip("streaming kernels exist in the split arithmetic only")
    g = torch.Generator().manual_seed(M + 3 * K + 5 * N + epi)
    c = lambda t: t.cuda()
    G, Y, X, add = (c(torch.randn(M, n, generator=g)) for n in (N, N, K, K))
    Wt = c(torch.randn(K, N, generator=g) / N ** 0.5)
    al, be, ga, msc, msh = (c(torch.randn(N, generator=g) * 0.5) for _ in range(5))
    esc, esh, mean, inv = (c(torch.rand(K, generator=g) + 0.5) for _ in range(4))

    def run():
        rows = ops.lib.kd_pwconv_stat_rows_for(M, N, K, 2, epi, int(with_addend))
        gin = torch.full((M, K), float("nan"), device="cuda")
        part = torch.zeros(rows * 2 * K, device="cuda") if epi == 2 else None
        ops.pw_gemm(G, Wt, gin, M=M, K=N, N=K, A2=Y, pro=2, pro_act=2, p=(al, be, ga, msc, msh), addend=add if with_addend else None,
                    epi=epi, X=X if epi == 2 else None, esc=esc, esh=esh, emean=mean, einv=inv, epi_act=2, partial=part,
                    partial_rows=rows)
        torch.cuda.synchronize()
        return gin, (part.view(rows, 2, K).double().sum(0) if epi == 2 else None), rows

    (g0, s0, r0), (g1, s1, r1) = _both_forms(run)
    if r1 == r0:
        pytest.skip("no streaming instance for this shape: the tiled kernel serves it in both modes")
    assert torch.equal(g0, g1)
    if epi == 2:
        scale = g0.double().abs().sum(0).max().item()
        assert (s0 - s1).abs().max().item() <= 1e-5 * max(scale, 1.0)


def test_streaming_lidar_l2_dgrad_same_bits_as_tiled():
    """kd_lidar_l2_dgrad (scatter-max gradient rebuilt from the per-cell tables on load, PRO4): streaming vs tiled."""
    from kdrt import ops
    from kdrt.ops import lib, P, stream
    if ops.get_gemm_arithmetic() != "split":
        pytest.skip("streaming kernels exist in the split arithmetic only")
    g = torch.Generator().manual_seed(77)
    c = lambda t: t.cuda()
    M, C0, C1, C2, cells = 4133, 64, 128, 128, 300
    Y2, Y1 = c(torch.randn(M, C2, generator=g)), c(torch.randn(M, C1, generator=g))
    rows_t = torch.randint(-1, cells, (M,), generator=g, dtype=torch.int32).sort().values.cuda()
    sc2, sh2 = c(torch.rand(C2, generator=g) + 0.5), c(torch.randn(C2, generator=g) * 0.2)
    # cell maxima consistent with Y2 so that a fair share of rows hold a maximum
    v2 = torch.clamp_min(Y2 * sc2 + sh2, 0)
    grid = torch.zeros(cells, C2, device="cuda")
    ok = rows_t >= 0
    grid.index_reduce_(0, rows_t[ok].long(), v2[ok], "amax", include_self=True)
    share = c(torch.randn(cells, C2, generator=g))
    al, be, ga = (c(torch.randn(C2, generator=g) * 0.5) for _ in range(3))
    Wt2 = c(torch.randn(C1, C2, generator=g) / C2 ** 0.5)
    sc1, sh1, mean1, inv1 = (c(torch.rand(C1, generator=g) + 0.5) for _ in range(4))

    def run_l2():
        rows = lib.kd_lidar_l2_dgrad_stat_rows(M, C2, C1)
        G1 = torch.full((M, C1), float("nan"), device="cuda")
        part = torch.zeros(rows * 2 * C1, device="cuda")
        lib.call("kd_lidar_l2_dgrad", P(Y2), C2, P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Wt2), P(G1), C1,
                 P(Y1), C1, P(sc1), P(sh1), P(mean1), P(inv1), 1, P(part), rows, M, C2, C1, stream())
        torch.cuda.synchronize()
        return G1, part.view(rows, 2, C1).double().sum(0), rows

    (a0, s0, r0), (a1, s1, r1) = _both_forms(run_l2)
    assert r0 != r1 and torch.equal(a0, a1)
    assert (s0 - s1).abs().max().item() <= 1e-5 * max(a0.double().abs().sum(0).max().item(), 1.0)


def test_statistics_slab_sized_for_the_other_kernel_form_is_refused():
    """Round 2's systematic 3e-4 gradient error came from a launch that silently took the tiled kernel
    after the caller had sized -- and later reduced -- its BatchNorm-statistics slab for the streaming form.  The C ABI
    now carries the caller's row count: a slab sized for the OTHER form is an argument error, for the forward
    statistics (epi 1), the data-gradient sums (epi 2) and the LiDAR table-form data gradient alike."""
    from kdrt import ops
    from kdrt.lib import KDError
    from kdrt.ops import lib, P, stream
    if ops.get_gemm_arithmetic() != "split":
        pytest.skip("the two kernel forms exist in the split arithmetic only")
    M, K, N = 4096 * 3, 64, 128
    g = torch.Generator().manual_seed(5)
    A, W = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    Cout = torch.empty(M, N, device="cuda")
    prev = lib.kd_set_gemm_stream(2)
    try:
        r_stream = lib.kd_pwconv_stat_rows_for(M, K, N, 0, 1, 0)
        lib.kd_set_gemm_stream(0)
        r_tiled = lib.kd_pwconv_stat_rows_for(M, K, N, 0, 1, 0)
        assert r_tiled == (M + 127) // 128 and r_stream != r_tiled
        part = torch.zeros(max(r_stream, r_tiled) * 2 * N, device="cuda")
        for mode, good, bad in ((0, r_tiled, r_stream), (2, r_stream, r_tiled)):
            lib.kd_set_gemm_stream(mode)
            ops.pw_gemm(A, W, Cout, M=M, K=K, N=N, epi=1, partial=part, partial_rows=good)          # the matching count runs
            for wrong in (bad, good - 1, 0):
                with pytest.raises(KDError, match="statistics slab"):
                    ops.pw_gemm(A, W, Cout, M=M, K=K, N=N, epi=1, partial=part, partial_rows=wrong)
        # data gradient through an activation (epi 2): reduction width 128, output width 128 (an instance the dispatcher selects)
        K = 128
        W = torch.randn(N, K, generator=g).cuda()
        G, Y, X = (torch.randn(M, n, generator=g).cuda() for n in (N, N, K))
        v = lambda n: torch.rand(n, generator=g).cuda() + 0.5
        al, be, ga, msc, msh, esc, esh, mean, inv = v(N), v(N), v(N), v(N), v(N), v(K), v(K), v(K), v(K)
        Wt = ops.transpose(W)
        gin = torch.empty(M, K, device="cuda")
        lib.kd_set_gemm_stream(2)
        rs = lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, 0)
        lib.kd_set_gemm_stream(0)
        rt = lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, 0)
        assert rs != rt
        part2 = torch.zeros(max(rs, rt) * 2 * K, device="cuda")
        kw = dict(M=M, K=N, N=K, A2=Y, pro=2, pro_act=1, p=(al, be, ga, msc, msh), epi=2, X=X, esc=esc, esh=esh, emean=mean,
                  einv=inv, epi_act=1, partial=part2)
        with pytest.raises(KDError, match="statistics slab"):
            ops.pw_gemm(G, Wt, gin, partial_rows=rs, **kw)            # tiled launch, slab sized for the streaming form
        ops.pw_gemm(G, Wt, gin, partial_rows=rt, **kw)
        lib.kd_set_gemm_stream(2)
        with pytest.raises(KDError, match="statistics slab"):
            ops.pw_gemm(G, Wt, gin, partial_rows=rt, **kw)            # and the other way round
        ops.pw_gemm(G, Wt, gin, partial_rows=rs, **kw)
        # instances that hipcc can only build with scratch are never selected: the same launch WITH a residual gradient takes
        # the tiled kernel in both modes (csrc/kd_gemm_stream.hip: stream_cfg)
        assert lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, 1) == rt
        torch.cuda.synchronize()
    finally:
        lib.kd_set_gemm_stream(prev)


@pytest.mark.parametrize("M", [20, 4133, 256 * 32 * 2 + 77, 300000])
def test_lidar_l2_fused_backward_against_the_two_kernels_and_fp64(M):
    """kd_lidar_l2_bwd (csrc/kd_lidar_bwd.hip: data + weight gradient of the last point-MLP layer in one kernel, one read and
    one split of Y2 / Y1): G1 has the bits of kd_lidar_l2_dgrad, the BatchNorm-backward sums and dW2 agree with
    kd_lidar_l2_dgrad / kd_lidar_l2_wgrad up to summation order, and dW2 is within 2e-5 of a float64 evaluation."""
    from kdrt import ops
    from kdrt.ops import lib, P, stream
    if ops.get_gemm_arithmetic() != "split":
        assert not lib.kd_lidar_l2_bwd_supported(128, 128)
        pytest.skip("the one-kernel backward exists in the split arithmetic only")
    assert lib.kd_lidar_l2_bwd_supported(128, 128) and not lib.kd_lidar_l2_bwd_supported(64, 128)
    g = torch.Generator().manual_seed(M)
    c = lambda t: t.cuda()
    C1 = C2 = 128
    cells = max(4, M // 9)
    Y2, Y1 = c(torch.randn(M, C2, generator=g)), c(torch.randn(M, C1, generator=g))
    rows_t = torch.randint(-1, cells, (M,), generator=g, dtype=torch.int32).sort().values.cuda()
    sc2, sh2 = c(torch.rand(C2, generator=g) + 0.5), c(torch.randn(C2, generator=g) * 0.2)
    v2 = torch.clamp_min(Y2 * sc2 + sh2, 0)
    grid = torch.zeros(cells, C2, device="cuda")
    ok = rows_t >= 0
    grid.index_reduce_(0, rows_t[ok].long(), v2[ok], "amax", include_self=True)
    share = c(torch.randn(cells, C2, generator=g))
    al, be, ga = (c(torch.randn(C2, generator=g) * 0.5) for _ in range(3))
    W2 = c(torch.randn(C2, C1, generator=g) / C2 ** 0.5)
    Wt2 = ops.transpose(W2)                                           # [C1][C2]
    sc1, sh1, mean1, inv1 = (c(torch.rand(C1, generator=g) + 0.5) for _ in range(4))
    sh1 = sh1 - 1.0                                                   # about half the pre-activations below zero

    # the two kernels
    rows_d = lib.kd_lidar_l2_dgrad_stat_rows(M, C2, C1)
    G1a = torch.full((M, C1), float("nan"), device="cuda")
    part_a = torch.zeros(rows_d * 2 * C1, device="cuda")
    lib.call("kd_lidar_l2_dgrad", P(Y2), C2, P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Wt2), P(G1a), C1,
             P(Y1), C1, P(sc1), P(sh1), P(mean1), P(inv1), 1, P(part_a), rows_d, M, C2, C1, stream())
    dWa = torch.empty(C2, C1, device="cuda")
    nb = lib.kd_pwconv_wgrad_ws_bytes(M, C2, C1)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    lib.call("kd_lidar_l2_wgrad", P(Y2), C2, P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Y1), C1, P(sc1),
             P(sh1), 1, P(dWa), M, C2, C1, P(ws), nb, stream())
    # one kernel
    rows_f = lib.kd_lidar_l2_bwd_stat_rows(M)
    G1b = torch.full((M + 3, C1), float("nan"), device="cuda")         # rows beyond M must stay untouched
    part_b = torch.zeros(rows_f * 2 * C1, device="cuda")
    dWb = torch.empty(C2, C1, device="cuda")
    nb2 = lib.kd_lidar_l2_bwd_ws_bytes(M, C2, C1)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    args = (P(Y2), C2, P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Wt2), P(G1b), C1, P(Y1), C1, P(sc1), P(sh1),
            P(mean1), P(inv1), 1, P(part_b))
    lib.call("kd_lidar_l2_bwd", *args, rows_f, P(dWb), M, C2, C1, P(ws2), nb2, stream())
    torch.cuda.synchronize()
    assert torch.equal(G1a, G1b[:M]) and bool(torch.isnan(G1b[M:]).all())
    sa, sb = part_a.view(rows_d, 2, C1).double().sum(0), part_b.view(rows_f, 2, C1).double().sum(0)
    assert (sa - sb).abs().max().item() <= 1e-5 * max(G1a.double().abs().sum(0).max().item(), 1.0)
    # float64 reference of the weight gradient
    d = lambda t: t.double()
    v = torch.clamp_min(d(Y2) * d(sc2) + d(sh2), 0)
    rl = rows_t.long().clamp_min(0)
    Gs = torch.where(ok[:, None] & (v.float() > 0) & (v.float() == grid[rl]), d(share[rl]), torch.zeros((), dtype=torch.float64, device="cuda"))
    dy = d(al) * Gs + d(be) * d(Y2) + d(ga)
    a1 = torch.clamp_min(d(Y1) * d(sc1) + d(sh1), 0)
    want = dy.t() @ a1
    scale = want.abs().max().item()
    assert (dWb.double() - want).abs().max().item() <= 2e-5 * scale, ((dWb.double() - want).abs().max().item(), scale)
    assert (dWb - dWa).abs().max().item() <= 1e-5 * scale
    # a slab sized for another launch is refused; so is an unsupported shape
    from kdrt.lib import KDError
    with pytest.raises(KDError, match="statistics slab"):
        lib.call("kd_lidar_l2_bwd", *args, rows_f + 1, P(dWb), M, C2, C1, P(ws2), nb2, stream())
    with pytest.raises(KDError, match="no instance"):
        lib.call("kd_lidar_l2_bwd", *args, rows_f, P(dWb), M, C2, 64, P(ws2), nb2, stream())


@pytest.mark.parametrize("M", [20, 4133, 256 * 32 * 2 + 77, 300000])
def test_lidar_l1_fused_backward_against_the_two_kernels_and_fp64(M):
    """kd_lidar_l1_bwd (csrc/kd_lidar_bwd.hip: layer 1's weight gradient, BatchNorm-0 backward sums and the G0 * point moments in
    one kernel, layer 0 recomputed from the points) against kd_lidar_l1_dgrad(G0 = NULL, moments) + kd_lidar_l1_wgrad and
    against a float64 evaluation."""
    from kdrt import ops
    from kdrt.ops import lib, P, stream
    if ops.get_gemm_arithmetic() != "split":
        assert not lib.kd_lidar_l1_bwd_supported(128, 64)
        pytest.skip("the one-kernel backward exists in the split arithmetic only")
    assert lib.kd_lidar_l1_bwd_supported(128, 64) and not lib.kd_lidar_l1_bwd_supported(128, 128)
    g = torch.Generator().manual_seed(M + 1)
    c = lambda t: t.cuda()
    N1, K0 = 128, 64
    G1, Y1 = c(torch.randn(M, N1, generator=g)), c(torch.randn(M, N1, generator=g))
    pts = c(torch.randn(M, 4, generator=g) * torch.tensor([20.0, 20.0, 2.0, 0.3]))
    w0, b0 = c(torch.randn(K0, 4, generator=g) * 0.1), c(torch.randn(K0, generator=g) * 0.1)
    al, be, ga = (c(torch.randn(N1, generator=g) * 0.5) for _ in range(3))
    W1 = c(torch.randn(N1, K0, generator=g) / N1 ** 0.5)
    Wt1 = ops.transpose(W1)                                           # [K0][N1]
    sc0, sh0, mean0, inv0 = c(torch.rand(K0, generator=g) + 0.5), c(torch.randn(K0, generator=g) * 0.3), c(torch.randn(K0, generator=g) * 0.1), c(torch.rand(K0, generator=g) + 0.5)

    # the two kernels (moments form: G0 is not stored)
    rows_d = lib.kd_lidar_l1_dgrad_stat_rows(M, N1, K0)
    part_a = torch.zeros(rows_d * 2 * K0, device="cuda")
    m1a = torch.empty(4, K0, device="cuda")
    nbm = lib.kd_lidar_l1_dgrad_ws_bytes(M, K0)
    wsm = torch.empty(nbm, dtype=torch.uint8, device="cuda")
    lib.call("kd_lidar_l1_dgrad", P(G1), N1, P(Y1), N1, P(al), P(be), P(ga), None, None, 0, P(Wt1), None, K0, P(pts), P(w0), P(b0),
             P(sc0), P(sh0), P(mean0), P(inv0), 1, P(part_a), rows_d, P(m1a), P(wsm), nbm, M, N1, K0, stream())
    dWa = torch.empty(N1, K0, device="cuda")
    nb = lib.kd_pwconv_wgrad_ws_bytes(M, N1, K0)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    lib.call("kd_lidar_l1_wgrad", P(G1), N1, P(Y1), N1, 0, P(al), P(be), P(ga), None, None, P(pts), P(w0), P(b0), P(sc0), P(sh0), 1,
             P(dWa), M, N1, K0, P(ws), nb, stream())
    # one kernel
    rows_f = lib.kd_lidar_l1_bwd_stat_rows(M)
    part_b = torch.zeros(rows_f * 2 * K0, device="cuda")
    m1b = torch.empty(4, K0, device="cuda")
    dWb = torch.empty(N1, K0, device="cuda")
    nb2 = lib.kd_lidar_l1_bwd_ws_bytes(M, N1, K0)
    ws2 = torch.empty(nb2, dtype=torch.uint8, device="cuda")
    args = (P(G1), N1, P(Y1), N1, P(al), P(be), P(ga), P(Wt1), P(pts), P(w0), P(b0), P(sc0), P(sh0), P(mean0), P(inv0), 1, P(part_b))
    lib.call("kd_lidar_l1_bwd", *args, rows_f, P(m1b), P(dWb), M, N1, K0, P(ws2), nb2, stream())
    torch.cuda.synchronize()
    # float64 reference
    d = lambda t: t.double()
    dy = d(al) * d(G1) + d(be) * d(Y1) + d(ga)
    x0 = d(pts) @ d(w0).t() + d(b0)
    z0 = x0 * d(sc0) + d(sh0)
    a0 = torch.clamp_min(z0, 0)
    G0 = (dy @ d(W1)) * (z0 > 0)
    want_dW = dy.t() @ a0
    want_s = torch.stack([G0.sum(0), (G0 * (x0 - d(mean0)) * d(inv0)).sum(0)])
    want_m1 = d(pts).t() @ G0
    sa, sb = part_a.view(rows_d, 2, K0).double().sum(0), part_b.view(rows_f, 2, K0).double().sum(0)
    tol = lambda ref: 2e-5 * max(ref.abs().max().item(), 1e-6) * max(1.0, (M / 4096) ** 0.5)
    assert (sb - want_s).abs().max().item() <= tol(want_s), ((sb - want_s).abs().max().item(), want_s.abs().max().item())
    assert (sb - sa).abs().max().item() <= tol(want_s)
    assert (m1b.double() - want_m1).abs().max().item() <= tol(want_m1)
    assert (m1b - m1a).abs().max().item() <= tol(want_m1)
    assert (dWb.double() - want_dW).abs().max().item() <= 2e-5 * want_dW.abs().max().item()
    assert (dWb - dWa).abs().max().item() <= 1e-5 * want_dW.abs().max().item()
    from kdrt.lib import KDError
    with pytest.raises(KDError, match="statistics slab"):
        lib.call("kd_lidar_l1_bwd", *args, rows_f + 1, P(m1b), P(dWb), M, N1, K0, P(ws2), nb2, stream())


@pytest.mark.parametrize("M,N,K", [(3000, 128, 256), (2049, 384, 64), (1777, 128, 768)])
def test_role_specialised_wgrad_on_column_slices_of_wider_buffers(M, N, K):
    """D / X / A as column slices of wider row-major buffers (row stride != width), as the fusion block hands its operands
    over (both projections live in one [M, 256] buffer): the role-specialised kernel must honour ldd / ldx / lda."""
    from kdrt import ops
    if ops.get_gemm_arithmetic() != "split":
        pytest.skip("the role-specialised weight gradient exists in the split arithmetic only")
    g = torch.Generator().manual_seed(M + N + K)
    Gb, Yb, Ab = torch.randn(M, N + 64, generator=g), torch.randn(M, 2 * N, generator=g), torch.randn(M, K + 32, generator=g)
    G, Y, A = Gb[:, 32:32 + N], Yb[:, N:], Ab[:, 16:16 + K]
    al, be, ga = (torch.randn(N, generator=g) * 0.5 for _ in range(3))
    msc, msh = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g) * 0.2
    asc, ash = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    d = lambda t: t.double()
    dy = d(al) * (d(G) * ((d(Y) * d(msc) + d(msh)) > 0).double()) + d(be) * d(Y) + d(ga)
    want = dy.t() @ (d(A) * d(asc) + d(ash)).clamp_min(0)
    c = lambda t: t.cuda()
    Gd, Yd, Ad = c(Gb)[:, 32:32 + N], c(Yb)[:, N:], c(Ab)[:, 16:16 + K]
    prev = ops.lib.kd_set_wgrad_rs(2)
    try:
        dW = torch.full((N, K), float("nan"), device="cuda")
        ops.pw_wgrad(Gd, Ad, dW, M=M, N=N, K=K, X=Yd, d_mode=2, d_act=1, al=c(al), be=c(be), ga=c(ga), msc=c(msc), msh=c(msh),
                     a_mode=1, a_act=1, asc=c(asc), ash=c(ash))
    finally:
        ops.lib.kd_set_wgrad_rs(prev)
    _close(dW, want, "rs wgrad, strided operands")
