"""Direct C-ABI checks of the GEMM family on awkward shapes (partial row / column / K tiles, M = 1, N = 4, K = 4, K not a
multiple of the 32-wide K-tile, strided operands), every prologue / epilogue, both arithmetics, against an fp64
reference computed with stock torch on the host.  Tolerance: 2e-5 of the result's largest magnitude."""
import itertools

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]
TOL = 2e-5

SHAPES = [(1, 4, 4), (37, 36, 20), (128, 32, 192), (300, 100, 132), (257, 192, 64), (1000, 16, 256), (513, 64, 16), (129, 768, 128)]


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
        rows = ops.lib.kd_pwconv_stat_rows(M)
        partial = torch.zeros(rows * 2 * N, device="cuda") if epi else None
        ops.pw_gemm(Ad, W.cuda(), C, M=M, K=K, N=N, pro=pro, pro_act=act, p=(sc.cuda(), sh.cuda(), None, None, None),
                    bias=bias.cuda() if use_bias else None, epi=epi, partial=partial)
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
    rows = ops.lib.kd_pwconv_stat_rows(M)
    part = torch.zeros(rows * 2 * K, device="cuda")
    Wt = ops.transpose(c(W))                                           # [K][N]
    ops.pw_gemm(c(G), Wt, gin, M=M, K=N, N=K, A2=c(Y), pro=2, pro_act=1, p=(c(al), c(be), c(ga), c(msc), c(msh)), epi=2,
                X=c(X), esc=c(esc), esh=c(esh), emean=c(mean), einv=c(inv), epi_act=1, partial=part)
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
