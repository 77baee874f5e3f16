"""Dev diagnostic: BN-backward sums (dgrad EPI2 partial slab -> kd_bn_bwd_finalize) for M = 8192 vs 16384 rows."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
import torch
from kdrt import ops
from kdrt.ops import BNC
for M in (8192, 16384, 16384 + 77):
    N, K = 64, 128
    g = torch.Generator().manual_seed(M)
    G, Y, X = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / 8
    al, be, ga = (torch.randn(N, generator=g) * 0.5 for _ in range(3))
    esc, esh = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    mean, inv = torch.randn(K, generator=g) * 0.1, torch.rand(K, generator=g) + 0.5
    d = lambda t: t.double()
    dy = d(al) * d(G) + d(be) * d(Y) + d(ga)
    zx = d(X) * d(esc) + d(esh)
    dx = (dy @ d(W)) * (zx > 0).double()
    xhat = (d(X) - d(mean)) * d(inv)
    c = lambda t: t.cuda()
    gin = torch.empty(M, K, device="cuda")
    rows = ops.lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, 0)
    part = torch.zeros(rows * 2 * K, device="cuda")
    Wt = ops.transpose(c(W))
    ops.pw_gemm(c(G), Wt, gin, M=M, K=N, N=K, A2=c(Y), pro=2, pro_act=0, p=(c(al), c(be), c(ga), None, None), epi=2,
                X=c(X), esc=c(esc), esh=c(esh), emean=c(mean), einv=c(inv), epi_act=1, partial=part, partial_rows=rows)
    st = part.view(rows, 2, K).double().sum(0).cpu()
    bnc = BNC(K, "cuda"); bnc.mean.copy_(c(mean)); bnc.invstd.copy_(c(inv))
    dgamma, dbeta, abg, _ = ops.bn_bwd_finalize(part, rows, K, M, torch.ones(K, device="cuda"), bnc, True)
    r1, r2 = dx.sum(0), (dx * xhat).sum(0)
    e = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max()).item()
    print(M, "rows", rows, "slab s1", e(st[0], r1), "s2", e(st[1], r2), "| finalize dbeta", e(dbeta, r1), "dgamma", e(dgamma, r2))
