import sys, torch
sys.path.insert(0, ".")
import lxt_amd.ops as ops
bf = torch.bfloat16
g_ = torch.Generator().manual_seed(1)
for (M, N, K) in ((4096, 4096, 1024), (8192, 4096, 1024), (8192, 4096, 256), (16384, 4096, 256)):
    x = torch.randn(M, K, generator=g_).to(bf).cuda(); W = (torch.randn(N, K, generator=g_) * K ** -0.5).to(bf).cuda(); res = torch.randn(M, N, generator=g_).to(bf).cuda()
    for rep in range(3):
        o = torch.full((M, N), float("nan"), dtype=bf, device="cuda"); ssq = torch.full((N // 64, M), float("nan"), device="cuda")
        ops.gemm_res_ssq(x, W, res, o, ssq)
        ref = (res.double() + x.double() @ W.double().T)
        bad = ((o.double() - ref).abs() > 0.1) | torch.isnan(o)
        nb = int(bad.sum())
        idx = bad.nonzero()[:8].tolist()
        sref = (o.double() ** 2).view(M, N // 64, 64).sum(-1).T
        sbad = (~torch.isclose(ssq.double(), sref, rtol=1e-4, atol=1e-5))
        print(M, N, K, "rep", rep, "bad out", nb, idx, "bad ssq", int(sbad.sum()), sbad.nonzero()[:6].tolist())
