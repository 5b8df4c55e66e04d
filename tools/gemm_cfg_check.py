"""dev: correctness of a forced GEMM variant (LRP_GEMM_TILE) against torch.matmul in fp32 accumulate"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lxt_amd.ops as ops
torch.manual_seed(0)
for (M, N, K) in [(512, 768, 256), (1000, 520, 384), (2048, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm_nt_2d(a, b, out)
    ref = a.float() @ b.float().T
    print("cfg", os.environ.get("LRP_GEMM_TILE"), (M, N, K), "max err", float((out.float() - ref).abs().max()), flush=True)
