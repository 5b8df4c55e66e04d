"""dev: run one GEMM shape a few times (for rocprofv3 --pmc passes).  usage: gemm_one.py M N K [hot]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lxt_amd.ops as ops
M, N, K = (int(v) for v in sys.argv[1:4])
hot = len(sys.argv) > 4 and sys.argv[4] == "hot"
if hot:
    a = torch.randn(1, K, device="cuda").bfloat16().expand(M, K)
    b = torch.randn(1, K, device="cuda").bfloat16().expand(N, K)
else:
    a = torch.randn(M, K, device="cuda").bfloat16()
    b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(6):
    ops.gemm_nt_2d(a, b, out)
torch.cuda.synchronize()
