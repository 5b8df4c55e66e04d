"""dev: sustained GEMM loop (cold or hot operands, ours or hipBLASLt) while sampling rocm-smi clocks / power"""
import os, subprocess, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lxt_amd.ops as ops
M, N, K = 8192, 28672, 4096
mode = sys.argv[1]
if mode == "hot":
    a = torch.randn(1, K, device="cuda").bfloat16().expand(M, K); b = torch.randn(1, K, device="cuda").bfloat16().expand(N, K)
else:
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
fn = (lambda: torch.matmul(a, b.T, out=out)) if mode == "blaslt" else (lambda: ops.gemm_nt_2d(a, b, out))
for _ in range(5): fn()
torch.cuda.synchronize()
t0 = time.time(); n = 0; samples = []
while time.time() - t0 < 5.0:
    for _ in range(200): fn()
    n += 200
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout
    samples.append(r.strip().splitlines()[-1] if r.strip() else "?")
    torch.cuda.synchronize()
dt = time.time() - t0
print(mode, f"{2*M*N*K*n/dt/1e12:.0f} TF/s sustained (incl. smi sampling gaps)")
hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
print(hdr[0] if hdr else "")
for s in samples[:: max(1, len(samples) // 6)]: print(s)
