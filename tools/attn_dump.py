"""dump / compare the bf16 d=128 attention kernels' outputs on fixed seeded inputs (A/B of two builds of liblrp_hip.so):
   python tools/attn_dump.py save /tmp/ref.pt      (with build A in place)
   python tools/attn_dump.py cmp  /tmp/ref.pt      (with build B in place)"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from lxt_amd import ops

CASES = [(2, 300, 4, 2, True, 0, 0.0), (1, 192, 4, 1, True, 0, 0.0), (2, 200, 2, 2, False, 0, 0.0), (1, 520, 4, 2, True, 100, 0.0),
         (2, 300, 4, 2, True, 0, 1e-8), (1, 2048, 8, 2, True, 0, 0.0), (1, 777, 2, 1, True, 0, 1e-8)]


def run(case):
    B, S, Hq, Hkv, causal, window, eps = case
    d = 128
    g = torch.Generator(device="cuda").manual_seed(S * 7 + Hq)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g).bfloat16()
    q, k, v, Go = r(B * S, Hq * d), r(B * S, Hkv * d), r(B * S, Hkv * d), r(B * S, Hq * d)
    o, lse = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
    ops.attn_fwd(q, k, v, None, o, lse, B, S, Hq, Hkv, d, d ** -0.5, causal, window)
    Gho, D = torch.empty_like(q), torch.empty(B, Hq, S, device="cuda")
    ops.attn_bwd_prep(Go, o, Gho, D, B, S, Hq, d, 1e-6 if eps else 0.0, 0.5)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    ops.attn_bwd_dq(q, k, v, None, Gho, lse, D, dq, B, S, Hq, Hkv, d, d ** -0.5, eps, eps, causal, window)
    ops.attn_bwd_dkv(q, k, v, None, Gho, None, lse, D, dk, dv, B, S, Hq, Hkv, d, d ** -0.5, eps, eps, causal, window)
    torch.cuda.synchronize()
    return dict(o=o.float().cpu(), lse=lse.cpu(), dq=dq.float().cpu(), dk=dk.float().cpu(), dv=dv.float().cpu())


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    out = [run(c) for c in CASES]
    if mode == "save":
        torch.save(out, path)
    else:
        ref = torch.load(path)
        bad = 0
        for c, a, b in zip(CASES, out, ref):
            errs = {n: float((a[n] - b[n]).abs().max() / b[n].abs().max().clamp_min(1e-30)) for n in a}
            nan = {n: bool(torch.isnan(a[n]).any()) for n in a}
            flag = any(e > 2e-2 for e in errs.values()) or any(nan.values())
            bad += flag
            print(("BAD " if flag else "ok  ") + str(c), {n: f"{e:.1e}" for n, e in errs.items()}, "nan" if any(nan.values()) else "", flush=True)
        print("RESULT", "FAIL" if bad else "PASS")
