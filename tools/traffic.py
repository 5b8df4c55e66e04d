#!/usr/bin/env python3
"""gpurun_out/<dir>/pmc_summary.json (tools/profile.sh: FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1 --layers 4`) ->
profiles/r06_gemm_traffic.json: fabric-side bytes per launch of the ping-pong GEMM instantiations next to their algorithmic bytes, STAMPED
with the sha256 of the kernel source they were measured on (bench.py reports the traffic only while that hash matches the tree's gemm_pp.hip)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "prof", "pmc_summary.json")
d = json.load(open(src))
M, out = 8192, {}


def alg(N, K):
    return 2 * (M * K + N * K + M * N)


def classify(name):
    """-> (nn, epi, skinny, rs) of a gemm_pp_kernel<bf16, NN, EPI, ACT, SK, LEAN, RS> row (mangled, or rocprof's half-demangled form), None otherwise"""
    m = re.search(r"gemm_pp_kernelIDF16bLb(\d)ELi(\d)ELi\dELb(\d)ELb\dELb(\d)E", name)
    if m:
        return int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4))
    m = re.search(r"gemm_pp_kernel<bool _Accum, bool, E, (\d), \d, (false|true), (?:false|true), (false|true)>", name)   # <bf16, true, ...>: NN
    if m:
        return 1, int(m.group(1)), int(m.group(2) == "true"), int(m.group(3) == "true")
    return None


def add(tag, key, algb):
    v = d[key]
    n = v["FETCH_SIZE"]["launches"]
    f = v["FETCH_SIZE"]["sum"] / n
    w = v["WRITE_SIZE"]["sum"] / v["WRITE_SIZE"]["launches"]
    t = f * 1024 * 2 + w * 1024
    out[tag] = {"kernel": key[:100], "launches": n, "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
                "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": algb, "ratio": t / algb}


keys = {classify(k): k for k in d if classify(k) is not None}
R = 2 * M * 4096                                   # one [M, 4096] bf16 operand of an epilogue (residual / residual gradient)
# one step of the 4-layer run (top layer: o-proj / MLP on one row per prompt; K1n parts fwd + bwd_qkv, the o-projection's dgrad with the 1/2 row
# scale): layer 0's qkv forward is the plain kernel, layers 1-3 take the row-scale + RoPE form; o-proj + down forward of layers 0-2 carry the residual +
# sum-of-squares epilogue; qkv dgrad x 4 and gate/up dgrad x 3 the residual one; o dgrad x 3 NN + row scale
add("nt_plain (qkv fwd, layer 0)", keys[(0, 0, 0, 0)], alg(6144, 4096))
if (0, 5, 0, 1) in keys:                       # QKV forward + row scale + RoPE in the epilogue (cos/sin table: 1 MB, not counted)
    add("nt_rowscale_rope (qkv fwd)", keys[(0, 5, 0, 1)], alg(6144, 4096))
else:
    add("nt_rowscale (qkv fwd)", keys[(0, 0, 0, 1)], alg(6144, 4096))
add("nt_residual_ssq (o-proj, down fwd)", keys[(0, 3, 0, 0)], (alg(4096, 4096) + alg(4096, 14336)) / 2 + R)
if (1, 0, 0, 0) in keys:                       # (gate/up dgrad as the plain kernel: ops.NORM_FUSION without "bwd_gu")
    add("nn_plain (gate/up dgrad)", keys[(1, 0, 0, 0)], alg(4096, 28672))
    add("nn_rowscale_residual (qkv dgrad)", keys[(1, 4, 0, 1)], alg(4096, 6144) + R)
else:                                          # default: 4 qkv dgrads + 3 gate/up dgrads per step carry the residual epilogue
    add("nn_rowscale_residual (qkv, gate/up dgrad)", keys[(1, 4, 0, 1)], (4 * alg(4096, 6144) + 3 * alg(4096, 28672)) / 7 + R)
add("nn_rowscale (o dgrad)", keys[(1, 0, 0, 1)], alg(4096, 4096))
add("gated_fwd", keys[(0, 1, 0, 1)], alg(28672, 4096) + 2 * M * 14336)
add("gated_bwd", keys[(1, 2, 0, 0)], 2 * (M * 4096 + 14336 * 4096) + 2 * 2 * M * 28672)
roof = [k for k in out if not k.startswith("gated")]          # the kernel set of bench.py's roofline key: every launch but the two gated ones
tt = sum(out[k]["traffic_bytes_per_launch"] * out[k]["launches"] for k in roof)
ta = sum(out[k]["algorithmic_bytes_per_launch"] * out[k]["launches"] for k in roof)
tn = sum(out[k]["launches"] for k in roof)
a_nt = a_nn = ta / tn
with open(os.path.join(ROOT, "lrp-explains-transformers_amd", "csrc", "gemm_pp.hip"), "rb") as f:
    sha = hashlib.sha256(f.read()).hexdigest()[:16]
res = {"gemm_pp_sha16": sha, "traffic_bytes_per_launch": tt / tn, "algorithmic_bytes_per_launch": (a_nt + a_nn) / 2,
       "ratio": tt / tn / ((a_nt + a_nn) / 2), "per_kernel": out,
       "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction: 128-B requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE x 1024; separate --pmc "
               "passes of `bench.py --steps 1 --warmup 1 --layers 4 --no-smallm --no-config5 --no-config4 --no-extra-modes` (tools/profile.sh); "
               "fabric-side bytes of the eight per-XCD L2s (Infinity-Cache hits included): every XCD fetches its own copy of the operand panels its 32 "
               "resident 256x256 tiles share (8 + 4 panels per 32 tiles), so ~2x the algorithmic bytes is the floor of this tiling.  gemm_pp_sha16 = "
               "sha256 of csrc/gemm_pp.hip at measurement time: bench.py nulls roofline.traffic when the tree's kernel differs."}
json.dump(res, open(os.path.join(ROOT, "profiles", "r06_gemm_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if k != "per_kernel" else {n: round(x["ratio"], 2) for n, x in v.items()}) for k, v in res.items() if k != "note"}))
