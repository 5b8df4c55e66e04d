#!/usr/bin/env python3
"""profiles/r03_pmc_summary.json (tools/r3_profile.sh: FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 1 --layers 4`) ->
profiles/r03_gemm_traffic.json: fabric-side bytes per launch of the ping-pong GEMM instantiations next to their algorithmic bytes."""
import json
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
d = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
M, out = 8192, {}


def alg(N, K):
    return 2 * (M * K + N * K + M * N)


def add(name, key, algb):
    v = d[key]
    n = v["FETCH_SIZE"]["launches"]
    f = v["FETCH_SIZE"]["sum"] / n
    w = v["WRITE_SIZE"]["sum"] / v["WRITE_SIZE"]["launches"]
    t = f * 1024 * 2 + w * 1024
    out[name] = {"kernel": key[:90], "launches": n, "fetch_size_kib_per_launch": f, "write_size_kib_per_launch": w,
                 "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": algb, "ratio": t / algb}


def find(sub):
    return [k for k in d if sub in k][0]


# one step of the 4-layer run (the top layer's o-proj / MLP run on one row per prompt): NT plain = 4 qkv + 3 o + 3 down forward,
# NN plain = 3 gate/up dgrad + 3 o dgrad + 4 qkv dgrad
a_nt = (4 * alg(6144, 4096) + 3 * alg(4096, 4096) + 3 * alg(4096, 14336)) / 10
a_nn = (3 * alg(4096, 28672) + 3 * alg(4096, 4096) + 4 * alg(4096, 6144)) / 10
add("plain_nt", find("gemm_pp_kernelIDF16bLb0ELi0ELi0"), a_nt)
add("plain_nn", find("gemm_pp_kernel<bool _Accum, bool, E, 0, 0>"), a_nn)
add("gated_fwd", find("gemm_pp_kernelIDF16bLb0ELi1ELi0"), alg(28672, 4096) + 2 * M * 14336)
add("gated_bwd", find("gemm_pp_kernel<bool _Accum, bool, E, 2, 0>"), 2 * (M * 4096 + 14336 * 4096) + 2 * 2 * M * 28672)
tt = sum(out[k]["traffic_bytes_per_launch"] * out[k]["launches"] for k in ("plain_nt", "plain_nn"))
tn = sum(out[k]["launches"] for k in ("plain_nt", "plain_nn"))
res = {"traffic_bytes_per_launch": tt / tn, "algorithmic_bytes_per_launch": (a_nt + a_nn) / 2, "ratio": tt / tn / ((a_nt + a_nn) / 2),
       "per_kernel": out,
       "note": "FETCH_SIZE x 1024 x 2 (gfx950 correction: 128-B requests tallied at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE x 1024; separate --pmc "
               "passes of `bench.py --steps 1 --warmup 1 --layers 4 --no-smallm --no-config5` (tools/r3_profile.sh); fabric-side bytes of the eight "
               "per-XCD L2s (Infinity-Cache hits included): every XCD fetches its own copy of the operand panels its 32 resident 256x256 tiles "
               "share (8 + 4 panels per 32 tiles), so ~2x the algorithmic bytes is the floor of this tiling"}
json.dump(res, open(os.path.join(ROOT, "profiles", "r03_gemm_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if k != "per_kernel" else {n: round(x["ratio"], 2) for n, x in v.items()}) for k, v in res.items() if k != "note"}))
