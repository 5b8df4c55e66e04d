"""CPU: a property of the COMPILED ping-pong GEMM (csrc/gemm_pp.hip) that no numerics test sees.  The K loop's staging pipeline lives on
hand-counted vmcnt waits; the compiler's own wait insertion must not add one at the top of the loop (it drains the LDS-DMA loads that were
issued 5-6 intervals ahead).  Round 6 found exactly that in one instantiation (row-scale loads of the K1n dgrad epilogue re-using the K loop's
fragment registers: -4 % on the K = 28672 gate/up dgrad), round 5's fused down-projection dgrad had carried the same; csrc/gemm_pp.hip: PP_VMWAIT.
hipcc cross-compiles the device code to assembly here (no GPU)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SRC = os.path.join(ROOT, "lrp-explains-transformers_amd", "csrc", "gemm_pp.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc")
def test_gemm_pp_k_loop_carries_only_the_counted_waits(tmp_path):
    out = str(tmp_path / "gemm_pp.s")
    subprocess.run([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                    "--cuda-device-only", "-S", SRC, "-o", out], check=True, capture_output=True, timeout=900)
    txt = open(out).read()
    kernels = [f for f in re.split(r"\n(?=_ZN\S+:)", txt) if re.match(r"_ZN\S*gemm_pp_kernel\S*:", f)]
    assert len(kernels) >= 15
    checked = 0
    for f in kernels:
        name = f.split(":", 1)[0]
        L = f.split("\n")
        assert not any("ScratchSize" in ln and not ln.strip().endswith(": 0") for ln in L), f"{name}: register spills"
        if re.search(r"gemm_pp_kernelI\w+?Lb[01]ELi\d+ELi\d+ELb1E", name):
            continue                                     # the skinny (SK) instantiations predicate their MFMAs: another loop shape
        hdr = [i for i, ln in enumerate(L) if "Inner Loop Header: Depth=2" in ln]
        assert len(hdr) == 1, name
        # the loop's blocks carry "in Loop: Header=<header block>" in the assembler's comments; the compiler may place part of the body (the
        # landing pad of the back edge) AHEAD of the header: body = first in-loop block ... last back edge into a block at or before the header
        hlab = next(re.match(r"\.(LBB\d+_\d+):", L[j].strip()).group(1) for j in range(hdr[0], hdr[0] - 3, -1) if re.match(r"\.LBB\d+_\d+:", L[j].strip()))
        hline = next(j for j in range(hdr[0], hdr[0] - 3, -1) if L[j].strip().startswith("." + hlab + ":"))
        inloop = {i: re.match(r"(\.LBB\d+_\d+):", ln.strip()).group(1) for i, ln in enumerate(L)
                  if re.match(r"\.LBB\d+_\d+:", ln.strip()) and ("Header=" + hlab[1:] + " " in ln or i == hline)}
        start = min(inloop)
        entry = [lb for i, lb in inloop.items() if i <= hline]
        end = max(i for i, ln in enumerate(L) if i > hline and any(re.search(r"s_c?branch\S*\s+" + re.escape(lb) + r"\s*$", ln.split(";")[0].rstrip()) for lb in entry))
        labs = entry
        body = [ln.split(";")[0].strip() for ln in L[start:end + 1]]
        body = [b for b in body if b and not b.startswith(".") and not b.startswith(";;")]
        mfma = sum(b.startswith("v_mfma") for b in body)
        assert mfma == 128, (name, mfma, labs)           # two K tiles of 2 x 32 MFMAs per iteration
        first_read = next(i for i, b in enumerate(body) if b.startswith("ds_read"))
        assert not any(b.startswith("s_waitcnt") and "vmcnt" in b for b in body[:first_read]), f"{name}: compiler wait at the top of the K loop"
        waits = [b for b in body if b.startswith("s_waitcnt") and "vmcnt" in b]
        # per K tile: vmcnt(8) after each of the two staging issues + the two alternatives of the peeled last tiles
        assert len(waits) <= 8 and all(re.fullmatch(r"s_waitcnt vmcnt\((8|2|0)\)", w) for w in waits), (name, waits)
        checked += 1
    assert checked >= 12
