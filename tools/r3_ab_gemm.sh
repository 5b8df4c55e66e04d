#!/bin/bash
# A/B of GEMM builds under tools/ab/ against the product library (tools/gemm_ab.py): bash tools/r3_ab_gemm.sh name=lib.so ...
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3abg
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 500 python tools/gemm_ab.py --rounds 3 --iters 10 "$@" > $O/gemm_ab.txt 2>&1; grep -v "check x_" $O/gemm_ab.txt | tail -24
