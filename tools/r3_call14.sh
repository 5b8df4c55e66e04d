#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c19
mkdir -p $O
export PYTHONUNBUFFERED=1
for m in 2048 4096; do
timeout 400 python tools/gemm_ab.py --m $m --rounds 3 --iters 20 --no-check ours=lrp-explains-transformers_amd/liblrp_hip.so > $O/gemm_m$m.txt 2>&1; tail -12 $O/gemm_m$m.txt
done
