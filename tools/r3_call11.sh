#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c15
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/gemm_ab.py --rounds 3 --iters 10 tailwait=tools/ab/pp_tailwait.so nowait=lrp-explains-transformers_amd/liblrp_hip.so > $O/gemm_ab.txt 2>&1; tail -14 $O/gemm_ab.txt
