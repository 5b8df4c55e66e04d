#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c9; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/gemma_gemm_insitu.py > $O/insitu.txt 2>&1; echo "rc=$?"; cat $O/insitu.txt
