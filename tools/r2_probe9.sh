#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p9
mkdir -p $O
for cfg in 30 28 30 28; do
  echo "== LRP_GEMM_BIG=$cfg" >> $O/gemm.txt
  LRP_GEMM_BIG=$cfg python tools/kbench.py --what onegemm 2>&1 | grep "^gemm" >> $O/gemm.txt
done
cat $O/gemm.txt
