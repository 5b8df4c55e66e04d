#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p7
mkdir -p $O
LRP_GEMM_BIG=30 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "gemm or matmul or linear" > $O/test_gemm_m32.txt 2>&1
tail -n 4 $O/test_gemm_m32.txt
for cfg in 28 30 7 28 30; do
  echo "== LRP_GEMM_BIG=$cfg" >> $O/gemm.txt
  LRP_GEMM_BIG=$cfg python tools/kbench.py --what onegemm 2>&1 | grep "^gemm" >> $O/gemm.txt
done
cat $O/gemm.txt
