#!/bin/bash
# A/B of GEMM builds: gemm tests with every tools/ab/dev_*.so / liblrp_*.so in turn, then tools/kbench.py --what onegemm (2 rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2gemm
mkdir -p $O
L=lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/product.so
: > $O/ab.txt
for f in tools/ab/dev_w4*.so; do cp $f $L; echo "== tests $(basename $f .so)" | tee -a $O/ab.txt; timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" 2>&1 | tail -n 1 | tee -a $O/ab.txt; done
for rep in 1 2; do
  for f in tools/ab/liblrp_*.so tools/ab/dev_w4*.so; do
    cp $f $L; echo "== $(basename $f .so)"; timeout 300 python tools/kbench.py --what onegemm 2>&1 | grep "gemm"
  done
done | tee -a $O/ab.txt
cp /tmp/product.so $L
