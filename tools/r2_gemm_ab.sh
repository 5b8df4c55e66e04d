#!/bin/bash
# A/B of GEMM builds: tests with the product library first, then tools/kbench.py --what onegemm with every tools/ab/liblrp_*.so
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2gemm
mkdir -p $O
L=lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/product.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" > $O/test_gemm.txt 2>&1; tail -n 6 $O/test_gemm.txt
for rep in 1 2; do
  for f in tools/ab/liblrp_*.so; do
    cp $f $L; echo "== $(basename $f .so)"; timeout 300 python tools/kbench.py --what onegemm 2>&1 | grep "gemm"
  done
done | tee $O/ab.txt
cp /tmp/product.so $L
if [ -f tools/ab/dev_w4_timeline.so ]; then cp tools/ab/dev_w4_timeline.so $L; python tools/gemm_w4_timeline.py | tee $O/timeline.txt; cp /tmp/product.so $L; fi
