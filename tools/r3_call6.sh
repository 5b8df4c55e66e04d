#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c9
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or linear" > $O/pytest_gemm.txt 2>&1
echo "rc=$?" >> $O/pytest_gemm.txt; tail -n 15 $O/pytest_gemm.txt
timeout 300 python tools/skinny_bench.py > $O/skinny.txt 2>&1; cat $O/skinny.txt
timeout 300 python tools/gemm_ab.py --rounds 2 --iters 10 --no-check old=tools/ab/pp_v1c.so new=lrp-explains-transformers_amd/liblrp_hip.so > $O/gemm_ab.txt 2>&1; tail -8 $O/gemm_ab.txt
