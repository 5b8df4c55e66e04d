#!/bin/bash
# round 3, GPU call 1: ping-pong GEMM (correctness, in-kernel timeline, interleaved A/B vs the round-2 library and hipBLASLt),
# the gemm tests on the new library, the explicit-BERT prompt set + padded batch, one short bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c1
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 420 python tools/gemm_ab.py --rounds 3 --iters 10 old=tools/ab/liblrp_hip_r02.so pp=lrp-explains-transformers_amd/liblrp_hip.so \
    noprio=tools/ab/pp_noprio.so tl=tools/ab/pp_tl.so > $O/gemm_ab.txt 2>&1
echo "gemm_ab rc=$?" >> $O/gemm_ab.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm or linear" > $O/pytest_gemm.txt 2>&1
echo "rc=$?" >> $O/pytest_gemm.txt
timeout 400 python -m pytest tests/test_bert_engine_gpu.py tests/test_hf_gpu.py -q -s -k "prompt_set or padded_batch or explicit_fp32" > $O/pytest_bert.txt 2>&1
echo "rc=$?" >> $O/pytest_bert.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "rc=$?" >> $O/bench.err
tail -n 40 $O/gemm_ab.txt; tail -n 3 $O/pytest_gemm.txt; tail -n 30 $O/pytest_bert.txt; cut -c1-600 $O/bench.json
