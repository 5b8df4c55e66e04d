#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p12
mkdir -p $O
timeout 900 python -m pytest tests/test_hf_gpu.py -q -m gpu -s -k "explicit" > $O/test_bert_explicit.txt 2>&1
grep -aE "bert-base explicit|passed|failed|Error" $O/test_bert_explicit.txt | cut -c1-300
timeout 1200 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_kernels_gpu.py tests/test_gamma.py -q -m gpu -x > $O/test_rest.txt 2>&1; tail -4 $O/test_rest.txt
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.txt 2>&1
grep -v amdgpu $O/bench.txt | cut -c1-3000
