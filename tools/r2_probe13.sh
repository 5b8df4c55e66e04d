#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p13
mkdir -p $O
timeout 900 python -m pytest tests/test_hf_gpu.py -q -m gpu -s -k "explicit" > $O/test_bert_explicit.txt 2>&1
grep -aE "bert-base explicit|passed|failed|Error" $O/test_bert_explicit.txt | grep -v "print" | cut -c1-300
timeout 1500 python tools/explicit_forward_error.py > $O/fwd_err.txt 2>&1
grep -v amdgpu $O/fwd_err.txt | tail -20
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -m gpu -x > $O/test_rest.txt 2>&1; tail -3 $O/test_rest.txt
