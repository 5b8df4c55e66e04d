#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q2
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -k "argmax or llama_fp32 or dense_seed or bf16" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
