#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c21
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gemma3_engine_gpu.py tests/test_kernels_gpu.py -q -x -s -k "gemma3 or skinny" > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
