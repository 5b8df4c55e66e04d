#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q3
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_gemma3_engine_gpu.py -q -k "gemma3" -s > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
