#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c20
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
timeout 300 python tools/skinny_bench.py > $O/skinny.txt 2>&1; tail -40 $O/skinny.txt
