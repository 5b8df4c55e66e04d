#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c11
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_baseline_size_gpu.py tests/test_api_gpu.py -q -x > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-400 $O/bench.json
