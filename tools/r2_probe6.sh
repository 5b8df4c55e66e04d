#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p6
mkdir -p $O
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_api_gpu.py tests/test_hf_gpu.py -q -m gpu -x > $O/test_engine.txt 2>&1
tail -n 6 $O/test_engine.txt
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.txt 2>&1
grep -v amdgpu $O/bench.txt | cut -c1-700
timeout 2400 python -m pytest tests/test_baseline_size_gpu.py -q -m gpu -s -k "not attention" > $O/test_baseline.txt 2>&1
grep -aE "H4096/S2048|drop-in H|passed|failed" $O/test_baseline.txt | grep -v "print\|f\"" | cut -c1-260
