#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p2
mkdir -p $O
nproc > $O/host.txt; free -g >> $O/host.txt
timeout 900 python -m pytest tests/test_api_gpu.py tests/test_hf_gpu.py -q -m gpu -x -s > $O/test_api_hf.txt 2>&1
echo "rc=$?" >> $O/test_api_hf.txt
timeout 2400 python -m pytest tests/test_baseline_size_gpu.py -q -m gpu -s > $O/test_baseline.txt 2>&1
echo "rc=$?" >> $O/test_baseline.txt
tail -n 30 $O/test_api_hf.txt
grep -E "^\[|passed|failed|rc=|Error|error" $O/test_baseline.txt | tail -60
