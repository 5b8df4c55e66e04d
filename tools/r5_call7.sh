#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c7; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "test_linear_stream_dgrad" 2>&1 | grep -E "^E|assert|Error|passed|failed" | head -30 | tee $O/tests.txt
