#!/bin/bash
# round 5, GPU call 10: 16-byte output stores in the attention kernels (T21): parity + in-situ A/B against the previous commit's library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c10; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_gemma3_engine_gpu.py -x -q -k "attention or golden or fixture or left_padded or gemma3" 2>&1 | tail -4 | tee $O/tests.txt
tools/r5_ab.sh r5c10 prev intree prev intree 2>&1 | grep -E "^===|attn32" | tee $O/ab.txt
