#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c11; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -s -k "bf16" > $O/test_bf16.txt 2>&1; echo "rc=$?"; grep -v "^E  \|amdgpu.ids" $O/test_bf16.txt | grep "bf16\|    \|passed\|failed" | head -40
