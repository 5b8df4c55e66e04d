#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c16; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hf_gpu.py -m gpu -q -s -k "8b_dims" > $O/test.txt 2>&1; echo "rc=$?"; grep "llama 8B\|passed\|failed\|Error" $O/test.txt | cut -c1-400
