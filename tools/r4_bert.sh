#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4bert
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 500 python tools/bert_bench.py > $O/bert.txt 2>&1; grep -v amdgpu.ids $O/bert.txt | tail -34
