#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3s2
export PYTHONUNBUFFERED=1
timeout 500 python tools/two_stream_probe.py > gpurun_out/r3s2/out.txt 2>&1; grep -v amdgpu.ids gpurun_out/r3s2/out.txt | tail -6
