#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c19; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "stream or smallm or skinny" > $O/test.txt 2>&1; echo "tests rc=$?"; tail -3 $O/test.txt | cut -c1-300
timeout 300 python tools/stream_ab.py > $O/stream_ab.txt 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/stream_ab.txt | head -40
