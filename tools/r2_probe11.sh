#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p11
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "smallm or linear_eps" > $O/test_smallm.txt 2>&1
tail -n 6 $O/test_smallm.txt
python tools/kbench.py --what stream > $O/stream.txt 2>&1
grep -v amdgpu $O/stream.txt | grep -v "^linear_eps"
