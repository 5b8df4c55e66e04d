#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p10
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gamma.py -q -m gpu -k "smallm or linear_eps or gamma" > $O/test_smallm.txt 2>&1
tail -n 15 $O/test_smallm.txt
python tools/kbench.py --what stream > $O/stream.txt 2>&1
grep -v amdgpu $O/stream.txt
timeout 900 python -m pytest tests/test_engine_gpu.py -q -m gpu -x > $O/test_engine.txt 2>&1; tail -3 $O/test_engine.txt
