#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p5
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -k "attention" > $O/test_attn.txt 2>&1
tail -n 25 $O/test_attn.txt
timeout 900 python -m pytest tests/test_baseline_size_gpu.py -q -m gpu -s -k "attention" > $O/test_attn_long.txt 2>&1
grep -aE "attention S=|passed|failed" $O/test_attn_long.txt | cut -c1-200
python tools/kbench.py --what attnb > $O/attn_v3.txt 2>&1
cat $O/attn_v3.txt
