#!/bin/bash
# round-2 probe 1: time the attention knobs committed untimed in round 1, S=4096 and B=1 bench lines
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p1
mkdir -p $O
python tools/kbench.py --what attnb > $O/attn_default.txt 2>&1
LRP_ATTN_DQ_QS=2 python tools/kbench.py --what attnb > $O/attn_qs2.txt 2>&1
LRP_ATTN_HOIST=1 python tools/kbench.py --what attnb > $O/attn_hoist.txt 2>&1
LRP_ATTN_DQ_QS=2 LRP_ATTN_HOIST=1 python tools/kbench.py --what attnb > $O/attn_qs2_hoist.txt 2>&1
LRP_ATTN_HOIST=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k attn > $O/test_hoist.txt 2>&1
timeout 600 python bench.py --seq 4096 --batch 2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_s4096.txt 2>&1
timeout 600 python bench.py --batch 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_b1.txt 2>&1
tail -n 8 $O/*.txt
