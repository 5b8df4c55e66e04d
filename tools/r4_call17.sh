#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c17; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
FP32_CHECK=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/dp -o dp -- python $GRAFT_REPO_ROOT/tools/llama_dropin_bench.py > $O/run.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/dp -name "*.db" | head -1) > $O/kernels.txt 2>&1; head -45 $O/kernels.txt | cut -c1-190
