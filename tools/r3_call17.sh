#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c22
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python tools/gemma3_engine_bench.py > $O/g3.txt 2>&1; tail -8 $O/g3.txt
cd /tmp && export TMPDIR=/tmp
G3_TEXT_LAYERS=6 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python $GRAFT_REPO_ROOT/tools/gemma3_engine_bench.py > $GRAFT_REPO_ROOT/$O/g3_prof.txt 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/g3_kernel_stats.txt 2>&1; head -24 $O/g3_kernel_stats.txt | cut -c1-200
find $O -name "*.db" -delete; rm -rf $O/kt
