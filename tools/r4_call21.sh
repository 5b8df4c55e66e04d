#!/bin/bash
# rocprofv3 kernel stats of the Gemma-3 probes (config 4 text + image) -- the headline step is kept to 1 + 1 steps
O=$GRAFT_REPO_ROOT/gpurun_out/r4c21; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/g3 -o g3 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes --no-dropin > $O/bench.json 2> $O/err.txt; echo rc=$?
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/g3 -name "*.db" | head -1) > $O/kernels.txt 2>&1; head -50 $O/kernels.txt | cut -c1-200
