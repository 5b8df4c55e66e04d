#!/bin/bash
# rocprofv3 kernel-trace of the headline engine under the lxt.explicit placement (8 layers, 4 prompts per step) -> per-kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-explprof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --mode explicit --layers 8 --steps 3 --warmup 1 --no-cpu-baseline --no-smallm --no-config5 --no-config4 --no-extra-modes --no-dropin > $O/bench_under_rocprof.json 2> $O/kt.log
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
head -40 $O/kernel_stats.txt | cut -c1-170
find $O -name "*.db" -delete; rm -rf $O/kt
