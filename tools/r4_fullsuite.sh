#!/bin/bash
# end-of-round validation as the driver runs it: pytest -m gpu, smoke(), default bench
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4full}; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q -s > $O/test_all.txt 2>&1; echo "pytest rc=$?" >> $O/test_all.txt; tail -4 $O/test_all.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 $O/smoke.txt
( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; cat $O/bench_time.txt
