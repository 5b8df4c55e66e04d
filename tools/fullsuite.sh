#!/bin/bash
# end-of-round validation as the driver runs it: pytest -m gpu, smoke(), default bench
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-full}; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( time timeout 2400 python -m pytest tests -m gpu -q -s > $O/test_all.txt 2>&1 ) 2> $O/test_time.txt; echo "pytest rc=$?" >> $O/test_all.txt; tail -4 $O/test_all.txt; cat $O/test_time.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?"; tail -5 $O/smoke.txt
( time timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt; echo "bench rc=$?"; cat $O/bench_time.txt; tail -c 600 $O/bench_default.err
