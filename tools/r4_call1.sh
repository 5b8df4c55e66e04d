#!/bin/bash
# round 4, GPU call 1: full GPU suite on the round-4 changes, the default bench (new keys), one-stabiliser-at-a-time on seeds (22,23)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c1; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -s > $O/test_all.txt 2>&1; echo "pytest rc=$?" >> $O/test_all.txt
tail -5 $O/test_all.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 900 python tools/explicit_site_sensitivity.py 22 23 > $O/sites_22_23.txt 2>&1; echo "sites rc=$?"
tail -12 $O/sites_22_23.txt
