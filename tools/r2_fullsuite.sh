#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2full
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -x > $O/test_all.txt 2>&1
tail -n 8 $O/test_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
