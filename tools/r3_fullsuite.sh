#!/bin/bash
# full -m gpu suite + smoke + the judged bench command (round 3)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r3full}
mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 2400 python -m pytest tests -q -m gpu --durations=15 ) > $O/test_all.txt 2>&1
tail -n 45 $O/test_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err; cut -c1-900 $O/bench.json
