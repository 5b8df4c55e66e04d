#!/bin/bash
# end-of-round validation on one MI355X: the full GPU suite, smoke(), the judged bench line, then the profiles behind the numbers
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2full
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/test_all.txt 2>&1
tail -n 4 $O/test_all.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
bash tools/r2_profile.sh > $O/profile.log 2>&1; tail -5 $O/profile.log
rm -rf gpurun_out/r2bert; bash tools/r2_bert_profile.sh > $O/bert_profile.log 2>&1
