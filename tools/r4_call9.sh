#!/bin/bash
# round 4, GPU call 9: tail-split dispatch for 257..511-tile GEMMs (test, A/B, Gemma-3 bench)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c9; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "tail_split or skinny or big_m" > $O/test.txt 2>&1; echo "tests rc=$?"; tail -6 $O/test.txt | cut -c1-300
timeout 300 python tools/tail_split_ab.py > $O/ab.txt 2>&1; echo "ab rc=$?"; cat $O/ab.txt
timeout 600 python -m pytest tests/test_gemma3_engine_gpu.py tests/test_gemma3_mm_engine_gpu.py -m gpu -x -q > $O/test_gemma.txt 2>&1; echo "gemma tests rc=$?"; tail -3 $O/test_gemma.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c9/bench.json").read().strip().splitlines()[-1])
print("headline", p["value"], {k: v for k, v in p.items() if k.startswith("config4")})
PY
