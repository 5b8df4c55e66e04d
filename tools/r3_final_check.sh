#!/bin/bash
# round-3 closing checks that are not part of r3_fullsuite.sh: the BERT tests on the cached oracle, explicit-mode and B = 1 bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3final
mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_bert_engine_gpu.py tests/test_hf_gpu.py -q -x -k "bert" --durations=8 ) > $O/pytest_bert.txt 2>&1; tail -16 $O/pytest_bert.txt
for extra in "--mode explicit" "--batch 1" "--batch 1 --graph"; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 $extra > $O/bench_x.json 2> $O/bench_x.err
  python - "$extra" <<PY
import json, sys
d=json.load(open("$O/bench_x.json")); r=d["roofline"]
print("bench", sys.argv[1], ":", round(d["value"],3), "expl/s", round(d["ms_per_step"],2), "ms/step | plain GEMM frac", r["frac"])
PY
done
