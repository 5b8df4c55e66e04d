#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3g3
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_gemma3_engine_gpu.py -q -k "skinny or gemma3" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config5 --no-smallm > $O/bench_x.json 2> $O/bench_x.err || tail -3 $O/bench_x.err
python - <<PY
import json
d=json.load(open("$O/bench_x.json"))
print("headline", round(d["value"],2), "| config4:", d["config4_gemma3_4b_text"])
PY
