#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3quick
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_baseline_size_gpu.py -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for r in 1 2; do
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 --no-smallm > $O/bench_x.json 2> $O/bench_x.err || tail -3 $O/bench_x.err
python - <<PY
import json
d=json.load(open("$O/bench_x.json")); r=d["roofline"]
print("bench:", round(d["value"],3), "expl/s", round(d["ms_per_step"],2), "ms/step | plain GEMM frac", round(r["frac"],4))
PY
done
