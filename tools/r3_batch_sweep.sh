#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3sweep
mkdir -p $O
export PYTHONUNBUFFERED=1
for b in 4 8 6 4; do
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 --no-smallm --batch $b > $O/bench_x.json 2> $O/bench_x.err || tail -3 $O/bench_x.err
  python - "$b" <<PY
import json, sys
d=json.load(open("$O/bench_x.json")); r=d["roofline"]
print("batch", sys.argv[1], ":", round(d["value"],3), "expl/s", round(d["ms_per_step"],2), "ms/step | plain GEMM frac", round(r["frac"],4))
PY
done
