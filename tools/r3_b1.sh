#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b1
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -q -x -k "skinny or gemm or llama_bf16 or batch or graph" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for extra in "--batch 1" "--batch 1 --graph" "--batch 2"; do
  timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 $extra > $O/bench_x.json 2> $O/bench_x.err
  python - "$extra" <<PY
import json, sys
d=json.load(open("$O/bench_x.json")); r=d["roofline"]
print("bench", sys.argv[1], ":", round(d["value"],3), "expl/s", round(d["ms_per_step"],2), "ms/step | plain GEMM frac", r["frac"], "| all", r["with_fused_epilogue_launches"].get("frac"), r["with_fused_epilogue_launches"].get("splitk"))
PY
done
