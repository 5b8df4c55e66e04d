#!/bin/bash
# Dev tool (GPU box): Gemma-3 engines -- tests, then bench.py's config-4 keys (text, image + text)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5gemma}
mkdir -p $O
timeout 1500 python -m pytest tests/test_gemma3_engine_gpu.py tests/test_gemma3_mm_engine_gpu.py -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-smallm --no-extra-modes --no-dropin $2 > $O/b.json 2> $O/b.err
python - <<PY
import json
for ln in open("$O/b.json"):
    if ln.startswith("{"): d = json.loads(ln)
print("headline", round(d["value"], 2))
for k in ("config4_gemma3_4b_text", "config4_gemma3_4b_image_text"):
    v = d[k]; print(k, round(v["value"], 2), "expl/s", round(v["ms_per_step"], 1), "ms", {a: round(b, 3) for a, b in v.items() if a.startswith("gemm") or a.startswith("host")})
PY
