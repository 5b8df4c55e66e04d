#!/bin/bash
# round 4, GPU call 5: d = 256 attention after the lazy-rescale / mask refinements (tests + timing), Gemma-3 keys
O=$GRAFT_REPO_ROOT/gpurun_out/r4c5; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemma3_engine_gpu.py tests/test_gemma3_mm_engine_gpu.py -m gpu -x -q -s -k "attention or gemma3" > $O/test_attn.txt 2>&1; echo "tests rc=$?"; tail -3 $O/test_attn.txt
timeout 300 python tools/attn_shape_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
timeout 900 python bench.py --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c5/bench.json").read().strip().splitlines()[-1])
print("headline", p["value"]); print("config4 text", {k:v for k,v in p["config4_gemma3_4b_text"].items() if k!="workload"})
print("config4 image+text", {k:v for k,v in p.get("config4_gemma3_4b_image_text",{}).items() if k!="workload"})
PY
