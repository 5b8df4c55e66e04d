#!/bin/bash
# round 4, GPU call 14: tile ranges derived from per-row key intervals (all six attention32 kernels)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c14; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or attn" > $O/test_attn.txt 2>&1; echo "attn tests rc=$?"; tail -4 $O/test_attn.txt | cut -c1-300
timeout 300 python tools/attn_shape_bench.py > $O/attn_bench.txt 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids $O/attn_bench.txt
timeout 1200 python -m pytest tests/test_gemma3_mm_engine_gpu.py tests/test_gemma3_engine_gpu.py tests/test_hf_gpu.py tests/test_api_gpu.py tests/test_engine_gpu.py tests/test_bert_engine_gpu.py -m gpu -q > $O/test_eng.txt 2>&1; echo "engine tests rc=$?"; tail -4 $O/test_eng.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c14/bench.json").read().strip().splitlines()[-1])
print("headline", p["value"])
for k in ("config4_gemma3_4b_text","config4_gemma3_4b_image_text"):
    print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in p[k].items() if a!='workload'})
PY
