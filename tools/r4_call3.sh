#!/bin/bash
# round 4, GPU call 3: fused image + text Gemma-3 driver tests, stream forward v2 (tests + A/B), default bench with the new keys
O=$GRAFT_REPO_ROOT/gpurun_out/r4c3; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemma3_mm_engine_gpu.py -m gpu -x -q -s > $O/test_mm.txt 2>&1; echo "mm tests rc=$?"; grep -E "gemma3|passed|failed|Error|error" $O/test_mm.txt | tail -12
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "stream or skinny or chunk" > $O/test_stream.txt 2>&1; echo "stream tests rc=$?"; tail -3 $O/test_stream.txt
timeout 300 python tools/stream_ab.py > $O/stream_ab.txt 2>&1; cat $O/stream_ab.txt
timeout 300 python tools/stream_ab.py 128256 4096 > $O/stream_ab_head.txt 2>&1; cat $O/stream_ab_head.txt
timeout 900 python bench.py --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes > $O/bench_mm.json 2> $O/bench_mm.err; echo "bench rc=$?"; tail -3 $O/bench_mm.err
python - <<'PY'
import json,os
try:
    p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c3/bench_mm.json").read().strip().splitlines()[-1])
    print("headline", p["value"]); print("config4 text", {k:v for k,v in p["config4_gemma3_4b_text"].items() if k!="workload"})
    print("config4 image+text", {k:v for k,v in p.get("config4_gemma3_4b_image_text",{}).items() if k!="workload"})
except Exception as e: print("no bench line", e)
PY
