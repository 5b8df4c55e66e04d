#!/bin/bash
# round 4, GPU call 2: d = 256 attention kernels (correctness + timing), Gemma-3 tests and bench, stream-vs-skinny A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r4c2; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "attention" > $O/test_attn.txt 2>&1; echo "attn tests rc=$?"; tail -4 $O/test_attn.txt
timeout 900 python -m pytest tests/test_gemma3_engine_gpu.py tests/test_hf_gpu.py tests/test_baseline_size_gpu.py -m gpu -x -q -s -k "gemma3 or attention or model_family" > $O/test_g3.txt 2>&1; echo "gemma tests rc=$?"; tail -4 $O/test_g3.txt
timeout 300 python tools/attn_shape_bench.py > $O/attn_bench.txt 2>&1; cat $O/attn_bench.txt
timeout 300 python tools/stream_ab.py > $O/stream_ab.txt 2>&1; cat $O/stream_ab.txt
timeout 300 python tools/stream_ab.py 128256 4096 > $O/stream_ab_head.txt 2>&1; cat $O/stream_ab_head.txt
timeout 600 python bench.py --no-cpu-baseline --no-smallm --no-config5 --no-extra-modes > $O/bench_g3.json 2> $O/bench_g3.err; echo "bench rc=$?"
python - <<'PY'
import json,os
p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c2/bench_g3.json").read().strip().splitlines()[-1])
print("headline", p["value"], "config4", {k:v for k,v in p["config4_gemma3_4b_text"].items() if k!="workload"})
PY
