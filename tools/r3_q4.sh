#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3q4
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_gemma3_engine_gpu.py -q -k "head_rmsnorm or gemma3" -s > $O/pytest.txt 2>&1; grep -v "^$\|Warning\|warnings\|float(\|detach\|Docs" $O/pytest.txt | tail -12
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-smallm > $O/bench_x.json 2> $O/bench_x.err || tail -3 $O/bench_x.err
python - <<PY
import json
d=json.load(open("$O/bench_x.json"))
c=d["config4_gemma3_4b_text"]; print("headline", round(d["value"],2), "| config4:", round(c["value"],2), "expl/s", round(c["ms_per_step"],1), "ms, gemm frac", round(c["gemm_frac_of_peak"],3), "gemm share", round(c["gemm_time_frac_of_step"],3))
PY
