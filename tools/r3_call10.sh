#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c17
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_baseline_size_gpu.py tests/test_dist_gpu.py -q -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for r in 1 2; do for f in 1; do
LXT_AMD_GATED_FUSION=$f timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5 > $O/bench_f${f}_r$r.json 2> $O/bench_f${f}_r$r.err
python - <<PY
import json
d=json.load(open("$O/bench_f${f}_r$r.json")); r=d["roofline"]
print("fusion=$f round=$r", round(d["value"],3), round(d["ms_per_step"],2), round(r["achieved"],1), r["launches"], round(r["avg_launch_us"],1), round(r["gemm_time_frac_of_step"],3))
PY
done; done
