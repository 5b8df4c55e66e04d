#!/bin/bash
# same-box A/B of the headline bench under a Python-side measurement switch (alternating, two rounds):
#   bash tools/r3_ab_bench.sh LXT_AMD_GATED_FUSION     fused gated-MLP epilogues (1) vs GEMM + rule kernels (0)
#   bash tools/r3_ab_bench.sh LXT_AMD_PITCH_PAD        128-byte row-pitch padding of the long-K operands (1) vs none (0)
cd "$GRAFT_REPO_ROOT" || exit 1
V=${1:-LXT_AMD_GATED_FUSION}
O=gpurun_out/r3ab_$V
mkdir -p $O
export PYTHONUNBUFFERED=1
for r in 1 2; do for f in 1 0; do
env $V=$f timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-config5 --no-config4 > $O/bench_${f}_r$r.json 2> $O/bench_${f}_r$r.err
python - <<PY
import json
d=json.load(open("$O/bench_${f}_r$r.json")); r=d["roofline"]
print("$V=$f round=$r", round(d["value"],3), "expl/s", round(d["ms_per_step"],2), "ms | plain GEMM", round(r["achieved"],1), "TFLOP/s", r["launches"], "launches", round(r["avg_launch_us"],1), "us")
PY
done; done
