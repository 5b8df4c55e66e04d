#!/bin/bash
# round 4, GPU call 8: fused stabiliser in the dgrad stream kernel (tests), explicit API tests, default bench table
O=$GRAFT_REPO_ROOT/gpurun_out/r4c8; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_api_gpu.py -m gpu -x -q -s -k "stream or linear or eps or golden" > $O/test.txt 2>&1; echo "tests rc=$?"; tail -6 $O/test.txt | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-config5 --no-config4 --no-extra-modes > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json,os
p=json.loads(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r4c8/bench.json").read().strip().splitlines()[-1])
print("headline", p["value"])
s=p['roofline_linear_eps_smallm']
for t in ('table_gate_up_sized','table_lm_head'):
    print(t)
    for row in s[t]:
        print('  '+' '.join(f"{k}={round(v,2) if isinstance(v,float) else v}" for k,v in row.items()))
PY
