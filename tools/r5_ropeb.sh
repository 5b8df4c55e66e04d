#!/bin/bash
# Dev tool (GPU box): RoPE's backward folded into the dQ store and dK's group sum -- tests, then the judged bench short with and without
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5ropeb}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "test_attention" -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_baseline_size_gpu.py -x -q >> $O/tests.txt 2>&1
tail -3 $O/tests.txt
BENCH="python bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8} --steps 8 --warmup 2"
for rep in 1 2; do
for v in fused standalone; do
  case $v in fused) f="";; standalone) f="--no-rope-bwd-fusion";; esac
  timeout 300 $BENCH $f > $O/b.json 2> $O/b.log
  echo "$v: $(python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(round(d['value'],2), 'expl/s', round(d['ms_per_step'],3), 'ms/step')" 2>&1)" | tee -a $O/parts.txt
done; done
