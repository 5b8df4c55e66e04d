#!/bin/bash
# Dev tool (GPU box): which parts of K1n pay in situ -- the judged bench command short (--layers 8), no profiler, each variant interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5k1np}
mkdir -p $O
BENCH="python bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8} --steps 8 --warmup 2"
for rep in 1 2 3; do
for v in fwd,bwd_qkv fwd,bwd_qkv,bwd_gu none; do
  case $v in none) f="--no-norm-fusion";; *) f="--norm-fusion-parts $v";; esac
  $BENCH $f > $O/b.json 2> $O/b.log
  echo "$v: $(python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print(round(d['value'],2), 'expl/s', round(d['ms_per_step'],3), 'ms/step')" 2>&1)" | tee -a $O/parts.txt
done; done
