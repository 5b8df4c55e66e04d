#!/bin/bash
# Dev tool (GPU box): the de-phased GEMM tile walk -- kernel tests, then the judged bench command short (--layers 8) with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5deph}
mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "gemm" -x -q > $O/tests.txt 2>&1
tail -4 $O/tests.txt
BENCH="python bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8} --steps 8 --warmup 2"
for rep in 1 2; do
for v in dephase lockstep dephase_allnorm lockstep_nonorm; do
  case $v in dephase) f="--dephase";; lockstep) f="";; dephase_allnorm) f="--dephase --norm-fusion-parts fwd,bwd_qkv,bwd_gu";; lockstep_nonorm) f="--no-norm-fusion";; esac
  timeout 300 $BENCH $f > $O/b.json 2> $O/b.log
  echo "$v: $(python -c "import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=d['roofline']; w=r['with_fused_epilogue_launches']; print(round(d['value'],2), 'expl/s', round(d['ms_per_step'],3), 'ms/step; plain', round(r['frac'],4), 'all', round(r['frac_all_gemm_launches'],4), 'gated_fwd', round(w['gated_fwd']['avg_launch_us'],1), 'gated_bwd', round(w['gated_bwd']['avg_launch_us'],1))" 2>&1)" | tee -a $O/parts.txt
done; done
