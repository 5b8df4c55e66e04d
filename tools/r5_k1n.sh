#!/bin/bash
# Dev tool (GPU box): K1n (RMSNorm folded into the GEMM epilogues) -- kernel + engine tests, then the judged bench command short (--layers 8)
# under rocprofv3 --kernel-trace with the fusion on and off.   tools/r5_k1n.sh <outdir-name>
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5k1n}
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -k "norm_fused or gated_fused" -x -q > $O/tests.txt 2>&1
tail -3 $O/tests.txt
timeout 1200 python -m pytest tests/test_engine_gpu.py -k "norm_folded or llama_bf16 or batch_equals" -x -q -s >> $O/tests.txt 2>&1
grep -E "K1n|passed|failed|Error" $O/tests.txt | tail -12
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8}"
for tag in fused plain; do
  flag=""; [ $tag = plain ] && flag="--no-norm-fusion"
  rm -rf /tmp/kt_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o kt -- $BENCH $flag --steps 4 --warmup 1 > $O/bench_$tag.json 2> $O/bench_$tag.log)
  python tools/rocpd_stats.py $(find /tmp/kt_$tag -name "*.db" | head -1) > $O/stats_$tag.txt 2>&1
  echo "=== $tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], 'expl/s', d['ms_per_step'], 'ms/step', 'roofline', d['roofline']['frac'], 'all', d['roofline']['frac_all_gemm_launches'])" 2>&1)"
  grep -E "gemm_pp_kernel|rmsnorm|rms_rstd" $O/stats_$tag.txt | head -14 | cut -c1-60,88-150
  rm -rf /tmp/kt_$tag
done
