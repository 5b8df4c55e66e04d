#!/bin/bash
# Dev tool (GPU box): same-box A/B of two TREES -- the working tree against a frozen copy under tools/ab/<name>/ (git archive of an older commit with its own
# library built in place; git-ignored, travels with gpurun).  Each runs the judged bench command SHORT (--layers ${LAYERS:-8}) under rocprofv3 --kernel-trace;
# prints expl/s and the per-kernel averages.      tools/r6_ab.sh <outdir-name> tree [tree ...]      ("." = the working tree)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8} ${BENCH_FLAGS}"
for rep in $(seq 1 ${REPS:-1}); do
for tree in "$@"; do
  if [ "$tree" = . ]; then root=$GRAFT_REPO_ROOT; tag=now; else root=$GRAFT_REPO_ROOT/tools/ab/$tree; tag=$tree; fi
  rm -rf /tmp/kt_$tag
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o kt -- python $root/bench.py $FLAGS --steps 4 --warmup 1 > $O/bench_${tag}_$rep.json 2> $O/bench_${tag}_$rep.log)
  python tools/rocpd_stats.py --by-grid $(find /tmp/kt_$tag -name "*.db" | head -1) > $O/stats_${tag}_$rep.txt 2>&1
  echo "=== $tag rep $rep: $(python -c "import json,sys; d=json.loads(open('$O/bench_${tag}_$rep.json').read().strip().splitlines()[-1]); print(d['value'], 'expl/s', d['ms_per_step'], 'ms/step')" 2>&1)"
  grep -E "gemm_pp_kernel|attn32|gqa_reduce|rmsnorm|rope|prep" $O/stats_${tag}_$rep.txt | head -14 | cut -c1-60,88-150
  rm -rf /tmp/kt_$tag
done
done
