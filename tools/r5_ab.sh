#!/bin/bash
# Dev tool (GPU box): in-situ A/B of library builds.  For every tools/ab/liblrp_<tag>.so named on the command line the library is swapped in and
# the judged bench command is run SHORT (--layers 8) under rocprofv3 --kernel-trace; prints expl/s and the per-kernel averages.
#   tools/r5_ab.sh <outdir-name> tag [tag ...]          ("intree" = the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $O
L=$GRAFT_REPO_ROOT/lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/intree.so
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --layers ${LAYERS:-8}"
for tag in "$@"; do
  if [ "$tag" = intree ]; then cp /tmp/intree.so $L; else cp tools/ab/liblrp_$tag.so $L; fi
  rm -rf /tmp/kt_$tag
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$tag -o kt -- $BENCH --steps 4 --warmup 1 > $O/bench_$tag.json 2> $O/bench_$tag.log)
  python tools/rocpd_stats.py $(find /tmp/kt_$tag -name "*.db" | head -1) > $O/stats_$tag.txt 2>&1
  echo "=== $tag: $(python -c "import json,sys; d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); print(d['value'], 'expl/s', d['ms_per_step'], 'ms/step')" 2>&1)"
  grep -E "gemm_pp_kernel|attn32|gqa_reduce|rmsnorm|rope|prep" $O/stats_$tag.txt | head -14 | cut -c1-60,88-150
  rm -rf /tmp/kt_$tag
done
cp /tmp/intree.so $L
