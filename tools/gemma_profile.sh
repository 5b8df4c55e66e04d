#!/bin/bash
# rocprofv3 kernel-trace of BASELINE config 4's probes alone (tools/gemma_only.py): the text tower, then image + text -> per-kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-gemmaprof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in text image; do
  rocprofv3 --kernel-trace -d $O/kt_$w -o kt -- python $GRAFT_REPO_ROOT/tools/gemma_only.py $w 3 > $O/${w}_under_rocprof.json 2> $O/kt_$w.log
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/kt_$w -name "*.db" | head -1) > $O/kernel_stats_$w.txt 2>&1
  find $O/kt_$w -name "*.db" -delete; rm -rf $O/kt_$w
done
head -40 $O/kernel_stats_text.txt | cut -c1-170
