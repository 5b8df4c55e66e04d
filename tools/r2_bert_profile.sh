#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2bert
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "1 float32 20" "64 float32 5" "64 bfloat16 5"; do
  set -- $cfg
  rm -rf /tmp/kt_b
  rocprofv3 --kernel-trace --stats -d /tmp/kt_b -o kt -- python $GRAFT_REPO_ROOT/tools/bert_engine_run.py $1 $2 $3 > /tmp/kt_b.log 2>&1
  echo "## BertLRP efficient, B=$1, $2, $3 explanations" >> $O/kernel_stats.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/kt_b -name "*.db" | head -1) 2>&1 | head -24 >> $O/kernel_stats.txt
done
cat $O/kernel_stats.txt
