#!/bin/bash
# Dev tool (GPU box): where does the run-to-run spread of the headline sit?  The judged bench command N times under rocprofv3 --kernel-trace,
# per-kernel totals of each run side by side
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5spread}
mkdir -p $O
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin"
for i in 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/kt_$i
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$i -o kt -- $BENCH --steps 4 --warmup 1 > $O/bench_$i.json 2> $O/bench_$i.log)
  python tools/rocpd_stats.py $(find /tmp/kt_$i -name "*.db" | head -1) > $O/stats_$i.txt 2>&1
  echo "run $i: $(python -c "import json; d=json.loads([l for l in open('$O/bench_$i.json') if l.startswith('{')][-1]); print(round(d['value'],2), round(d['ms_per_step'],1), round(d['roofline']['frac'],3))")"
  rm -rf /tmp/kt_$i
done
