#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2p8
mkdir -p $O
for cfg in 7 30 28; do
  echo "== LRP_GEMM_BIG=$cfg hot" >> $O/hot.txt
  LRP_GEMM_BIG=$cfg python tools/kbench.py --what hot 2>&1 | grep "^gemm" >> $O/hot.txt
done
cat $O/hot.txt
cd /tmp && export TMPDIR=/tmp
for cfg in 7 30; do
  LRP_GEMM_BIG=$cfg rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_$cfg -o pmc -- python $GRAFT_REPO_ROOT/tools/kbench.py --what onegemm > $O/pmc_$cfg.log 2>&1
done
ls -R $O | head -30
python - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/r2p8"
for cfg in (7,30):
    files=glob.glob(f"{O}/pmc_{cfg}/**/*counter_collection.csv", recursive=True)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in files:
        for row in csv.DictReader(open(f)):
            k=row.get("Kernel_Name","")[:60]
            if "gemm" not in k: continue
            agg[k][row["Counter_Name"]]+=float(row["Counter_Value"]); 
    for k,v in agg.items():
        print(cfg,k,{a:f"{b:.3e}" for a,b in v.items()})
PY
