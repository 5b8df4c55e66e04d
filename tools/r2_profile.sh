#!/bin/bash
# rocprofv3 kernel-trace summary of the judged bench command (profiles/r02_bench_*), then the two PMC passes for the GEMM traffic
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r2prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.log
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
head -40 $O/kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --layers 4 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.log
done
python - <<'PY'
import sqlite3, glob, os, json
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r2prof"
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    db=glob.glob(f"{O}/pmc_{c}/**/*.db", recursive=True)[0]
    cur=sqlite3.connect(db).cursor()
    rows=cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k,cn,v,n in rows:
        if "gemm_nt_swp" in k or "gemm_pp" in k or "attn32" in k or "smallm" in k:
            res.setdefault(k[:60],{})[cn]=(v,n)
for k,v in res.items(): print(k, {a:(f"{b[0]/b[1]:.4e} per launch", b[1]) for a,b in v.items()})
json.dump({k:{a:{"sum":b[0],"launches":b[1]} for a,b in v.items()} for k,v in res.items()}, open(O+"/pmc_summary.json","w"), indent=1)
PY
# the rocpd databases exceed what gpurun copies back: keep the text summaries only
find $O -name "*.db" -delete
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
