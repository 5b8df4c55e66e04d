#!/bin/bash
# rocprofv3 kernel-trace summaries of the judged bench command (without the small-M tables and the S=4096 probe that FOLLOW the timed region:
# their launches carry the same kernel names at other shapes) with the fused gated epilogues on (product) and off (A/B), then the two
# PMC passes for the GEMM traffic.  Summaries -> gpurun_out/r3prof (copied to profiles/r03_*).
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r3prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  LXT_AMD_GATED_FUSION=$f rocprofv3 --kernel-trace --stats -d $O/kt$f -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config5 --no-config4 --no-smallm > $O/bench_under_rocprof_f$f.json 2> $O/kt$f.log
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/kt$f -name "*.db" | head -1) > $O/kernel_stats_f$f.txt 2>&1
  head -30 $O/kernel_stats_f$f.txt | cut -c1-220
done
if [ "$1" != "nopmc" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --layers 4 --no-cpu-baseline --no-config5 --no-config4 --no-smallm > $O/pmc_$c.json 2> $O/pmc_$c.log
done
python - <<'PY'
import sqlite3, glob, os, json
O=os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3prof"
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    db=glob.glob(f"{O}/pmc_{c}/**/*.db", recursive=True)[0]
    cur=sqlite3.connect(db).cursor()
    rows=cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k,cn,v,n in rows:
        if "gemm_pp" in k or "attn32" in k or "smallm" in k or "splitk" in k:
            res.setdefault(k[:90],{})[cn]=(v,n)
for k,v in res.items(): print(k, {a:(f"{b[0]/b[1]:.4e} per launch", b[1]) for a,b in v.items()})
json.dump({k:{a:{"sum":b[0],"launches":b[1]} for a,b in v.items()} for k,v in res.items()}, open(O+"/pmc_summary.json","w"), indent=1)
PY
fi
find $O -name "*.db" -delete
rm -rf $O/kt1 $O/kt0 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
