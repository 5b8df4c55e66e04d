#!/bin/bash
# profiling call: (1) rocprofv3 kernel-trace stats of the judged bench command (tails skipped: same kernel names at other shapes),
# (2) FETCH_SIZE / WRITE_SIZE PMC passes of a 4-layer run -> GEMM traffic per launch (stamped with the kernel source hash by tools/traffic.py),
# (3) SQ counter passes of the d = 128 attention kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $BENCH --steps 4 --warmup 1 > $O/bench_under_rocprof.json 2> $O/kt.log
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/kernel_stats.txt 2>&1
head -32 $O/kernel_stats.txt | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- $BENCH --steps 1 --warmup 1 --layers 4 > $O/pmc_$c.json 2> $O/pmc_$c.log
done
O=$O python - <<'PY'
import sqlite3, glob, os, json
O=os.environ["O"]
res={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    db=glob.glob(f"{O}/pmc_{c}/**/*.db", recursive=True)[0]
    cur=sqlite3.connect(db).cursor()
    rows=cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
    for k,cn,v,n in rows:
        if "gemm_pp" in k or "attn32" in k or "smallm" in k or "splitk" in k or "linear_stream" in k:
            res.setdefault(k[:140],{})[cn]=(v,n)
for k,v in res.items(): print(k[:100], {a:(f"{b[0]/b[1]:.4e} per launch", b[1]) for a,b in v.items()})
json.dump({k:{a:{"sum":b[0],"launches":b[1]} for a,b in v.items()} for k,v in res.items()}, open(O+"/pmc_summary.json","w"), indent=1)
PY
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
: > $O/pmc_attn.txt
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d /tmp/pa_$i -- python $GRAFT_REPO_ROOT/tools/attn_shape_bench.py > /tmp/pa_$i.log 2>&1
  db=$(find /tmp/pa_$i -name "*.db" | head -1)
  echo "## pass $i: $P" >> $O/pmc_attn.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db - attn32 >> $O/pmc_attn.txt 2>&1 || tail -5 /tmp/pa_$i.log >> $O/pmc_attn.txt
done
tail -n 70 $O/pmc_attn.txt | cut -c1-170
find $O -name "*.db" -delete
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
ls -la $O
