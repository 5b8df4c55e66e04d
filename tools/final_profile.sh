#!/bin/bash
# round-end evidence: rocprofv3 kernel-trace stats of the bench command + FETCH_SIZE / WRITE_SIZE PMC passes (separate runs)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_bench_v4}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $db $R/gpurun_out/${TAG}_kernel_stats.txt > /dev/null
grep '"metric"' /tmp/kt.log | tail -1 > $R/gpurun_out/${TAG}_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmc_$C.log 2>&1
  db=$(find /tmp/pmc_$C -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $db $R/gpurun_out/${TAG}_pmc_$C.txt gemm_nt > /dev/null 2>&1 || echo "pmc $C failed" >> $R/gpurun_out/${TAG}_pmc_$C.txt
done
head -8 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-150
cat $R/gpurun_out/${TAG}_pmc_FETCH_SIZE.txt $R/gpurun_out/${TAG}_pmc_WRITE_SIZE.txt | cut -c1-150
