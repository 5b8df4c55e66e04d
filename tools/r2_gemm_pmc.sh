#!/bin/bash
# fabric traffic + clock of our GEMM vs hipBLASLt's on the same shapes (tools/kbench.py --what onegemm), separate --pmc passes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2gemm
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc_gemm.txt
i=0
for P in "FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1)); rm -rf /tmp/pg_$i
  timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/pg_$i -- python $R/tools/kbench.py --what onegemm > /tmp/pg_$i.log 2>&1
  db=$(find /tmp/pg_$i -name "*.db" | head -1)
  echo "## pass $i" >> $O/pmc_gemm.txt
  python $R/tools/rocpd_pmc.py $db - >> $O/pmc_gemm.txt 2>&1 || tail -5 /tmp/pg_$i.log >> $O/pmc_gemm.txt
done
grep -v "^$" $O/pmc_gemm.txt | grep -A9 "gemm_nt_w4\|Cijk\|## pass" | cut -c1-150
