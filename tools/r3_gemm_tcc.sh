#!/bin/bash
# L2 (TCC) hit / miss and memory-side request counters of the product GEMM, one --pmc pass per group
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/gemm_pmc_tcc.txt
i=0
for P in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_sum"; do
  i=$((i+1))
  for shape in "8192 28672 4096" "8192 4096 14336"; do
    rm -rf /tmp/tcc_$i
    timeout 120 rocprofv3 --pmc $P --kernel-trace -d /tmp/tcc_$i -- python $R/tools/gemm_one.py $shape > /tmp/tcc_$i.log 2>&1
    db=$(find /tmp/tcc_$i -name "*.db" | head -1)
    echo "## pass $i: $P | M N K = $shape" >> $O/gemm_pmc_tcc.txt
    python $R/tools/rocpd_pmc.py $db - gemm_pp >> $O/gemm_pmc_tcc.txt 2>&1 || tail -5 /tmp/tcc_$i.log >> $O/gemm_pmc_tcc.txt
  done
done
cat $O/gemm_pmc_tcc.txt | cut -c1-160
