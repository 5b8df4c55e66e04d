#!/bin/bash
# SQ counters of the product GEMM (plain NT, one layer's gate/up-forward-sized problem and the long-K down forward), separate --pmc passes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/gemm_pmc_sq.txt
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  for shape in "8192 28672 4096" "8192 4096 14336"; do
    rm -rf /tmp/sq_$i
    timeout 120 rocprofv3 --pmc $P --kernel-trace -d /tmp/sq_$i -- python $R/tools/gemm_one.py $shape > /tmp/sq_$i.log 2>&1
    db=$(find /tmp/sq_$i -name "*.db" | head -1)
    echo "## pass $i: $P | M N K = $shape" >> $O/gemm_pmc_sq.txt
    python $R/tools/rocpd_pmc.py $db - gemm_pp >> $O/gemm_pmc_sq.txt 2>&1 || tail -5 /tmp/sq_$i.log >> $O/gemm_pmc_sq.txt
  done
done
cat $O/gemm_pmc_sq.txt | cut -c1-160
