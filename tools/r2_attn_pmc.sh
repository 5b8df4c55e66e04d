#!/bin/bash
# SQ counters of the three bf16 d=128 attention kernels (kbench attnb), separate --pmc passes, kernel-trace only
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2attn
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
P3="SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_THREAD_CYCLES_VALU"
: > $O/pmc_attn.txt
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pa_$i
  timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/pa_$i -- python $R/tools/kbench.py --what attnb > /tmp/pa_$i.log 2>&1
  db=$(find /tmp/pa_$i -name "*.db" | head -1)
  echo "## pass $i" >> $O/pmc_attn.txt
  python $R/tools/rocpd_pmc.py $db - attn32 >> $O/pmc_attn.txt 2>&1 || tail -5 /tmp/pa_$i.log >> $O/pmc_attn.txt
done
tail -n 100 $O/pmc_attn.txt
