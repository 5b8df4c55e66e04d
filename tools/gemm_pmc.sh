#!/bin/bash
# dev: TCP/TCC/TA counter passes over one cold and one hot GEMM (separate --pmc passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
P1="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
P3="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TD_TC_STALL_sum TD_TD_BUSY_sum"
P4="TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"
P5="TCC_BUSY_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum TCC_REQ_sum"
P6="TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i+1))
  for mode in cold hot; do
    rm -rf /tmp/pmc_$i_$mode
    timeout 120 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc_${i}_$mode -- python $R/tools/gemm_one.py 8192 28672 4096 $mode > /tmp/pmc_${i}_$mode.log 2>&1
    db=$(find /tmp/pmc_${i}_$mode -name "*.db" | head -1)
    echo "## pass $i $mode" >> $R/gpurun_out/gemm_pmc.txt
    python $R/tools/rocpd_pmc.py $db - gemm_nt >> $R/gpurun_out/gemm_pmc.txt 2>&1 || tail -5 /tmp/pmc_${i}_$mode.log >> $R/gpurun_out/gemm_pmc.txt
  done
done
