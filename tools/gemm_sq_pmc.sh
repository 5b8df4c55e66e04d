#!/bin/bash
# SQ counter passes (MFMA pipe busy, wave cycles, instruction mix, effective clock) of the ping-pong GEMM instantiations inside the judged bench
# command, 4 layers (separate --pmc passes, kernel-trace only) -> gpurun_out/<out>/gemm_sq_pmc.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-gemmsq}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes --no-dropin --steps 1 --warmup 1 --layers 4"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
: > $O/gemm_sq_pmc.txt
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/gs_$i
  timeout 600 rocprofv3 --pmc $P --kernel-trace -d /tmp/gs_$i -- $BENCH > /tmp/gs_$i.log 2>&1
  db=$(find /tmp/gs_$i -name "*.db" | head -1)
  echo "## pass $i: $P" >> $O/gemm_sq_pmc.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db - gemm_pp >> $O/gemm_sq_pmc.txt 2>&1 || tail -5 /tmp/gs_$i.log >> $O/gemm_sq_pmc.txt
done
tail -n 120 $O/gemm_sq_pmc.txt | cut -c1-150
