#!/bin/bash
# round 5, GPU call 2: where does the d = 128 dK/dV kernel spend its time?  in-kernel timeline (two builds) + SQ counters of the product build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c2; mkdir -p $O
L=lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/intree.so
for t in tl_p0 tl_p2; do cp tools/ab/liblrp_$t.so $L; echo "== $t"; python tools/attn_dkv_timeline.py 2>&1 | grep -v amdgpu.ids; done | tee $O/dkv_timeline.txt
cp /tmp/intree.so $L
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_WAVES"
: > $O/pmc_dkv.txt
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1)); rm -rf /tmp/pa_$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $P --kernel-trace -d /tmp/pa_$i -- python $GRAFT_REPO_ROOT/tools/attn_dkv_only.py > /tmp/pa_$i.log 2>&1)
  db=$(find /tmp/pa_$i -name "*.db" | head -1)
  echo "## pass $i: $P" >> $O/pmc_dkv.txt
  python tools/rocpd_pmc.py $db - dkv >> $O/pmc_dkv.txt 2>&1 || tail -5 /tmp/pa_$i.log >> $O/pmc_dkv.txt
done
cat $O/pmc_dkv.txt | cut -c1-200
