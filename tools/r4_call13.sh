#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c13; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/siglip_tower_time.py > $O/tower.txt 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/tower.txt
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tw -o tw -- python $GRAFT_REPO_ROOT/tools/siglip_tower_time.py > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/tw -name "*.db" | head -1) > $O/tower_kernels.txt 2>&1; head -30 $O/tower_kernels.txt | cut -c1-180
