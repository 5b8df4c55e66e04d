#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r4c15; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "tail_split" > $O/test.txt 2>&1; echo "tests rc=$?"; tail -3 $O/test.txt | cut -c1-300
python - <<'PY' > $O/tower_ab.txt 2>&1
import subprocess, sys, os
import importlib
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lxt_amd.ops as ops
import runpy
for flag in (False, True):
    ops.TAIL_SPLIT = flag
    print("TAIL_SPLIT", flag, flush=True)
    runpy.run_path(os.environ["GRAFT_REPO_ROOT"] + "/tools/siglip_tower_time.py", run_name="__main__")
PY
echo "ab rc=$?"; grep -v amdgpu.ids $O/tower_ab.txt
