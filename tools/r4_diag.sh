#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "--- A: python - with ops first"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lxt_amd.ops as ops
import torch
x = torch.randn(64, 64, device="cuda").bfloat16()
print("transpose ok", ops.transpose(x).shape)
PY
echo "--- B: python - , Generator(device) first"
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -3
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lxt_amd.ops as ops
import torch
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)
x = (torch.randn(64, 64, generator=g, device=dev) * 0.1).to(torch.bfloat16)
print("transpose ok", ops.transpose(x).shape)
PY
echo "--- C: TAIL_SPLIT via env-free runner file"
cat > /tmp/run_ab.py <<'PY'
import sys, os, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import lxt_amd.ops as ops
ops.TAIL_SPLIT = sys.argv[1] == "1"
print("TAIL_SPLIT", ops.TAIL_SPLIT, flush=True)
runpy.run_path(os.environ["GRAFT_REPO_ROOT"] + "/tools/siglip_tower_time.py", run_name="__main__")
PY
python /tmp/run_ab.py 0 2>&1 | grep -v amdgpu.ids | tail -4
python /tmp/run_ab.py 1 2>&1 | grep -v amdgpu.ids | tail -4
