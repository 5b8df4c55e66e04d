#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p4
mkdir -p $O
./tools/micro/trread > $O/trread.txt 2>&1
tail -n 8 $O/trread.txt
timeout 1500 python tools/explicit_forward_error.py > $O/fwd_err.txt 2>&1
grep -v amdgpu.ids $O/fwd_err.txt | tail -30
