#!/bin/bash
# round 5, GPU call 3: dK/dV ping-pong + lean gated epilogue: parity, timeline, in-situ A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c3; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or attention or gated" 2>&1 | tail -5 | tee $O/kernels.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "golden or fixture or ragged or left_padded" 2>&1 | tail -3 | tee -a $O/kernels.txt
L=lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/intree.so
cp tools/ab/liblrp_tl_pp.so $L; python tools/attn_dkv_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/dkv_timeline_pp.txt
cp /tmp/intree.so $L
tools/r5_ab.sh r5c3 nopp intree pp_prio pp_p0 2>&1 | tee $O/ab.txt
