#!/bin/bash
# round 5, GPU call 9: dK/dV element-wise / MFMA interleave (A32_DKV_IL) and key-block-major workgroup order (A32_DKV_ORDER): parity + in-situ A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c9; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "attention or golden or fixture or rccl or left_padded" 2>&1 | tail -4 | tee $O/tests.txt
tools/r5_ab.sh r5c9 noil il_noorder noil_order intree noil intree 2>&1 | grep -E "^===|attn32" | tee $O/ab.txt
