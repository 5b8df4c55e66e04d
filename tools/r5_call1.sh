#!/bin/bash
# round 5, GPU call 1: kernel parity of the new default build (persistent GEMM walk, dK/dV read-ahead), the small explicit-mode cases against their
# reference-run yardsticks, then the in-situ A/B of the variant builds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c1; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or attention or gated" 2>&1 | tail -5 | tee $O/kernels.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -q -s -k "ragged or left_padded or dense_seed or golden or fixture" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee $O/small_cases.txt | tail -40
timeout 600 python -m pytest tests/test_bert_engine_gpu.py -q -s -k "ragged" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee -a $O/small_cases.txt | tail -12
timeout 300 python tests/hf_family_worker.py bert_explicit_padded 2>&1 | grep -E "^\[|WORST|Error" | tee -a $O/small_cases.txt
tools/r5_ab.sh r5c1 old intree pers_p2 pers_st20_p2ew st20_p3ew pers_st8 2>&1 | tee $O/ab.txt
