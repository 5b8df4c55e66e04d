#!/bin/bash
# round 4, GPU call 10: attention32 templated on head dim (64 / 96 / 128) + lazy running max in the d = 128 forward
O=$GRAFT_REPO_ROOT/gpurun_out/r4c10; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attention or attn" > $O/test_attn.txt 2>&1; echo "attn tests rc=$?"; tail -5 $O/test_attn.txt | cut -c1-300
timeout 300 python tools/attn_shape_bench.py > $O/attn_bench.txt 2>&1; echo "bench rc=$?"; grep -v amdgpu.ids $O/attn_bench.txt
timeout 900 python -m pytest tests/test_gemma3_mm_engine_gpu.py tests/test_bert_engine_gpu.py tests/test_engine_gpu.py -m gpu -x -q > $O/test_eng.txt 2>&1; echo "engine tests rc=$?"; tail -5 $O/test_eng.txt | cut -c1-300
