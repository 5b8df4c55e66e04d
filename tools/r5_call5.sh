#!/bin/bash
# round 5, GPU call 5: new Python-side features (explicit Llama composite, fused drop-in MLP / RoPE, Gemma-3 4B-dim reference fixture, re-barred BERT cases)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c5; mkdir -p $O
timeout 900 python -m pytest tests/test_hf_gpu.py -q -s -k "explicit_composite or fused_mlp or gemma3_mm_4bdims" 2>&1 | grep -E "^\[|passed|failed|Error|assert|error" | tee $O/hf.txt | tail -30
timeout 900 python -m pytest tests/test_gemma3_mm_engine_gpu.py -q -s -k "full_dims" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee $O/g3mm.txt | tail
timeout 600 python -m pytest tests/test_bert_engine_gpu.py -q -s -k "ragged" 2>&1 | grep -E "^\[|passed|failed|Error|assert" | tee $O/bert.txt | tail -5
timeout 300 python tests/hf_family_worker.py bert_explicit_padded 2>&1 | grep -E "^\[|WORST|Error" | tee -a $O/bert.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], 'dropin', d.get('dropin_monkey_patch'))" | tee $O/dropin.txt
