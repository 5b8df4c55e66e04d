#!/bin/bash
# round 5, GPU call 11: row-pitch padding of the big stored weights (NN dgrad operand): parity + headline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c11; mkdir -p $O
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_baseline_size_gpu.py tests/test_hf_gpu.py -x -q -k "not seed_set and not dropin_fp32_full" 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 --no-config4 --no-smallm --no-extra-modes 2>&1 | tail -1 > $O/bench.json
python -c "import json; d=json.loads(open('$O/bench.json').read()); print('headline', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], d['roofline']['frac_all_gemm_launches'], 'dropin', d.get('dropin_monkey_patch',{}).get('value'))"
