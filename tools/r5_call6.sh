#!/bin/bash
# round 5, GPU call 6: in-kernel split reduction of the weight-streaming dgrad: parity + the driver's small-M table with and without it
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c6; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "linear_stream or smallm or skinny" 2>&1 | tail -4 | tee $O/tests.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/smallm_ab.txt
import sys, json, torch
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
import lxt_amd.ops as ops
cfg = dict(b.LLAMA3_8B, n_layers=32)
for flag in (False, True, False, True):
    ops.STREAM_DGRAD_INKERNEL_REDUCE = flag
    r = b.smallm_roofline(ops, torch.bfloat16, torch.device("cuda"), cfg, 4)
    print("in-kernel reduce" if flag else "two launches    ", " | ".join(f"M={t['M']}: fwd {t['fwd_us']:.1f} dgrad {t['dgrad_us']:.1f} pair {t['pair_frac']:.3f}" for t in r["table_gate_up_sized"]))
PY
