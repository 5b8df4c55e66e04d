#!/bin/bash
# Runs the GPU test groups each in its own process (a memory fault in one kernel must not hide the
# rest), then the kernel micro-benchmarks.  Logs go to gpurun_out/<tag>_*.log
TAG=${1:-ci}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/${TAG}_summary.log
  timeout $to "$@" > gpurun_out/${TAG}_$name.log 2>&1
  echo "rc=$? $(tail -n 1 gpurun_out/${TAG}_$name.log)" | tee -a gpurun_out/${TAG}_summary.log
}
rm -f gpurun_out/${TAG}_summary.log
rocminfo | grep -E "Marketing Name|gfx" | head -4 >> gpurun_out/${TAG}_summary.log
for grp in gemm transpose linear_eps eps_scale goldens_elementwise gated rope rmsnorm readout attention goldens_attention; do
  run k_$grp 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "$grp" -p no:cacheprovider
done
run engine 900 python -m pytest tests/test_engine_gpu.py -m gpu -q -s -p no:cacheprovider
run kbench 600 python tools/kbench.py
cat gpurun_out/${TAG}_summary.log
