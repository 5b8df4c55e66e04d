#!/bin/bash
# attention32 staging A/B (buffer_load..lds vs global_load_lds), attention tests, B=1 eager vs hipGraph
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c10
mkdir -p $O
export PYTHONUNBUFFERED=1
L=lrp-explains-transformers_amd/liblrp_hip.so
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_baseline_size_gpu.py -q -x -k "attention" > $O/pytest_attn.txt 2>&1; tail -3 $O/pytest_attn.txt
cp $L /tmp/product.so
for rep in 1 2; do
  for f in /tmp/product.so tools/ab/attn_glds.so; do cp $f $L; echo "== $(basename $f)"; timeout 200 python tools/kbench.py --what attnb 2>&1 | grep "attn"; done
done > $O/attn_ab.txt 2>&1
cp /tmp/product.so $L
cat $O/attn_ab.txt
timeout 300 python bench.py --batch 1 --steps 10 --warmup 2 --graph --no-cpu-baseline --no-config5 2>$O/b1g.err | cut -c1-240 > $O/bench_b1_graph.txt; cat $O/bench_b1_graph.txt; tail -3 $O/b1g.err
timeout 300 python bench.py --batch 1 --steps 10 --warmup 2 --no-cpu-baseline --no-config5 2>/dev/null | cut -c1-240
