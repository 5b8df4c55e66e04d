#!/bin/bash
# A/B of attention-kernel builds: every tools/ab/liblrp_*.so is put in place of the product library in turn; outputs are compared
# with those of tools/ab/liblrp_old.so on seeded inputs (tools/attn_dump.py), then tools/kbench.py --what attnb is timed (2 rounds).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2attn
mkdir -p $O
L=lrp-explains-transformers_amd/liblrp_hip.so
cp $L /tmp/product.so
cp tools/ab/liblrp_old.so $L; timeout 300 python tools/attn_dump.py save /tmp/ref.pt
: > $O/ab.txt
for f in tools/ab/liblrp_*.so; do
  n=$(basename $f .so); [ "$n" = liblrp_old ] && continue
  cp $f $L; echo "== cmp $n" | tee -a $O/ab.txt; timeout 300 python tools/attn_dump.py cmp /tmp/ref.pt 2>&1 | grep "BAD\|RESULT" | tee -a $O/ab.txt
done
for rep in 1 2; do
  for f in tools/ab/liblrp_*.so; do
    cp $f $L; echo "== $(basename $f .so)"; timeout 300 python tools/kbench.py --what attnb 2>&1 | grep "attn fwd\|attn dq\|attn dkv"
  done
done | tee -a $O/ab.txt
cp /tmp/product.so $L
