#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p15
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu -s > $O/test_all.txt 2>&1
grep -aE "H4096/S2048|drop-in H|bert-base explicit|passed|failed|FAILED|dense seed" $O/test_all.txt | grep -v "print\|f\"" | cut -c1-260
