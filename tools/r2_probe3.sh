#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2p3
mkdir -p $O
timeout 1500 python tools/explicit_site_sensitivity.py > $O/sites.txt 2>&1
cat $O/sites.txt | tail -12
timeout 600 python -m pytest tests/test_api_gpu.py -q -m gpu -x > $O/test_api.txt 2>&1; tail -3 $O/test_api.txt
