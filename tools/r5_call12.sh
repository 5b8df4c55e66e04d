#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_hf_gpu.py -x -q 2>&1 | tail -40 | cut -c1-250
