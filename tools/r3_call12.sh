#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c16
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 400 python tools/pitch_probe.py > $O/pitch.txt 2>&1; cat $O/pitch.txt
