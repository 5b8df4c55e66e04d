#!/bin/bash
# the judged command with its defaults (driver: python bench.py --gpus 1 --steps K --warmup W), wall-clocked
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3bench
mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python bench.py --steps 10 --warmup 2 ) > $O/bench.json 2> $O/bench.err; tail -5 $O/bench.err; cut -c1-300 $O/bench.json
