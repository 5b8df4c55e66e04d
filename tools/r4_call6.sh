#!/bin/bash
# round 4, GPU call 6: dgrad stream kernel (tests + A/B)
O=$GRAFT_REPO_ROOT/gpurun_out/r4c6; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -s -k "stream or skinny or smallm" > $O/test_stream.txt 2>&1; echo "tests rc=$?"; tail -15 $O/test_stream.txt | cut -c1-300
timeout 300 python tools/stream_ab.py > $O/stream_ab.txt 2>&1; cat $O/stream_ab.txt
timeout 300 python tools/stream_ab.py 4096 14336 > $O/stream_ab_down.txt 2>&1; cat $O/stream_ab_down.txt
timeout 300 python tools/stream_ab.py 128256 4096 > $O/stream_ab_head.txt 2>&1; cat $O/stream_ab_head.txt
