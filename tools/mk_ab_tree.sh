#!/bin/bash
# Dev tool (build container): freeze a copy of the working tree's product files under tools/ab/<name>/ (git-ignored; travels with gpurun) and build
# ITS library, optionally with extra flags for gemm_pp.hip / attention32.hip -- the trees tools/r6_ab.sh compares on one box.
#   tools/mk_ab_tree.sh <name> ["<gemm_pp flags>"] ["<attention32 flags>"] [git-rev]      (git-rev: archive that commit instead of the working tree)
set -e
cd "$(dirname "$0")/.."
name=$1; gflags=$2; aflags=$3; rev=$4
rm -rf tools/ab/$name && mkdir -p tools/ab/$name
if [ -n "$rev" ]; then git archive $rev lrp-explains-transformers_amd include bench.py lxt_amd.py oracle | tar -x -C tools/ab/$name
else tar -c --exclude=build --exclude='*.so' --exclude=__pycache__ lrp-explains-transformers_amd include bench.py lxt_amd.py oracle | tar -x -C tools/ab/$name; fi
cd tools/ab/$name/lrp-explains-transformers_amd/csrc
make -j8 FLAGS_gemm_pp="$gflags" FLAGS_attention32="-fno-slp-vectorize $aflags" > /dev/null
rm -rf build
ls -la ../liblrp_hip.so
