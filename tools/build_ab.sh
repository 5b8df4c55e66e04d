#!/bin/bash
# Dev tool (build container): A/B builds of liblrp_hip.so -> tools/ab/liblrp_<tag>.so (git-ignored; they travel to the GPU box with gpurun).
#   tools/build_ab.sh <tag> "<extra flags for gemm_pp.hip>" "<extra flags for attention32.hip>" ["<extra flags for linear_stream.hip>"]
# Objects of the untouched sources are taken from csrc/build (run `make -C lrp-explains-transformers_amd/csrc` first).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/lrp-explains-transformers_amd/csrc
T=$ROOT/tools/ab/obj_$1
mkdir -p $T
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off"
/opt/rocm/bin/hipcc $FL $2 -c $C/gemm_pp.hip -o $T/gemm_pp.o &
/opt/rocm/bin/hipcc $FL -fno-slp-vectorize $3 -c $C/attention32.hip -o $T/attention32.o &
/opt/rocm/bin/hipcc $FL $4 -c $C/linear_stream.hip -o $T/linear_stream.o &
wait
OBJS=""
for s in capi gemm eltwise rowops attention linear_smallm; do OBJS="$OBJS $C/build/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/ab/liblrp_$1.so $OBJS $T/gemm_pp.o $T/attention32.o $T/linear_stream.o
rm -rf $T
echo "built tools/ab/liblrp_$1.so"
