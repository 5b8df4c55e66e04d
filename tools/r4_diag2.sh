#!/bin/bash
# (historical) ran diagnostic builds of the library (tools/ab/liblrp_diagN.so: the dK/dV kernel with parts of its work removed; build switches not committed) -> profiles/r04_dkv_diagnostic_builds.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "release:"; python tools/attn_dkv_only.py 2>&1 | grep -v amdgpu.ids
cp lrp-explains-transformers_amd/liblrp_hip.so /tmp/release.so
for n in 1 2 3 4; do
  cp tools/ab/liblrp_diag$n.so lrp-explains-transformers_amd/liblrp_hip.so
  echo "diag $n:"; python tools/attn_dkv_only.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/release.so lrp-explains-transformers_amd/liblrp_hip.so
