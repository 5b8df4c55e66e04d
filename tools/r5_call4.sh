#!/bin/bash
# round 5, GPU call 4: persistent tile walk on/off on the short-K shapes (SigLIP K = 1152, Gemma-3 320-tile problems) and the Llama shapes, side by side
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r5c4; mkdir -p $O
python tools/gemm_ab.py --siglip --no-check --rounds 3 --iters 20 nopersist=tools/ab/liblrp_old.so persist=tools/ab/liblrp_pers_p2.so 2>&1 | grep -v amdgpu.ids | tee $O/gemm_ab_siglip.txt
python tools/gemm_ab.py --no-check --rounds 3 --iters 10 nopersist=tools/ab/liblrp_old.so persist=tools/ab/liblrp_pers_p2.so 2>&1 | grep -v amdgpu.ids | tee $O/gemm_ab_llama.txt
