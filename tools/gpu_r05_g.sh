#!/bin/bash
# round 5, call G: five backward waves per SIMD (the kernel now needs 93 VGPRs) and the hardware exp, headline and synth-v2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
timeout 900 python tools/ab_variants.py --steps 200 base eu5=GSR_BLEND_WAVES_PER_SIMD=5,GSR_FWD_WAVES_PER_SIMD=4@eu5 fast=GSR_FAST_EXP=1 baseb eu5b=GSR_BLEND_WAVES_PER_SIMD=5,GSR_FWD_WAVES_PER_SIMD=4@eu5 > $O/r05g_ab.txt 2>&1; cat $O/r05g_ab.txt
timeout 900 python tools/ab_variants.py --steps 100 --scene=v2 v2base v2eu5=GSR_BLEND_WAVES_PER_SIMD=5,GSR_FWD_WAVES_PER_SIMD=4@eu5 v2fast=GSR_FAST_EXP=1 v2baseb > $O/r05g_ab_v2.txt 2>&1; cat $O/r05g_ab_v2.txt
