#!/bin/bash
# K8+K9: every input of a visible Gaussian requested together (two dependent round trips instead of five); variant k9old = HEAD~
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_round2.py -q -x 2>&1 | tail -4
timeout 300 python tools/ab_variants.py --smoke --no-extra-configs new old@k9old new2 old2@k9old
timeout 300 python tools/ab_variants.py --no-extra-configs --steps 50 --gaussians 6000000 new old@k9old new2 old2@k9old
} > $O/r04_k9.txt 2>&1
cat $O/r04_k9.txt
