#!/bin/bash
# K1: position / scale / rotation / opacity fetched together, the SH record requested in front of the covariance arithmetic
# (GSR_K1_PREFETCH); variant k1late = the loads where they were.  Parity first, then same-box A/B.
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py -q -x 2>&1 | tail -4
timeout 300 python tools/ab_variants.py --smoke --no-extra-configs new old@k1late new2 old2@k1late
timeout 300 python tools/ab_variants.py --no-extra-configs --steps 50 --gaussians 6000000 new old@k1late new2 old2@k1late
timeout 300 python tools/ab_variants.py --no-extra-configs --width 512 --height 512 new old@k1late
} > $O/r04_k1.txt 2>&1
cat $O/r04_k1.txt
