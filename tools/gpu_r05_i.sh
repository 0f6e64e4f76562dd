#!/bin/bash
# round 5, call I: the work-list kernel back under 64 VGPRs (two clearing blocks per CU again): suite + same-box A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export GSR_REQUIRE_REF=1
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/r05i_pytest.txt; tail -4 $O/r05i_pytest.txt
unset GSR_REQUIRE_REF
timeout 900 python tools/ab_variants.py --steps 200 r4@r4 new r4b@r4 newb > $O/r05i_ab.txt 2>&1; cat $O/r05i_ab.txt
