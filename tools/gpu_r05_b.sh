#!/bin/bash
# round 5, call B: correctness of the new blend kernels + same-box A/B against the round-4 library and the two switches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05b_smoke.txt 2>&1; tail -2 $O/r05b_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r05b_pytest.txt; tail -5 $O/r05b_pytest.txt
timeout 900 python tools/ab_variants.py --steps 200 r4@r4 new nopop@nopop nodefer@nodefer h8=GSR_BWD_HALVES=8 h6=GSR_BWD_HALVES=6 r4b@r4 newb nopopb@nopop nodeferb@nodefer > $O/r05b_ab.txt 2>&1; cat $O/r05b_ab.txt
