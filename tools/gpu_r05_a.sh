#!/bin/bash
# round 5, call A: correctness of the new blend kernels + same-box A/B against the round-4 library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05a_smoke.txt 2>&1; tail -2 $O/r05a_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/r05a_pytest.txt; tail -5 $O/r05a_pytest.txt
timeout 600 python tools/ab_variants.py --steps 200 r4@r4 new nodefer@nodefer r4b@r4 newb > $O/r05a_ab.txt 2>&1; cat $O/r05a_ab.txt
