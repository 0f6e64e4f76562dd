#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export GSR_REQUIRE_REF=1
for i in 1 2 3 4 5 6; do timeout 300 python -m pytest "tests/test_gpu_parity.py::test_alpha_tile_bounds_leave_results_unchanged" -m gpu -q -s 2>&1 | grep -E "alpha vs reference|passed|failed|Error" ; done > $O/r05h_alpha.txt 2>&1
cat $O/r05h_alpha.txt | cut -c1-200 | tail -80
for i in 1 2 3; do timeout 600 python -m pytest "tests/test_gpu_round5.py::test_three_way_parity_on_synth_v2" -m gpu -q -s 2>&1 | grep -E "dL_drotations|dL_dscales|passed|failed" | cut -c1-120; done > $O/r05h_v2.txt 2>&1
cat $O/r05h_v2.txt | tail -60
