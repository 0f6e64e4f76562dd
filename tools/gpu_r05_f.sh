#!/bin/bash
# round 5, call F: the round's new GPU tests alone
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
export GSR_REQUIRE_REF=1
timeout 1500 python -m pytest tests/test_gpu_round5.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" | tail -60 > $O/r05f_pytest.txt; tail -45 $O/r05f_pytest.txt | cut -c1-250
