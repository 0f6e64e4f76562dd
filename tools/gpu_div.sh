#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp GSR_REQUIRE_REF=1
python tools/fuzz_v2_repeat.py 365 256 143 110 77 101 --runs 2 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_round5.py -x -q -k "segment or three_way or fuzz" 2>&1 | tail -5
python tools/ab_variants.py --steps 150 base rcp@rcp baseb rcpb@rcp 2>&1 | grep -v amdgpu.ids | tail -6
python tools/ab_variants.py --steps 100 --s0 0.05 deep deeprcp@rcp deepb 2>&1 | grep -v amdgpu.ids | tail -5
