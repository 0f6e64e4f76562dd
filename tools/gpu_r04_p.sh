#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/bwd_items_profile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_p_bwd_items.txt
