#!/bin/bash
# Development session: same-box A/B of the wave-priority builds (tools/build_variants.sh ... -DGSR_FWD_PRIO_STEP / -DGSR_BWD_PRIO_STEP)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/ab_variants.py --steps 150 --smoke "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prio_ab.txt | tail -20
