#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "--gaussians 500000" "--gaussians 2000000" "--gaussians 3000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000"; do
  python tools/c3_knobs.py $cfg 2>/dev/null
  for ck in 8 6 4; do
    GSR_CK_CHUNKS=$ck python tools/c3_knobs.py $cfg 2>/dev/null
  done
done
