#!/bin/bash
# The work estimate of a cut quadrant: largest count among its sub-items (shipped) against their sum (GSR_EST_SUM=1), and
# uncut quadrants (GSR_FWD_SPLIT_FORCE=1) for reference.
TAG=${1:-r06_r3}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 3000000" "--scene v2 --gaussians 1000000" "--scene v2 --gaussians 2000000" "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 256 --height 256 --gaussians 1000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_est.txt
  echo "max      $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
  echo "sum      $(GSR_EST_SUM=1 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
  echo "uncut    $(GSR_FWD_SPLIT_FORCE=1 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
done
