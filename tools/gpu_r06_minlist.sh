#!/bin/bash
# From which mean list length do checkpoints pay on small images?  Default threshold (1 200) against always-on.
TAG=${1:-r06_p}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--gaussians 100000" "--gaussians 200000" "--gaussians 300000" "--gaussians 400000" "--width 256 --height 256 --gaussians 50000" "--width 256 --height 256 --gaussians 100000" "--width 400 --height 400 --gaussians 200000" "--width 400 --height 400 --gaussians 300000" "--width 800 --height 800 --gaussians 500000" "--width 800 --height 800 --gaussians 1000000" "--scene v2 --gaussians 300000" "--scene v2 --gaussians 600000"; do
  echo "== $cfg" | tee -a $O/${TAG}_minlist.txt
  echo "off  $(GSR_CK_MIN_LIST=100000 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_minlist.txt
  echo "on   $(GSR_CK_MIN_LIST=1 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_minlist.txt
done
