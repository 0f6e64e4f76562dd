#!/bin/bash
# The forward's quadrant cut (SPLIT 1 / 2 / 4) on small images with long lists: fewer pixels per item saturate sooner.
TAG=${1:-r06_r}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--gaussians 500000" "--gaussians 1000000" "--gaussians 3000000" "--gaussians 6000000" "--scene v2 --gaussians 1000000" "--width 400 --height 400 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 6000000" "--width 1920 --height 1080 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_split.txt
  for sp in 0 1 2 4; do
    echo "split=$sp $(GSR_FWD_SPLIT_FORCE=$sp python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_split.txt
  done
done
