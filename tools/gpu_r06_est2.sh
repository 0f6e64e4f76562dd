#!/bin/bash
# The forward's quadrant cut by image size: the shipped rule against uncut quadrants and a cut in 2, v1 and v2 scenes.
TAG=${1:-r06_r4}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for wh in 320 400 448 512 576 640 720; do
  for sc in "--gaussians 1000000" "--scene v2 --gaussians 1000000" "--gaussians 3000000"; do
    cfg="--width $wh --height $wh $sc"
    echo "== $cfg" | tee -a $O/${TAG}_est.txt
    for sp in 0 1 2; do
      echo "split=$sp $(GSR_FWD_SPLIT_FORCE=$sp python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_est.txt
    done
  done
done
