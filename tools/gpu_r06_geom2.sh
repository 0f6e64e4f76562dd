#!/bin/bash
# The fine-in-front table on the larger image classes (GSR_CK_FINE_TILES moves the class boundary), against their 512 x 8.
TAG=${1:-r06_o2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--width 640 --height 640 --gaussians 1000000" "--width 640 --height 640 --gaussians 2000000" "--width 800 --height 800 --gaussians 3000000" "--width 800 --height 800 --gaussians 6000000" \
           "--width 1920 --height 1080 --gaussians 1000000 --s0 0.05" "--width 1920 --height 1080 --gaussians 6000000" "--width 1920 --height 1080 --scene v2 --gaussians 6000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_geom.txt
  echo "default  $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_geom.txt
  echo "geom     $(GSR_CK_FINE_TILES=100000 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_geom.txt
  echo "geom ml1 $(GSR_CK_FINE_TILES=100000 GSR_CK_MIN_LIST=1 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_geom.txt
done
