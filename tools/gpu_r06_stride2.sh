#!/bin/bash
# Coarser strides with all 16 slots where lists are longer than 256 x 16 reaches (small images, many Gaussians).
TAG=${1:-r06_n2}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--width 256 --height 256 --gaussians 1000000" "--width 256 --height 256 --gaussians 300000" "--gaussians 2000000" "--gaussians 3000000" "--scene v2 --gaussians 2000000" "--scene v2 --gaussians 3000000" "--width 400 --height 400 --gaussians 1000000" "--width 400 --height 400 --gaussians 2000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_stride.txt
  echo "default  $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_stride.txt
  for ck in 6 8 12 16 24; do
    echo "ck=$ck   $(GSR_CK_CHUNKS=$ck GSR_CK_SLOTS=16 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_stride.txt
  done
done
