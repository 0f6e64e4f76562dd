#!/bin/bash
# Even finer strides on the small-image class (16 slots): 192 and 128 positions against the shipped 256.
TAG=${1:-r06_n}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
for cfg in "--gaussians 500000" "--gaussians 1000000" "--scene v2 --gaussians 1000000" "--width 400 --height 400 --gaussians 1000000" "--width 256 --height 256 --gaussians 1000000"; do
  echo "== $cfg" | tee -a $O/${TAG}_stride.txt
  echo "default  $(python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_stride.txt
  for ck in 3 2; do
    echo "ck=$ck   $(GSR_CK_DEBUG=1 GSR_CK_CHUNKS=$ck GSR_CK_SLOTS=16 python tools/c3_knobs.py $cfg 2>/dev/null)" | tee -a $O/${TAG}_stride.txt
  done
done
